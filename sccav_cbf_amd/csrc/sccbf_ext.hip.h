// sccbf_ext.hip.h - the filter for the unicycle-input (DUM) and steer-rate (SADBM) models
// Part of the sccbf library (one translation unit, see sccbf.hip); device code for gfx950.
//
// Same structure as k_group (G lanes per QP, rows in registers and in the per-wavefront LDS
// table, the same active-set iteration); only the contraction of the partials into a row and
// the input / output maps differ, so the kernels of the two bicycle models stay as they are.
#pragma once
#include "sccbf_filter.hip.h"

namespace {

template <typename T> struct XArgs {
    T* slip;   // SADBM: [A][2] in/out = beta, beta_ref_last; unused for DUM
    T dt, inv_dt;
};

// DUM: reference cbf/cbf.py:237-245, 279-286.   Lg_h = [h_v, h_th],  Lf_h = v (cos(th) h_x + sin(th) h_y)
// SADBM: reference cbf/cbf.py:338-347, 398-405. Lg_h = [h_v, h_beta],
//        Lf_h = h_x v cos(th + beta) + h_y v sin(th + beta) + h_th v sin(beta) / lr
template <bool INLINE6 = false, typename T>
__device__ inline void ext_slot_row_(int type, T (&o)[8], const Ego<T>& e, const KArgs<T>& p, bool sadbm,
                                     bool cone_beta_state, T beta, T cthb, T sthb, T sb,
                                     T& a0, T& a1, T& b, T& h)
{
    Partials<T> d;
    T hb = T(0);
    // a wavefront without a box among its slots skips the conversion as a whole (scalar branch, as in slot_partials_)
    if (__any(type == SCCBF_OBS_ELLIPSE_BOX || type == SCCBF_OBS_CONE_BOX)) {
        keep_branch_();
        if (type == SCCBF_OBS_ELLIPSE_BOX || type == SCCBF_OBS_CONE_BOX) {
            // the obstacle the box stands for: the ellipse as written, the cone with its velocity (speed, 0) resolved
            // as the bicycle models take it, or as written (theta_o = 0) when it has to follow the agent's beta
            T q[8];
            if (type == SCCBF_OBS_ELLIPSE_BOX) {
                box_ellipse_(o, q);
                type = SCCBF_OBS_ELLIPSE;
            } else if (cone_beta_state) {
                box_cone_(o, q);
                type = SCCBF_OBS_CONE;
            } else {
                box_cone_prep_(o, q);
                type = SCCBF_OBS_CONE_PREP;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = q[i];
        }
    }
    switch (type) {
    case SCCBF_OBS_ELLIPSE: ellipse_(e.x, e.y, o, d); break;
    case SCCBF_OBS_CONE:
        if (cone_beta_state) o[5] = beta;
        cone_(e.x, e.y, e.th, e.v, e.cth, e.sth, o, p.zero_tol, d);
        hb = d.hth;  // CollisionCone2D.dbeta, reference cbf/obstacles.py:460-466
        break;
    case SCCBF_OBS_LANE: lane_(e.x, e.y, o, p.zero_tol, d); break;
#ifndef SCCBF_NO_LANE6
    case SCCBF_OBS_LANE6: lane6_<INLINE6>(e.x, e.y, o, p.zero_tol, d); break;
#endif
    case SCCBF_OBS_RADIAL: radial_(e.x, e.y, e.v, o, d); break;
    case SCCBF_OBS_ELLIPSE_PREP: ellipse_prep_(e.x, e.y, o, d); break;
    case SCCBF_OBS_CONE_PREP:
        // the obstacle's velocity was resolved with the slot's own beta: the slot keeps it
        cone_prep_(e.x, e.y, e.th, e.v, e.cth, e.sth, o, p.zero_tol, d);
        hb = d.hth;
        break;
    default:
        a0 = T(0); a1 = T(0); b = T(-1); h = T(1);
        return;
    }
    h = d.h;
    a0 = d.hv;
    if (sadbm) {
        a1 = hb;
        const T lf_h = e.v * (cthb * d.hx + sthb * d.hy) + (e.v * sb * p.inv_lr) * d.hth;
        b = -(lf_h + p.alpha * d.h + d.ht);
    } else {
        a1 = d.hth;
        const T lf_h = e.v * (e.cth * d.hx + e.sth * d.hy);
        b = -(lf_h + p.alpha * d.h + d.ht);
    }
}

// Everything of one QP of these models that does not depend on a slot: the agent's state, its
// nominal input in QP coordinates and, for the steer-rate model, the slip state and its sines / cosines.
template <typename T> struct ExtEgo {
    Ego<T> e;
    T ur0, ur1;
    typename VT<T>::v2 sl;  // SADBM: beta, beta_ref_last
    T beta_ref, cthb, sthb, sb;
    bool sadbm, cone_beta_state;
    int bad;
};

template <typename T>
__device__ inline void ext_prologue_(const KArgs<T>& p, const XArgs<T>& x, long long qp,
                                     const typename VT<T>::v4& sv, const typename VT<T>::v2& urv, ExtEgo<T>& q)
{
    using v2 = typename VT<T>::v2;
    Ego<T>& e = q.e;
    q.sadbm = p.dynamics == SCCBF_DYN_SADBM;
    q.sl = v2{T(0), T(0)};
    if (q.sadbm) q.sl = reinterpret_cast<const v2*>(x.slip)[qp];
    e.x = sv.x; e.y = sv.y; e.th = sv.z; e.v = sv.w;
    q.ur0 = urv.x;
    q.ur1 = urv.y;
    q.bad = !(finite_(q.ur0) && finite_(q.ur1) && finite_(e.x) && finite_(e.y) && finite_(e.th) && finite_(e.v) &&
              finite_(q.sl.x) && finite_(q.sl.y));
    sincos_(e.th, &e.sth, &e.cth);
    e.derive(p);
    q.beta_ref = T(0);
    q.cthb = e.cth;
    q.sthb = e.sth;
    q.sb = T(0);
    e.un0 = q.ur0;
    e.un1 = q.ur1;
    if (q.sadbm) {
        q.beta_ref = atan_(tan_(q.ur1) * p.k_in);       // reference cbf/cbf.py:359
        e.un1 = (q.beta_ref - q.sl.y) * x.inv_dt;       // cbf.py:367-372
        T cb;
        sincos_(e.th + q.sl.x, &q.sthb, &q.cthb);
        sincos_(q.sl.x, &q.sb, &cb);
    }
    q.cone_beta_state = q.sadbm && (p.flags & SCCBF_FLAG_CONE_BETA_STATE) != 0;
}

// u = u_nom + Lc^-T w, or the nominal input when the QP has no answer; SADBM integrates its slip angle
template <typename T>
__device__ inline void ext_epilogue_(const KArgs<T>& p, const XArgs<T>& x, long long qp, const ExtEgo<T>& q,
                                     int st, T w0, T w1)
{
    using v2 = typename VT<T>::v2;
    T u0 = q.e.un0, u1 = q.e.un1;
    if (st == SCCBF_STATUS_OPTIMAL) {
        const T y1 = w1 * p.inv_l11;
        const T y0 = (w0 - p.l10 * y1) * p.inv_l00;
        u0 += y0;
        u1 += y1;
    }
    v2 out = v2{u0, u1};
    if (q.sadbm) {
        if (st == SCCBF_STATUS_NONFINITE) {
            out = v2{q.ur0, q.ur1};  // nothing to integrate: inputs pass through, the slip state stays
        } else {
            const T beta = q.sl.x + u1 * x.dt;          // reference cbf/cbf.py:421
            out.y = atan_(tan_(beta) * p.k_out);        // cbf.py:430
            reinterpret_cast<v2*>(x.slip)[qp] = v2{beta, q.beta_ref};
        }
    }
    reinterpret_cast<v2*>(p.u_out)[qp] = out;
    p.status[qp] = st;
}

template <typename T, int G, int RPL>
__global__ __launch_bounds__(kBlock) void k_group_ext(const KArgs<T> p, const XArgs<T> x)
{
    using v4 = typename VT<T>::v4;
    const int g = (int)(threadIdx.x % G);
    const long long qp = (long long)blockIdx.x * (kBlock / G) + threadIdx.x / G;
    if (qp >= p.A) return;  // groups are lane-aligned: a whole group leaves together
    LaneLoads<T, RPL> L;
    group_load_<T, RPL, false, false>(p, qp, g, G, L);
    ExtEgo<T> q;
    ext_prologue_(p, x, qp, L.sv, L.urv, q);
    int bad = q.bad;

    T at0[RPL], at1[RPL], bt[RPL];
    T chk = T(0);
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
        const int m = g + r * G;
        T o[8] = {L.oa[r].x, L.oa[r].y, L.oa[r].z, L.oa[r].w, L.ob[r].x, L.ob[r].y, L.ob[r].z, L.ob[r].w};
        const int ty = m < p.M ? L.ty[r] : SCCBF_OBS_EMPTY;
        T a0, a1, b, h;
        ext_slot_row_<(RPL <= 4)>(ty, o, q.e, p, q.sadbm, q.cone_beta_state, q.sl.x, q.cthb, q.sthb, q.sb, a0, a1, b, h);
        if (p.rows_out && m < p.M) reinterpret_cast<v4*>(p.rows_out)[(long long)m * p.A + qp] = v4{a0, a1, b, h};
        chk += a0 * T(0) + a1 * T(0) + b * T(0);
        to_wspace_(a0, a1, b, q.e, p, at0[r], at1[r], bt[r]);
    }
    bad |= !(chk == T(0));
    bad = (int)group_or<G>((unsigned)bad);

    constexpr int kMpad = G * RPL + 1;
    constexpr int kPlane = (kBlock / G) * kMpad;
    constexpr bool kLdsPick = (size_t)3 * kPlane * sizeof(T) <= 53 * 1024;
    __shared__ T s_rows[kLdsPick ? 3 * kPlane : 1];
    const int qslot = (int)(threadIdx.x / G) * kMpad;
    if (kLdsPick) {
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            s_rows[qslot + r * G + g] = at0[r];
            s_rows[kPlane + qslot + r * G + g] = at1[r];
            s_rows[2 * kPlane + qslot + r * G + g] = bt[r];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    const Pivot<T> s = active_set_<T, G, RPL, kLdsPick>(p.M, g, at0, at1, bt, s_rows, kPlane, qslot, bad);
    if (g != 0) return;
    ext_epilogue_(p, x, qp, q, s.st, s.w0, s.w1);
}

// This kernel carries more per agent than k_wave (the slip state, its sines and cosines): held to the 96 registers of five
// wavefronts per SIMD it spills and runs 10 - 15 % slower than at four (128 registers), with 16, 32 and 64 lanes per QP alike.
template <typename T> constexpr int kWaveExtWaves = sizeof(T) == 4 ? 4 : 1;

// G lanes per QP, rows staged in LDS only: any M up to SCCBF_MAX_ROWS (k_wave for these models, the same
// two thread layouts).
template <typename T, int QPB, int G, bool PAIRED = false, int NB = kWaveBatch<T>>
__global__ __launch_bounds__(QPB * G) __attribute__((amdgpu_waves_per_eu(kWaveExtWaves<T>))) void k_wave_ext(const KArgs<T> p, const XArgs<T> x)
{
    using v4 = typename VT<T>::v4;
    using v2 = typename VT<T>::v2;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    static_assert(QPB * G % 64 == 0, "whole wavefronts");
    const int M = p.M;
    const WaveTables<T> tab(smem_raw, M, QPB, G);
    const int stride = wave_table_stride(M, G);
    const int qa = (int)(threadIdx.x % QPB), rsub = (int)(threadIdx.x / QPB);  // forming rows
    const int qs = (int)(threadIdx.x / G), g = (int)(threadIdx.x % G);         // solving

    long long qp = wave_first_agent<QPB, PAIRED>() + qa;
    const bool live = qp < p.A;
    if (!live) qp = p.A - 1;  // the thread stays for the barriers; its stores are masked
    constexpr bool kFirst = true;  // the first scan is taken while the rows are formed, as in k_wave
    if (threadIdx.x < QPB) {
        tab.bad[threadIdx.x] = 0;
        tab.first[threadIdx.x] = (int)0x80000000u;
    }

    ExtEgo<T> q;
    ext_prologue_(p, x, qp, reinterpret_cast<const v4*>(p.state)[qp], reinterpret_cast<const v2*>(p.u_ref)[qp], q);
    int bad = q.bad;
    __syncthreads();

    T* table = tab.rows + (size_t)qa * stride;
    T chk = T(0);  // stays 0 while every row is finite
    int kfirst = (int)0x80000000u;
    staged_rows_<T, G, false, NB>(p, qp, rsub, [&](int m, const v4& oa, const v4& ob, int ty) {
        T o[8] = {oa.x, oa.y, oa.z, oa.w, ob.x, ob.y, ob.z, ob.w};
        T a0, a1, b, h;
        ext_slot_row_(ty, o, q.e, p, q.sadbm, q.cone_beta_state, q.sl.x, q.cthb, q.sthb, q.sb, a0, a1, b, h);
        if (p.rows_out && live) reinterpret_cast<v4*>(p.rows_out)[(long long)m * p.A + qp] = v4{a0, a1, b, h};
        chk += a0 * T(0) + a1 * T(0) + b * T(0);
        T t0, t1, tb;
        to_wspace_(a0, a1, b, q.e, p, t0, t1, tb);
        wave_table_put(table, m, t0, t1, tb);
        if (kFirst) {
            const int key = wave_first_key_(m, tb);
            kfirst = key > kfirst ? key : kfirst;
        }
    });
    wave_table_pad<T, G>(table, M, rsub);
    if (kFirst) atomicMax(&tab.first[qa], kfirst);
    if (bad | !(chk == T(0))) tab.bad[qa] = 1;
    __syncthreads();
    const Pivot<T> s = wave_active_set_<T, G, kFirst>(M, g, tab.rows + (size_t)qs * stride, tab.bad[qs], tab.first[qs]);
    if (g == 0) {
        tab.answer[2 * qs] = s.w0;
        tab.answer[2 * qs + 1] = s.w1;
        tab.status[qs] = s.st;
    }
    __syncthreads();
    if (rsub == 0 && live) ext_epilogue_(p, x, qp, q, tab.status[qa], tab.answer[2 * qa], tab.answer[2 * qa + 1]);
}

}  // namespace
