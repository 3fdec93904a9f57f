// sccbf_rows.hip.h - barrier functions of the obstacle kinds, constraint rows, ego quantities
// Part of the sccbf library (one translation unit, see sccbf.hip); device code for gfx950.
#pragma once
#include "sccbf_qp.hip.h"

namespace {

// Inside a block whose condition is uniform over the wavefront: an empty statement the compiler may not hoist, so the block
// stays behind its scalar branch instead of being evaluated by everybody and selected away afterwards.
__device__ inline void keep_branch_() { __asm__ volatile(""); }

// --------------------------------------------------------------------- barriers
template <typename T> struct Partials {
    T h, hx, hy, hth, hv, ht;
};

// Ellipse2D, reference cbf/obstacles.py:183-236, 310-317 (dt as written: no rotation).
template <typename T>
__device__ inline void ellipse_(T x, T y, const T o[8], Partials<T>& r)
{
    T xd = x - o[0], yd = y - o[1];
    T ia2 = rcp_(o[2] * o[2]), ib2 = rcp_(o[3] * o[3]);
    T st, ct;
    sincos_(o[4], &st, &ct);
    T p = xd * ct + yd * st;
    T q = -xd * st + yd * ct;
    r.h = p * p * ia2 + q * q * ib2 - T(1);
    r.hx = T(2) * (ct * ia2 * p - st * ib2 * q);
    r.hy = T(2) * (st * ia2 * p + ct * ib2 * q);
    r.hth = T(0);
    r.hv = T(0);
    r.ht = T(-2) * (xd * ia2 * o[5] + yd * ib2 * o[6]);
}

// Ellipse2D whose per-obstacle constants are already evaluated (SCCBF_OBS_ELLIPSE_PREP):
// o = cx, cy, 1/a^2, 1/b^2, cos(theta_o), sin(theta_o), vx, vy.  Same formulas as ellipse_,
// without the sine / cosine / reciprocals of values that do not depend on the agent.
template <typename T>
__device__ inline void ellipse_prep_(T x, T y, const T o[8], Partials<T>& r)
{
    T xd = x - o[0], yd = y - o[1];
    T ia2 = o[2], ib2 = o[3], ct = o[4], st = o[5];
    T p = xd * ct + yd * st;
    T q = -xd * st + yd * ct;
    T pa = p * ia2, qb = q * ib2;
    r.h = p * pa + q * qb - T(1);
    r.hx = T(2) * (ct * pa - st * qb);
    r.hy = T(2) * (st * pa + ct * qb);
    r.hth = T(0);
    r.hv = T(0);
    r.ht = T(-2) * (xd * ia2 * o[6] + yd * ib2 * o[7]);
}

// Time-varying ellipse with speed margin, reference
// test_scripts/radial_dynamic_obstacles.py:391-405.
template <typename T>
__device__ inline void radial_(T x, T y, T v, const T o[8], Partials<T>& r)
{
    T xd = x - o[0], yd = y - o[1];
    T ia2 = rcp_(o[2] * o[2]), ib2 = rcp_(o[3] * o[3]);
    T iv = rcp_(T(1) + v);
    r.h = xd * xd * ia2 + yd * yd * ib2 - T(1) - o[6] * v * iv;
    r.hx = T(2) * xd * ia2;
    r.hy = T(2) * yd * ib2;
    r.hth = T(0);
    r.hv = -o[6] * iv * iv;
    r.ht = T(-2) * (xd * ia2 * o[4] + yd * ib2 * o[5]);
}

// CollisionCone2D, reference cbf/obstacles.py:468-502 (state), 401-458 (partials), from the
// relative position (xd, yd), the agent's velocity (svx, svy), the obstacle's (ovx, ovy), the cone
// radius a and (c, sn) = cos, sin of the agent's heading plus the cone's beta.
template <typename T>
__device__ inline void cone_core_(T xd, T yd, T svx, T svy, T ovx, T ovy, T a, T c, T sn, T eps,
                                  Partials<T>& r)
{
    T dvx = svx - ovx, dvy = svy - ovy;
    T d2 = xd * xd + yd * yd;
    T dist = sqrt_(d2);
    T vn = sqrt_(dvx * dvx + dvy * dvy);
    T cb = eps;
    if (dist > abs_(a)) cb = sqrt_(d2 - a * a) + eps;
    T cos_phi = (dist > eps) ? div_(cb, dist) : T(0);
    T icb = rcp_(cb + eps);
    T cbn = div_(cb, vn + eps);
    r.h = (xd * dvx + yd * dvy) + dist * vn * cos_phi;
    r.hx = dvx + vn * xd * icb;
    r.hy = dvy + vn * yd * icb;
    r.hv = (xd * c + yd * sn) + (dvx * c + dvy * sn) * cbn;
    r.hth = (-xd * svy + yd * svx) + (-dvx * svy + dvy * svx) * cbn;
    r.ht = (-dvx * ovx - dvy * ovy) - vn * (xd * ovx + yd * ovy) * icb;
}

// o = cx, cy, theta_o, v_o, a, beta (SCCBF_OBS_CONE)
template <typename T>
__device__ inline void cone_(T x, T y, T th, T v, T cth, T sth, const T o[8], T eps,
                             Partials<T>& r)
{
    T beta = o[5];
    T so, co;
    sincos_(o[2] + beta, &so, &co);
    T c = cth, sn = sth;
    if (__any(beta != T(0))) {  // scalar branch: without it the sine and cosine are evaluated for every cone and then discarded
        keep_branch_();
        if (beta != T(0)) sincos_(th + beta, &sn, &c);
    }
    cone_core_(x - o[0], y - o[1], v * cth, v * sth, o[3] * co, o[3] * so, o[4], c, sn, eps, r);
}

// o = cx, cy, v_o cos(theta_o + beta), v_o sin(theta_o + beta), a, beta (SCCBF_OBS_CONE_PREP):
// the obstacle's velocity is already resolved, beta only turns the direction of dh/dv
template <typename T>
__device__ inline void cone_prep_(T x, T y, T th, T v, T cth, T sth, const T o[8], T eps,
                                  Partials<T>& r)
{
    T beta = o[5];
    T c = cth, sn = sth;
    if (__any(beta != T(0))) {  // scalar branch, as in cone_
        keep_branch_();
        if (beta != T(0)) sincos_(th + beta, &sn, &c);
    }
    cone_core_(x - o[0], y - o[1], v * cth, v * sth, o[2], o[3], o[4], c, sn, eps, r);
}

// Box kinds: o = loc x, loc y, extent x, extent y, yaw, speed, buffer (BoundingBox, reference
// cbf/obstacles.py:59-64, and the buffer of from_bounding_box) -> the as-written slot of the obstacle that
// Ellipse2D.from_bounding_box (320-331) / CollisionCone2D.from_bounding_box (534-543) make of it, by the
// same arithmetic as k_ingest.
// radius of the cone around a box: both squares rounded, then their sum (no fused multiply-add, whose choice of
// the fused product could differ from one inlined copy to the next), the root, the buffer
template <typename T> __device__ inline T box_cone_radius_(T ex, T ey, T buffer)
{
#pragma clang fp contract(off)
    const T s2 = ex * ex + ey * ey;
    return sqrt_(s2) + buffer;
}

template <typename T> __device__ inline void box_ellipse_(const T o[8], T q[8])
{
    q[0] = o[0]; q[1] = o[1]; q[2] = o[2] + o[6]; q[3] = o[3] + o[6];
    q[4] = o[4]; q[5] = T(0); q[6] = T(0); q[7] = T(0);
}
template <typename T> __device__ inline void box_cone_(const T o[8], T q[8])  // theta_o = 0, beta = 0
{
    q[0] = o[0]; q[1] = o[1]; q[2] = T(0); q[3] = o[5];
    q[4] = box_cone_radius_(o[2], o[3], o[6]); q[5] = T(0); q[6] = T(0); q[7] = T(0);
}
// the same cone with its velocity (speed, 0) resolved: the prepared form, no sine / cosine
template <typename T> __device__ inline void box_cone_prep_(const T o[8], T q[8])
{
    q[0] = o[0]; q[1] = o[1]; q[2] = o[5]; q[3] = T(0);
    q[4] = box_cone_radius_(o[2], o[3], o[6]); q[5] = T(0); q[6] = T(0); q[7] = T(0);
}

template <typename T> __device__ inline void poly3_(const T c[4], T x, T& f, T& df, T& ddf)
{
    f = c[0] + x * (c[1] + x * (c[2] + x * c[3]));
    df = c[1] + x * (T(2) * c[2] + x * T(3) * c[3]);
    ddf = T(2) * c[2] + T(6) * c[3] * x;
}

// Abscissa of the curve point nearest to (px, py): descent from x0 = px, Newton with
// the true curvature, halving until the distance does not grow.  Replaces the scipy
// Newton-CG call of PolyLane.get_shortest_distance_x, reference cbf/obstacles.py:641-679.
// Kept out of line (one copy, rolled loops): lanes are a minority of the rows and the
// iteration would otherwise be replicated into every unrolled row of the kernels.
template <typename T>
__device__ __noinline__ T lane_closest_x_(T c0, T c1, T c2, T c3, T px, T py)
{
    const T c[4] = {c0, c1, c2, c3};
    T x = px;
#pragma nounroll
    for (int it = 0; it < kLaneIters; ++it) {
        T f, df, ddf;
        poly3_(c, x, f, df, ddf);
        T rr = f - py;
        T g0 = (x - px) * (x - px) + rr * rr;
        T g1 = T(2) * (x - px) + T(2) * rr * df;
        T g2 = T(2) * (T(1) + df * df + rr * ddf);
        T step = (g2 > T(1e-12)) ? div_(-g1, g2) : div_(-g1, abs_(g2) + T(1));
        if (step == T(0)) break;
        T xn = x + step;
        T glim = g0 * (T(1) + T(8) * Lim<T>::eps);
#pragma nounroll
        for (int k = 0; k < kLaneHalvings; ++k) {
            T fn, dfn, ddfn;
            poly3_(c, xn, fn, dfn, ddfn);
            T gn = (xn - px) * (xn - px) + (fn - py) * (fn - py);
            if (gn <= glim) break;
            step *= T(0.5);
            xn = x + step;
        }
        x = xn;
        if (abs_(step) <= Lim<T>::lane_xtol * (T(1) + abs_(x))) break;
    }
    return x;
}

// PolyLane, reference cbf/obstacles.py:620-636, 607-612, 681-689 (eta and the
// un-squared buffer as written).
template <typename T>
__device__ inline void lane_(T x, T y, const T o[8], T eps, Partials<T>& r)
{
    T cx = lane_closest_x_(o[0], o[1], o[2], o[3], x, y);
    T g, dg, ddg;
    const T c[4] = {o[0], o[1], o[2], o[3]};
    poly3_(c, cx, g, dg, ddg);
    T eta = T(1) + dg * ddg + dg * dg - y * ddg;
    if (abs_(eta) < eps) eta = eps;
    T ie = div_(T(2), eta);
    r.h = (cx - x) * (cx - x) + (g - y) * (g - y) - o[4];
    r.hx = ie * ((x - cx) * (eta - T(1)) - (y - g) * dg);
    r.hy = ie * (-(x - cx) * dg + (y - g) * (eta - dg * dg));
    r.hth = T(0);
    r.hv = T(0);
    r.ht = T(0);
}

// A lane polynomial of degree up to 6 (SCCBF_OBS_LANE6: o = c0 ... c6, buffer): PolyLane takes a coefficient array
// of any length, reference cbf/obstacles.py:578-593.  Its own out-of-line closest-point iteration, so that the cubic
// one above stays what it is; otherwise the same Newton with halving and the same formulas.
template <typename T> __device__ inline void poly6_(const T c[7], T x, T& f, T& df, T& ddf)
{
    f = c[0] + x * (c[1] + x * (c[2] + x * (c[3] + x * (c[4] + x * (c[5] + x * c[6])))));
    df = c[1] + x * (T(2) * c[2] + x * (T(3) * c[3] + x * (T(4) * c[4] + x * (T(5) * c[5] + x * (T(6) * c[6])))));
    ddf = T(2) * c[2] + x * (T(6) * c[3] + x * (T(12) * c[4] + x * (T(20) * c[5] + x * (T(30) * c[6]))));
}

// Two ways to run it.  Inlined (its loops stay rolled), in the kernels that keep up to four rows per lane: as a function of
// its own it needs 24 registers in fp32 and 66 in fp64 below the caller's live ones, which costs those kernels a wavefront
// per SIMD, while inlined it works on the slot's own registers.  Called, everywhere else: kernels that unroll sixteen rows
// per lane would carry sixteen copies of the loop and outgrow the instruction cache (the 128-row launch went from 59 to
// 306 us), and the kernels that keep the rows in LDS are not bound by registers.  The state carried around the loops is
// kept to x, the step and the distance at x.
template <typename T>
__device__ __forceinline__ T lane6_closest_x_body_(T c0, T c1, T c2, T c3, T c4, T c5, T c6, T px, T py)
{
    const T c[7] = {c0, c1, c2, c3, c4, c5, c6};
    T x = px;
#pragma nounroll
    for (int it = 0; it < kLaneIters; ++it) {
        T f, df, ddf;
        poly6_(c, x, f, df, ddf);
        const T rr = f - py;
        const T g0 = (x - px) * (x - px) + rr * rr;
        const T g1 = T(2) * (x - px) + T(2) * rr * df;
        const T g2 = T(2) * (T(1) + df * df + rr * ddf);
        T step = (g2 > T(1e-12)) ? div_(-g1, g2) : div_(-g1, abs_(g2) + T(1));
        if (step == T(0)) break;
#pragma nounroll
        for (int k = 0; k < kLaneHalvings; ++k) {
            const T xn = x + step;
            const T fn = c[0] + xn * (c[1] + xn * (c[2] + xn * (c[3] + xn * (c[4] + xn * (c[5] + xn * c[6])))));
            const T gn = (xn - px) * (xn - px) + (fn - py) * (fn - py);
            // a step already below the tolerance is taken as it is: the distances before and after it differ by less
            // than their own rounding, and halving it on that noise would leave x one step short
            if (gn <= g0 * (T(1) + T(8) * Lim<T>::eps) || abs_(step) <= Lim<T>::lane_xtol * (T(1) + abs_(x))) break;
            step *= T(0.5);
        }
        x += step;
        if (abs_(step) <= Lim<T>::lane_xtol * (T(1) + abs_(x))) break;
    }
    return x;
}

template <typename T>
__device__ __noinline__ T lane6_closest_x_call_(T c0, T c1, T c2, T c3, T c4, T c5, T c6, T px, T py)
{
    return lane6_closest_x_body_(c0, c1, c2, c3, c4, c5, c6, px, py);
}

template <bool INLINE6, typename T>
__device__ inline void lane6_(T x, T y, const T o[8], T eps, Partials<T>& r)
{
    T cx = INLINE6 ? lane6_closest_x_body_(o[0], o[1], o[2], o[3], o[4], o[5], o[6], x, y)
                   : lane6_closest_x_call_(o[0], o[1], o[2], o[3], o[4], o[5], o[6], x, y);
    T g, dg, ddg;
    const T c[7] = {o[0], o[1], o[2], o[3], o[4], o[5], o[6]};
    poly6_(c, cx, g, dg, ddg);
    T eta = T(1) + dg * ddg + dg * dg - y * ddg;
    if (abs_(eta) < eps) eta = eps;
    T ie = div_(T(2), eta);
    r.h = (cx - x) * (cx - x) + (g - y) * (g - y) - o[7];
    r.hx = ie * ((x - cx) * (eta - T(1)) - (y - g) * dg);
    r.hy = ie * (-(x - cx) * dg + (y - g) * (eta - dg * dg));
    r.hth = T(0);
    r.hv = T(0);
    r.ht = T(0);
}

// Per-agent quantities every row needs.
template <typename T> struct Ego {
    T x, y, th, v, cth, sth;
    T evx, evy;  // velocity v (cos, sin)(th)
    T vlr;       // v / lr
    T un0, un1;  // nominal input in QP coordinates
    __device__ void derive(const KArgs<T>& p)
    {
        evx = v * cth;
        evy = v * sth;
        vlr = v * p.inv_lr;
    }
};

// Where the two 4-vectors of slot m of agent a live, in units of 4-vectors:
//   planes (default)  obs[m][h][a]            index = a + m * 2A + h * A
//   tiled             obs[a/64][m][h][a%64]   index = (a/64) * 128 M + a%64 + m * 128 + h * 64
//   shared            obs[m][h]               index = m * 2 + h
// The strides are 32-bit (2 A < 2^32) so that base + m * slot is one 64-bit multiply-add.
struct ObsAddr {
    unsigned long long base;
    unsigned slot, half;
    __device__ unsigned long long at(int m) const { return base + (unsigned long long)(unsigned)m * slot; }
};
template <typename T> __device__ inline ObsAddr obs_addr_(const KArgs<T>& p, long long a)
{
    ObsAddr r;
    const bool shared = (p.flags & SCCBF_FLAG_OBS_SHARED) != 0;
    const bool tiled = (p.flags & SCCBF_FLAG_OBS_TILED) != 0;
    const bool by_agent = (p.flags & SCCBF_FLAG_OBS_AGENT_MAJOR) != 0;
    if (shared || by_agent) {  // the slots of one table, or of one agent, side by side
        r.base = by_agent ? (unsigned long long)a * (unsigned)(2 * p.M) : 0ull;
        r.slot = 2u;
        r.half = 1u;
        return r;
    }
    r.base = tiled ? (unsigned long long)(a / kTile) * (unsigned)(2 * kTile * p.M) + (unsigned)(a % kTile)
                   : (unsigned long long)a;
    r.slot = tiled ? 2u * kTile : 2u * (unsigned)p.A;
    r.half = tiled ? (unsigned)kTile : (unsigned)p.A;
    return r;
}

// Partials -> constraint row  a . u >= b  (u-space).
// DBM contraction: reference cbf/cbf.py:159-164, 200-207.  KBM: cbf/cbf.py:94-101.
template <typename T>
__device__ inline void contract_row_(const Partials<T>& d, const Ego<T>& e, const KArgs<T>& p,
                                     T& a0, T& a1, T& b, T& h)
{
    h = d.h;
    if (p.dynamics == SCCBF_DYN_KBM) {
        a0 = e.cth * d.hx + e.sth * d.hy;
        a1 = d.hth;
        b = -(p.alpha * d.h);
    } else {
        a0 = d.hv;
        a1 = (e.evx * d.hy - e.evy * d.hx) + e.vlr * d.hth;
        T lf_h = e.evx * d.hx + e.evy * d.hy;
        b = -(lf_h + p.alpha * d.h + d.ht);
    }
}

// Barrier value and partials of one slot whose kind is known; false for an empty slot.
// INLINE6: the closest-point loop of a degree-6 lane inlined (callers that unroll at most four rows) or called.
template <bool INLINE6 = false, typename T>
__device__ inline bool slot_partials_(int type, const T o_in[8], const Ego<T>& e, const KArgs<T>& p, Partials<T>& d)
{
    T o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = o_in[i];
    // A wavefront without a box among its slots skips the conversion as a whole (a scalar branch): the lanes of a
    // masked branch would pay a register copy per slot value for the join with the converted ones.
    if (__any(type == SCCBF_OBS_ELLIPSE_BOX || type == SCCBF_OBS_CONE_BOX)) {
        keep_branch_();
        if (type == SCCBF_OBS_ELLIPSE_BOX) {         // the obstacle the box stands for: the ellipse as written,
            box_ellipse_(o_in, o);
            type = SCCBF_OBS_ELLIPSE;
        } else if (type == SCCBF_OBS_CONE_BOX) {     // the cone with its velocity (speed, 0) resolved
            box_cone_prep_(o_in, o);
            type = SCCBF_OBS_CONE_PREP;
        }
    }
    switch (type) {
    case SCCBF_OBS_ELLIPSE: ellipse_(e.x, e.y, o, d); return true;
    case SCCBF_OBS_CONE: cone_(e.x, e.y, e.th, e.v, e.cth, e.sth, o, p.zero_tol, d); return true;
    case SCCBF_OBS_LANE: lane_(e.x, e.y, o, p.zero_tol, d); return true;
#ifndef SCCBF_NO_LANE6
    case SCCBF_OBS_LANE6: lane6_<INLINE6>(e.x, e.y, o, p.zero_tol, d); return true;
#endif
    case SCCBF_OBS_RADIAL: radial_(e.x, e.y, e.v, o, d); return true;
    case SCCBF_OBS_ELLIPSE_PREP: ellipse_prep_(e.x, e.y, o, d); return true;
    case SCCBF_OBS_CONE_PREP: cone_prep_(e.x, e.y, e.th, e.v, e.cth, e.sth, o, p.zero_tol, d); return true;
    default: return false;
    }
}

// One obstacle slot -> constraint row  a . u >= b  (u-space) and barrier value h.
template <bool INLINE6 = false, typename T>
__device__ inline void slot_row_(int type, const T o[8], const Ego<T>& e, const KArgs<T>& p,
                                 T& a0, T& a1, T& b, T& h)
{
    Partials<T> d;
    if (!slot_partials_<INLINE6>(type, o, e, p, d)) {
        a0 = T(0); a1 = T(0); b = T(-1); h = T(1);
        return;
    }
    contract_row_(d, e, p, a0, a1, b, h);
}

// u-space row -> unit-normal row in w-space; zero rows keep at = 0.  Straight-line code.
template <typename T>
__device__ inline void to_wspace_(T a0, T a1, T b, const Ego<T>& e, const KArgs<T>& p,
                                  T& at0, T& at1, T& bt)
{
    T t0 = a0 * p.inv_l00;
    T t1 = (a1 - p.l10 * t0) * p.inv_l11;
    T rhs = b - (a0 * e.un0 + a1 * e.un1);
    T n2 = t0 * t0 + t1 * t1;
    const bool nz = n2 > Lim<T>::tiny_n2;
    T inv = nz ? rsqrt_(n2) : T(0);
    // 0 . u >= rhs holds for every u or for none: a positive rhs must win every comparison of
    // violations, whatever tolerance the size of w has earned the other rows by then
    T zb = rhs > Lim<T>::tiny_n2 ? T(INFINITY) : rhs;
    at0 = nz ? t0 * inv : T(0);
    at1 = nz ? t1 * inv : T(0);
    bt = nz ? rhs * inv : zb;
}

// Ego quantities from the already loaded state / nominal-input vectors.
template <typename T>
__device__ inline void make_ego_(const KArgs<T>& p, const typename VT<T>::v4& s,
                                 const typename VT<T>::v2& ur, bool qp_only, Ego<T>& e,
                                 T& ur0, T& ur1, int& bad)
{
    ur0 = ur.x;
    ur1 = ur.y;
    bad = !(finite_(ur0) && finite_(ur1));
    if (qp_only) {
        e.x = e.y = e.th = e.v = T(0);
        e.cth = T(1);
        e.sth = T(0);
        e.evx = e.evy = e.vlr = T(0);
        e.un0 = ur0;
        e.un1 = ur1;
        return;
    }
    e.x = s.x; e.y = s.y; e.th = s.z; e.v = s.w;
    bad |= !(finite_(s.x) && finite_(s.y) && finite_(s.z) && finite_(s.w));
    sincos_(e.th, &e.sth, &e.cth);
    e.derive(p);
    e.un0 = ur0;
    if (p.flags & SCCBF_FLAG_SLIP_IO)
        e.un1 = ur1;
    else if (p.dynamics == SCCBF_DYN_KBM)
        e.un1 = ur0 * tan_(ur1) * p.inv_L;                       // reference cbf/cbf.py:75
    else
        e.un1 = atan_(tan_(ur1) * p.k_in);   // atan2(lr tan(delta), lf + lr), lf + lr > 0: reference cbf/cbf.py:175
}

template <typename T>
__device__ inline void load_ego_(const KArgs<T>& p, long long qp, bool qp_only, Ego<T>& e,
                                 T& ur0, T& ur1, int& bad)
{
    using v4 = typename VT<T>::v4;
    using v2 = typename VT<T>::v2;
    const v2 ur = reinterpret_cast<const v2*>(p.u_ref)[qp];
    v4 s = v4{T(0), T(0), T(0), T(0)};
    if (!qp_only) s = reinterpret_cast<const v4*>(p.state)[qp];
    make_ego_(p, s, ur, qp_only, e, ur0, ur1, bad);
}

// The filter's answer in the caller's coordinates from the optimum w of the QP.
template <typename T>
__device__ inline typename VT<T>::v2 result_u_(const KArgs<T>& p, bool qp_only, int st, T w0, T w1,
                                               const Ego<T>& e, T ur0, T ur1)
{
    using v2 = typename VT<T>::v2;
    v2 out;
    if (st != SCCBF_STATUS_OPTIMAL) {
        out.x = ur0;
        out.y = ur1;
    } else {
        // u = u_nom + Lc^-T w
        T y1 = w1 * p.inv_l11;
        T y0 = (w0 - p.l10 * y1) * p.inv_l00;
        T u0 = e.un0 + y0, u1 = e.un1 + y1;
        out.x = u0;
        if (qp_only || (p.flags & SCCBF_FLAG_SLIP_IO))
            out.y = u1;
        else if (p.dynamics == SCCBF_DYN_KBM)
            out.y = atan2_(u1 * p.L, ur0);                    // reference cbf/cbf.py:109
        else
            out.y = atan_(tan_(u1) * p.k_out);  // atan2((lf + lr) tan(beta), lr), lr > 0: reference cbf/cbf.py:216
    }
    return out;
}

// ------------------------------------------------------- vehicle step (update_com)
// State.update_com, reference test_scripts/stanley_controller_ellipse.py:122-131; returns beta.
template <typename T>
__device__ inline T vehicle_step_(typename VT<T>::v4& s, T acc, T delta_in, T k_in, T inv_lr, T max_steer, T dt)
{
    const T delta = min_(max_(delta_in, -max_steer), max_steer);
    const T beta = atan_(tan_(delta) * k_in);  // atan2(lr tan(delta), lf + lr), lf + lr > 0
    T sth, cth;
    sincos_(s.z, &sth, &cth);
    const T v = s.w;
    s.x += (v * cth - v * sth * beta) * dt;
    s.y += (v * sth + v * cth * beta) * dt;
    s.z += (v * beta * inv_lr) * dt;
    s.w += acc * dt;
    return beta;
}

// The kinematic steps of the drivers' State class, reference test_scripts/stanley_controller_ellipse.py:
// update (86-100): u = (a, delta); update_by_vel (102-118): u = (v, delta), the speed is set, not
// integrated; both normalise the yaw (cbf/utils.py:93-106).  SCCBF_STEP_UNICYCLE, u = (a, omega): the
// Euler step of the model DUM_CBF_2DS filters (cbf/cbf.py:237-245); the reference ships no driver for it.
template <typename T> __device__ inline T wrap_angle_(T a)
{
    constexpr T kPi = T(3.14159265358979323846), k2Pi = T(6.28318530717958647692);
    a -= k2Pi * rint(a * (T(1) / k2Pi));
    a = a > kPi ? a - k2Pi : a;
    a = a < -kPi ? a + k2Pi : a;
    return a;
}
template <typename T>
__device__ inline void kin_step_(typename VT<T>::v4& s, int mode, T u0, T u1, T inv_L, T max_steer, T dt)
{
    T sth, cth;
    sincos_(s.z, &sth, &cth);
    const T v = s.w;
    s.x += v * cth * dt;
    s.y += v * sth * dt;
    if (mode == SCCBF_STEP_UNICYCLE) {
        s.z += u1 * dt;
        s.w += u0 * dt;
    } else {
        const T delta = min_(max_(u1, -max_steer), max_steer);
        s.z = wrap_angle_(s.z + v * inv_L * tan_(delta) * dt);
        s.w = mode == SCCBF_STEP_KIN_VEL ? u0 : s.w + u0 * dt;
    }
}

// The lane that holds the answer stores it and, where the caller asked for it (sccbf_solve_step_*),
// moves the agent by one vehicle step with the filtered input.
template <typename T>
__device__ inline void store_result_(const KArgs<T>& p, long long qp, bool qp_only, int st,
                                     T w0, T w1, const Ego<T>& e, T ur0, T ur1)
{
    using v2 = typename VT<T>::v2;
    using v4 = typename VT<T>::v4;
    const v2 u = result_u_(p, qp_only, st, w0, w1, e, ur0, ur1);
    reinterpret_cast<v2*>(p.u_out)[qp] = u;
    p.status[qp] = st;
    if (p.step_state) {
        v4 s = v4{e.x, e.y, e.th, e.v};
        const T beta = vehicle_step_<T>(s, u.x, u.y, p.k_in, p.inv_lr, p.step_max_steer, p.step_dt);
        reinterpret_cast<v4*>(p.step_state)[qp] = s;
        if (p.step_beta) p.step_beta[qp] = beta;
    }
}

}  // namespace
