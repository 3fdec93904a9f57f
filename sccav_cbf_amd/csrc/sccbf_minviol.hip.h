// sccbf_minviol.hip.h - the least-violation input for agents whose rows admit no input at all
// Part of the sccbf library (one translation unit, see sccbf.hip); device code for gfx950.
//
// The reference hands cvxopt an infeasible program and applies whatever comes back without reading the status
// (cbf/cbf.py:213-220).  The filter kernels report such agents (status 1) and pass the nominal input through.
// With SCCBF_FLAG_INFEASIBLE_MIN_VIOLATION this second pass, one thread per such agent on the rows the filter
// formed (or was given), replaces that by the input whose largest row violation is least, the violation of a row
// being the distance to its half-plane in the metric of the cost (unit normals in w = Lc^T (u - u_nom)); among those
// inputs the one nearest to the nominal input.  Found with the filter's own pivot: solve the rows relaxed by t; when
// that fails, the pivot names the rows without a common point (an antiparallel pair, or three rows whose normals span
// the plane positively); the least worst-case violation of those alone, (bt_i + bt_j) / 2 or sum D_i bt_i / sum D_i with
// D the cross products of the other two normals, is a lower bound of the whole problem's and the next t.  t grows
// strictly, no certificate repeats, and the first t that is feasible is the least one (to the pivot's tolerances).
#pragma once
#include "sccbf_filter.hip.h"

namespace {

constexpr int kMinViolOuter = 48;  // certificates tried per agent before it keeps its nominal input

template <typename T> struct MvLim;
template <> struct MvLim<float> { static constexpr float slack = 4e-6f; };
template <> struct MvLim<double> { static constexpr double slack = 1e-12; };

// least worst-case violation of the rows the pivot has just found without a common point: c (the row it was moving
// onto) and the rows stored in s, all relaxed by the same t, so the value is the amount by which t has to grow
template <typename T> __device__ inline T certificate_value_(T c0, T c1, T cb, const Pivot<T>& s)
{
    T v = T(0);
    const bool r0 = (s.sa0[0] != T(0)) | (s.sa1[0] != T(0)), r1 = (s.sa0[1] != T(0)) | (s.sa1[1] != T(0));
    const T x0 = c0 * s.sa1[0] - c1 * s.sa0[0], d0 = c0 * s.sa0[0] + c1 * s.sa1[0];   // c x s0, c . s0
    const T x1 = c0 * s.sa1[1] - c1 * s.sa0[1], d1 = c0 * s.sa0[1] + c1 * s.sa1[1];   // c x s1, c . s1
    if (r0 && abs_(x0) <= Lim<T>::par_tol && d0 < T(0)) v = max_(v, T(0.5) * (cb + s.sb[0]));
    if (r1 && abs_(x1) <= Lim<T>::par_tol && d1 < T(0)) v = max_(v, T(0.5) * (cb + s.sb[1]));
    if (r0 && r1) {
        const T dc = s.sa0[0] * s.sa1[1] - s.sa1[0] * s.sa0[1];   // s0 x s1, the weight of c
        const T ds0 = -x1;                                        // s1 x c, the weight of s0
        const T ds1 = x0;                                         // c x s0, the weight of s1
        const T lim = Lim<T>::par_tol;
        const bool pos = (dc > lim) & (ds0 > lim) & (ds1 > lim), neg = (dc < -lim) & (ds0 < -lim) & (ds1 < -lim);
        if (pos | neg) v = max_(v, div_(dc * cb + ds0 * s.sb[0] + ds1 * s.sb[1], dc + ds0 + ds1));
    }
    return v;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_min_violation(const KArgs<T> p, const T* rows, int no_transform)
{
    using v4 = typename VT<T>::v4;
    using v2 = typename VT<T>::v2;
    const long long a = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (a >= p.A) return;
    if (p.status[a] != SCCBF_STATUS_INFEASIBLE) return;
    Ego<T> e;
    T ur0, ur1;
    int bad;
    load_ego_(p, a, no_transform == 2, e, ur0, ur1, bad);
    if (no_transform) {  // the nominal input is in QP coordinates already (unicycle model, slip coordinates, QP stage)
        e.un0 = ur0;
        e.un1 = ur1;
    }
    const v4* r4 = reinterpret_cast<const v4*>(rows);
    const int M = p.M;
    const int cap = kCapPerRow * M + kCapBase;
    T t = T(0);
#pragma nounroll
    for (int outer = 0; outer < kMinViolOuter; ++outer) {
        Pivot<T> s;
        s.reset(SCCBF_STATUS_OPTIMAL);
        T c0 = T(0), c1 = T(0), cb = T(0);
        bool solved = false;
#pragma nounroll
        for (int it = 0; it <= cap; ++it) {
            const T thr = Lim<T>::viol_tol * (abs_(s.w0) + abs_(s.w1) + abs_(t)) + Lim<T>::tiny_n2;
            T worst = thr;
            bool any = false;
#pragma nounroll
            for (int m = 0; m < M; ++m) {
                const v4 r = r4[(long long)m * p.A + a];
                T at0, at1, bt;
                to_wspace_(r.x, r.y, r.z, e, p, at0, at1, bt);
                if ((at0 == T(0)) & (at1 == T(0))) continue;   // no input helps such a row: left out
                const T viol = (bt - t) - at0 * s.w0 - at1 * s.w1;
                if (viol > worst) {
                    worst = viol; c0 = at0; c1 = at1; cb = bt - t; any = true;
                }
            }
            if (!any) {
                solved = true;
                break;
            }
            pivot_onto_<2>(s, c0, c1, cb, true);
            if (s.st != SCCBF_STATUS_OPTIMAL) break;
        }
        if (solved) {
            const v2 u = result_u_(p, no_transform != 0, SCCBF_STATUS_OPTIMAL, s.w0, s.w1, e, ur0, ur1);
            if (finite_(u.x) && finite_(u.y)) reinterpret_cast<v2*>(p.u_out)[a] = u;
            return;
        }
        if (s.st == SCCBF_STATUS_OPTIMAL) return;             // ran out of steps: the nominal input stays
        const T v = certificate_value_(c0, c1, cb, s);
        if (!(v > T(0))) return;                              // no usable certificate: the nominal input stays
        t = (t + v) * (T(1) + MvLim<T>::slack) + Lim<T>::tiny_n2;
    }
}

}  // namespace
