// sccbf_qp.hip.h - the 2-input QP: pivot of the active-set iteration
// Part of the sccbf library (one translation unit, see sccbf.hip); device code for gfx950.
#pragma once
#include "sccbf_common.hip.h"

namespace {

// ------------------------------------------------------------------- the 2-D QP
// State of the active-set iteration, identical in every lane of a cooperative group:
// the current optimum w and two stored rows: the rows active at w, or rows that were
// active earlier and are satisfied at w (keeping a satisfied row of the problem among the
// stored ones changes no optimum), or the null row 0 . w >= -1 of an unused slot.
template <typename T> struct Pivot {
    T w0, w1;
    T sa0[2], sa1[2], sb[2];
    int st;
    __device__ void reset(int status)
    {
        w0 = w1 = T(0);
        sa0[0] = sa0[1] = sa1[0] = sa1[1] = T(0);
        sb[0] = sb[1] = T(-1);
        st = status;
    }
};

// Move the optimum onto the violated row c . w = cb (unit normal c): the new optimum of
// {stored rows} + {this row} lies on that line; along it, w = p + t d, the stored rows
// bound t and the point nearest to the origin inside the bounds is taken.  The row that
// clamps t (if any) is kept with the new one.  Straight-line code: every lane evaluates it
// and `go` selects which lanes commit the result.
// NS: how many stored slots can hold a real row, known at compile time (the first step of
// every QP has none, the second at most one); the arithmetic of the other slots vanishes.
template <int NS = 2, typename T> __device__ inline void pivot_onto_(Pivot<T>& s, T c0, T c1, T cb, bool go)
{
    constexpr T kInf = INFINITY;
    const bool zero_row = (c0 == T(0)) & (c1 == T(0));  // 0 . u >= positive
    const T p0 = c0 * cb, p1 = c1 * cb;
    const T d0 = -c1, d1 = c0;
    T tlo = -kInf, thi = kInf;
    bool l1 = false, h1 = false, inf = false;
    if (NS > 0) {
        const T cs0 = s.sa1[0] * d1 + s.sa0[0] * d0;
        const T ee0 = s.sb[0] - s.sa0[0] * p0 - s.sa1[0] * p1;
        const bool np0 = abs_(cs0) > Lim<T>::par_tol;  // a null row is "parallel" and never binds
        const T t0 = div_(ee0, cs0);                   // garbage for parallel rows
        const bool up0 = cs0 > T(0);
        tlo = (np0 & up0) ? t0 : -kInf;
        thi = (np0 & !up0) ? t0 : kInf;
        // parallel rows facing each other with a gap: no common point
        inf = !np0 & (ee0 > Lim<T>::feas_tol * (abs_(s.sb[0]) + abs_(cb)) + Lim<T>::tiny_n2);
    }
    if (NS > 1) {
        const T cs1 = s.sa1[1] * d1 + s.sa0[1] * d0;
        const T ee1 = s.sb[1] - s.sa0[1] * p0 - s.sa1[1] * p1;
        const bool np1 = abs_(cs1) > Lim<T>::par_tol;
        const T t1 = div_(ee1, cs1);
        const bool up1 = cs1 > T(0);
        const T tl1 = (np1 & up1) ? t1 : -kInf;
        const T th1 = (np1 & !up1) ? t1 : kInf;
        l1 = tl1 > tlo;
        h1 = th1 < thi;
        tlo = l1 ? tl1 : tlo;
        thi = h1 ? th1 : thi;
        inf |= !np1 & (ee1 > Lim<T>::feas_tol * (abs_(s.sb[1]) + abs_(cb)) + Lim<T>::tiny_n2);
        // bounds that cross: three rows without a common point
        inf |= tlo > thi + Lim<T>::feas_tol * (abs_(tlo) + abs_(thi));
    }
    const bool bad = zero_row | inf;
    const bool uselo = tlo > T(0);
    const bool usehi = !uselo & (thi < T(0));
    const T t = uselo ? tlo : (usehi ? thi : T(0));
    // the clamping row must end up in slot 1; it sits there already when keep1
    const bool keep1 = (uselo & l1) | (usehi & h1);
    const bool upd = go & !bad;
    const bool mv = upd & !keep1;  // slot 0 (clamping, or merely satisfied) moves to slot 1
    s.w0 = upd ? p0 + t * d0 : s.w0;
    s.w1 = upd ? p1 + t * d1 : s.w1;
    if (NS > 0) {
        s.sa0[1] = mv ? s.sa0[0] : s.sa0[1];
        s.sa1[1] = mv ? s.sa1[0] : s.sa1[1];
        s.sb[1] = mv ? s.sb[0] : s.sb[1];
    }
    s.sa0[0] = upd ? c0 : s.sa0[0];
    s.sa1[0] = upd ? c1 : s.sa1[0];
    s.sb[0] = upd ? cb : s.sb[0];
    s.st = (go & bad) ? SCCBF_STATUS_INFEASIBLE : s.st;
}

}  // namespace
