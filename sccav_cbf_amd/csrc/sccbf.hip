// sccbf.hip - batched CBF-QP safety filter for gfx950 (MI355X): HIP kernels + C ABI.
//
// One control tick per agent: barrier value and partials of every obstacle slot,
// Lie derivatives -> constraint rows, then the 2-input QP.  Rows never touch HBM.
//
// Two kernel families share the row and QP arithmetic:
//   k_group<T, G, RPL>  G cooperating lanes per QP (G = 1, 2, 4, 8, 16), each lane
//                       keeps RPL rows in registers; cross-lane steps are DPP moves
//                       inside the G-lane group, the winning row of a step comes back
//                       from an LDS copy of the rows by index.  G = 1 is one thread per QP.
//   k_wave<T, WPB>      one 64-lane wavefront per QP, rows staged in LDS, any M;
//                       the most violated row is found by a wave-wide max.
//
// The QP (reference cbf/cbf.py:182-213 hands it to cvxopt.solvers.cp) is solved
// exactly: in w = Lc^T (u - u_nom), R = Lc Lc^T, it is the least-norm point of the
// polygon { at_i . w >= bt_i }.  An active-set iteration starts at w = 0 with no
// active row; while some row is violated, the most violated one is taken: the optimum
// of {stored rows} + {that row} lies on its line, so the 1-D problem along the line,
// bounded by the two stored rows, is solved; the new row and the row that clamps the
// 1-D optimum (else a row that is merely satisfied there) are stored for the next step.
// |w| grows strictly with every step, so no active set repeats; the loop ends with no
// violated row (optimal) or an empty 1-D interval (three rows without a common point:
// infeasible).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cmath>
#include <type_traits>

#include "sccbf.h"

namespace {

#ifndef SCCBF_BLOCK
#define SCCBF_BLOCK 256  // threads per workgroup of the group / passthrough / partials kernels
#endif
constexpr int kBlock = SCCBF_BLOCK;
#ifndef SCCBF_TILE
#define SCCBF_TILE 64  // agents per tile of the tiled obstacle layout
#endif
constexpr int kTile = SCCBF_TILE;
#ifndef SCCBF_ROLLOUT_WAVES
#define SCCBF_ROLLOUT_WAVES 4
#endif
#ifndef SCCBF_CAP_PER_ROW
#define SCCBF_CAP_PER_ROW 8
#define SCCBF_CAP_BASE 32
#endif
// active-set steps allowed per QP: kCapPerRow * M + kCapBase (a QP that needs more is reported infeasible)
constexpr int kCapPerRow = SCCBF_CAP_PER_ROW, kCapBase = SCCBF_CAP_BASE;
constexpr int kLaneIters = 32;     // Newton iterations for the lane closest point
constexpr int kLaneHalvings = 30;  // step halvings per iteration

template <typename T> struct VT;
template <> struct VT<float> {
    using v4 = float4;
    using v2 = float2;
};
template <> struct VT<double> {
    using v4 = double4;
    using v2 = double2;
};

template <typename T> __device__ inline typename VT<T>::v4 nt_load4_(const typename VT<T>::v4* ptr)
{
    typedef T raw4 __attribute__((ext_vector_type(4)));
    const raw4 r = __builtin_nontemporal_load(reinterpret_cast<const raw4*>(ptr));
    return typename VT<T>::v4{r.x, r.y, r.z, r.w};
}

template <typename T> struct KArgs {
    T alpha, lr, lf, L, inv_lr, inv_L;
    T k_in, k_out;  // lr / (lf + lr) and its inverse: the steering <-> slip maps
    T l10, inv_l00, inv_l11;  // Cholesky factor of R
    T zero_tol;
    int dynamics;
    unsigned flags;
    long long A;
    int M;
    const T* state;
    const T* u_ref;
    const T* obs;   // obstacle slots, or rows (M,A,4) for the QP-only entry
    const uint8_t* obs_type;
    T* u_out;
    int* status;
    T* rows_out;
};

// ------------------------------------------------------------------ math shims
// fp32: the hardware sin/cos unit (argument in revolutions, reduced to [0, 1) so that any
// magnitude is in its range; the angle error is |x| 2^-24, 2e-7 rad for a heading in
// [-pi, pi]); 1-ulp rcp / sqrt / rsq instead of the IEEE division and denormal-safe sqrt
// sequences.  Library sincosf/tanf would pull a Payne-Hanek reduction with a scratch
// table into every call site.  fp64 keeps the full-accuracy library forms.
__device__ inline void sincos_(float x, float* s, float* c)
{
    const float r = __builtin_amdgcn_fractf(x * 0.15915494309189535f);
    *s = __builtin_amdgcn_sinf(r);
    *c = __builtin_amdgcn_cosf(r);
}
__device__ inline void sincos_(double x, double* s, double* c) { sincos(x, s, c); }
__device__ inline float sqrt_(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ inline double sqrt_(double x) { return sqrt(x); }
__device__ inline float rsqrt_(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ inline double rsqrt_(double x) { return 1.0 / sqrt(x); }
__device__ inline float rcp_(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ inline double rcp_(double x) { return 1.0 / x; }
__device__ inline float div_(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ inline double div_(double a, double b) { return a / b; }
__device__ inline float tan_(float x)
{
    float s, c;
    sincos_(x, &s, &c);
    return s * __builtin_amdgcn_rcpf(c);
}
__device__ inline double tan_(double x) { return tan(x); }
// atan for fp32: t P(t^2) on [0, 1] (Chebyshev fit, 1.1e-7 absolute), 1/t reflection above 1
__device__ inline float atan_(float z)
{
    const float az = fabsf(z);
    const bool big = az > 1.0f;
    const float t = big ? __builtin_amdgcn_rcpf(az) : az;
    const float s = t * t;
    float q = 0.0027662834618240595f;
    q = fmaf(q, s, -0.015731249004602432f);
    q = fmaf(q, s, 0.04213762283325195f);
    q = fmaf(q, s, -0.07456854730844498f);
    q = fmaf(q, s, 0.10618370771408081f);
    q = fmaf(q, s, -0.14197798073291779f);
    q = fmaf(q, s, 0.1999187171459198f);
    q = fmaf(q, s, -0.333330363035202f);
    q = fmaf(q, s, 1.0f);
    float r = t * q;
    r = big ? 1.5707963267948966f - r : r;
    return copysignf(r, z);
}
__device__ inline double atan_(double z) { return atan(z); }
__device__ inline float atan2_(float y, float x) { return atan2f(y, x); }
__device__ inline double atan2_(double y, double x) { return atan2(y, x); }
__device__ inline float abs_(float x) { return fabsf(x); }
__device__ inline double abs_(double x) { return fabs(x); }
__device__ inline float max_(float a, float b) { return fmaxf(a, b); }
__device__ inline double max_(double a, double b) { return fmax(a, b); }
__device__ inline float min_(float a, float b) { return fminf(a, b); }
__device__ inline double min_(double a, double b) { return fmin(a, b); }
__device__ inline bool finite_(float x) { return isfinite(x); }
__device__ inline bool finite_(double x) { return isfinite(x); }

template <typename T> struct Lim;
template <> struct Lim<float> {
    static constexpr float eps = 1.1920929e-7f;
    static constexpr float tiny_n2 = 1e-30f;   // squared row norm below this = zero row
    static constexpr float viol_tol = 2e-6f;   // relative slack of the "violated" test
    static constexpr float par_tol = 1e-6f;    // |sin| below this = parallel rows
    static constexpr float feas_tol = 1e-6f;   // relative slack of the "bounds cross" test
    static constexpr float lane_xtol = 3e-4f;
};
template <> struct Lim<double> {
    static constexpr double eps = 2.220446049250313e-16;
    static constexpr double tiny_n2 = 1e-300;
    static constexpr double viol_tol = 1e-12;
    static constexpr double par_tol = 1e-14;
    static constexpr double feas_tol = 1e-10;
    static constexpr double lane_xtol = 1e-8;
};

// ------------------------------------------------------------- cross-lane helpers
// Reductions over the G consecutive lanes of a cooperative group (G power of two <= 64).
// Inside a 16-lane row the partner exchanges are DPP moves (no LDS round trip):
// quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror; after each step
// every lane of the growing sub-group holds that sub-group's result.  Across rows
// (G = 64 only) the partners come through ds_bpermute.
template <int CTRL> __device__ inline unsigned dpp_(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
struct OpMax { __device__ unsigned operator()(unsigned a, unsigned b) const { return a > b ? a : b; } };
struct OpOr { __device__ unsigned operator()(unsigned a, unsigned b) const { return a | b; } };

template <int G, typename Op> __device__ inline unsigned group_reduce(unsigned v, Op op)
{
    if (G >= 2) v = op(v, dpp_<0xB1>(v));
    if (G >= 4) v = op(v, dpp_<0x4E>(v));
    if (G >= 8) v = op(v, dpp_<0x141>(v));
    if (G >= 16) v = op(v, dpp_<0x140>(v));
    if (G >= 32) v = op(v, (unsigned)__shfl_xor((int)v, 16, 64));
    if (G >= 64) v = op(v, (unsigned)__shfl_xor((int)v, 32, 64));
    return v;
}
template <int G> __device__ inline unsigned group_umax(unsigned v) { return group_reduce<G>(v, OpMax()); }
template <int G> __device__ inline unsigned group_or(unsigned v) { return group_reduce<G>(v, OpOr()); }
struct OpIMax { __device__ unsigned operator()(unsigned a, unsigned b) const { return (int)a > (int)b ? a : b; } };
template <int G> __device__ inline int group_imax(int v) { return (int)group_reduce<G>((unsigned)v, OpIMax()); }
// value that is non-zero in at most one lane of the group -> all lanes
template <int G> __device__ inline float group_bcast(float v) { return __uint_as_float(group_or<G>(__float_as_uint(v))); }
template <int G> __device__ inline double group_bcast(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = group_or<G>((unsigned)b), hi = group_or<G>((unsigned)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// value held by exactly one lane of the group (the others pass `mine = false`) -> all lanes
template <int G> __device__ inline float group_pick(float v, bool mine)
{
    return __uint_as_float(group_or<G>(mine ? __float_as_uint(v) : 0u));
}
template <int G> __device__ inline double group_pick(double v, bool mine)
{
    const unsigned long long b = mine ? (unsigned long long)__double_as_longlong(v) : 0ull;
    const unsigned lo = group_or<G>((unsigned)b), hi = group_or<G>((unsigned)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// 32-bit image of a violation whose order as a signed integer is the order of the
// violations that exceed thr; everything at or below thr maps below every one of those.
// fp32: the bits themselves (negative floats are negative integers; the comparison with
// thr happens once, on the group's maximum).  fp64: compared here at full precision, then
// narrowed (clamped into the normal fp32 range so the image stays positive and ordered).
__device__ inline unsigned viol_bits(float v, float) { return __float_as_uint(v); }
__device__ inline unsigned viol_bits(double v, double thr)
{
    return v > thr ? __float_as_uint(fminf(fmaxf((float)v, 1.17549435e-38f), 3.0e38f)) : 0x80000000u;
}
// order-preserving image of a violation already known to be positive
__device__ inline unsigned viol_bits_pos(float v) { return __float_as_uint(v); }
__device__ inline unsigned viol_bits_pos(double v) { return __float_as_uint(fmaxf((float)v, 1.17549435e-38f)); }
// does the winning key stand for a violation above thr?
__device__ inline bool viol_key_above(int key, float thr) { return __uint_as_float((unsigned)key & ~0xFFu) > thr; }
__device__ inline bool viol_key_above(int key, double) { return key > 0xFF; }

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

// CollisionCone2D, reference cbf/obstacles.py:468-502 (state), 401-458 (partials).
template <typename T>
__device__ inline void cone_(T x, T y, T th, T v, T cth, T sth, const T o[8], T eps,
                             Partials<T>& r)
{
    T xd = x - o[0], yd = y - o[1];
    T beta = o[5], a = o[4];
    T svx = v * cth, svy = v * sth;
    T so, co;
    sincos_(o[2] + beta, &so, &co);
    T ovx = o[3] * co, ovy = o[3] * so;
    T dvx = svx - ovx, dvy = svy - ovy;
    T d2 = xd * xd + yd * yd;
    T dist = sqrt_(d2);
    T vn = sqrt_(dvx * dvx + dvy * dvy);
    T cb = eps;
    if (dist > abs_(a)) cb = sqrt_(d2 - a * a) + eps;
    T cos_phi = (dist > eps) ? div_(cb, dist) : T(0);
    T c = cth, sn = sth;
    if (beta != T(0)) sincos_(th + beta, &sn, &c);
    T icb = rcp_(cb + eps);
    T cbn = div_(cb, vn + eps);
    r.h = (xd * dvx + yd * dvy) + dist * vn * cos_phi;
    r.hx = dvx + vn * xd * icb;
    r.hy = dvy + vn * yd * icb;
    r.hv = (xd * c + yd * sn) + (dvx * c + dvy * sn) * cbn;
    r.hth = (-xd * svy + yd * svx) + (-dvx * svy + dvy * svx) * cbn;
    r.ht = (-dvx * ovx - dvy * ovy) - vn * (xd * ovx + yd * ovy) * icb;
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

// Per-agent quantities every row needs.
template <typename T> struct Ego {
    T x, y, th, v, cth, sth;
    T un0, un1;  // nominal input in QP coordinates
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
    r.base = shared ? 0ull : (tiled ? (unsigned long long)(a / kTile) * (unsigned)(2 * kTile * p.M) + (unsigned)(a % kTile)
                                    : (unsigned long long)a);
    r.slot = shared ? 2u : (tiled ? 2u * kTile : 2u * (unsigned)p.A);
    r.half = shared ? 1u : (tiled ? (unsigned)kTile : (unsigned)p.A);
    return r;
}

// One obstacle slot -> constraint row  a . u >= b  (u-space) and barrier value h.
// DBM contraction: reference cbf/cbf.py:159-164, 200-207.  KBM: cbf/cbf.py:94-101.
template <typename T>
__device__ inline void slot_row_(int type, const T o[8], const Ego<T>& e, const KArgs<T>& p,
                                 T& a0, T& a1, T& b, T& h)
{
    Partials<T> d;
    switch (type) {
    case SCCBF_OBS_ELLIPSE: ellipse_(e.x, e.y, o, d); break;
    case SCCBF_OBS_CONE: cone_(e.x, e.y, e.th, e.v, e.cth, e.sth, o, p.zero_tol, d); break;
    case SCCBF_OBS_LANE: lane_(e.x, e.y, o, p.zero_tol, d); break;
    case SCCBF_OBS_RADIAL: radial_(e.x, e.y, e.v, o, d); break;
    default:
        a0 = T(0); a1 = T(0); b = T(-1); h = T(1);
        return;
    }
    h = d.h;
    if (p.dynamics == SCCBF_DYN_KBM) {
        a0 = e.cth * d.hx + e.sth * d.hy;
        a1 = d.hth;
        b = -(p.alpha * d.h);
    } else {
        a0 = d.hv;
        a1 = e.v * (-e.sth * d.hx + e.cth * d.hy) + (e.v * p.inv_lr) * d.hth;
        T lf_h = e.v * (e.cth * d.hx + e.sth * d.hy);
        b = -(lf_h + p.alpha * d.h + d.ht);
    }
}

// u-space row -> unit-normal row in w-space; zero rows keep at = 0.
template <typename T>
__device__ inline void to_wspace_(T a0, T a1, T b, const Ego<T>& e, const KArgs<T>& p,
                                  T& at0, T& at1, T& bt)
{
    T t0 = a0 * p.inv_l00;
    T t1 = (a1 - p.l10 * t0) * p.inv_l11;
    T rhs = b - (a0 * e.un0 + a1 * e.un1);
    T n2 = t0 * t0 + t1 * t1;
    if (n2 > Lim<T>::tiny_n2) {
        T inv = rsqrt_(n2);
        at0 = t0 * inv;
        at1 = t1 * inv;
        bt = rhs * inv;
    } else {
        at0 = T(0);
        at1 = T(0);
        bt = rhs;
    }
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
        e.un0 = ur0;
        e.un1 = ur1;
        return;
    }
    e.x = s.x; e.y = s.y; e.th = s.z; e.v = s.w;
    bad |= !(finite_(s.x) && finite_(s.y) && finite_(s.z) && finite_(s.w));
    sincos_(e.th, &e.sth, &e.cth);
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

template <typename T>
__device__ inline void store_result_(const KArgs<T>& p, long long qp, bool qp_only, int st,
                                     T w0, T w1, const Ego<T>& e, T ur0, T ur1)
{
    using v2 = typename VT<T>::v2;
    reinterpret_cast<v2*>(p.u_out)[qp] = result_u_(p, qp_only, st, w0, w1, e, ur0, ur1);
    p.status[qp] = st;
}

// ------------------------------------------------- G lanes per QP, rows in registers
// What one lane holds of one QP between its loads and its arithmetic.
template <typename T, int RPL> struct LaneLoads {
    typename VT<T>::v2 urv;
    typename VT<T>::v4 sv;
    typename VT<T>::v4 oa[RPL], ob[RPL];
    int ty[RPL];
};

// Every load of lane g of QP `qp` is issued up front, in the order the arithmetic
// consumes them (ego first, then slot by slot), as straight-line code: slots past M
// re-read slot M-1 and are turned into empty rows afterwards.
// NT: obstacle slots are loaded with the non-temporal hint (read once, not worth keeping in
// L2 / Infinity Cache); pays off for batches much larger than those caches.
// The lane's slots are m0, m0 + dm, ..., m0 + (RPL - 1) dm.
template <typename T, int RPL, bool QP_ONLY, bool NT>
__device__ inline void group_load_(const KArgs<T>& p, long long qp, int m0, int dm, LaneLoads<T, RPL>& L)
{
    using v4 = typename VT<T>::v4;
    using v2 = typename VT<T>::v2;
    const int M = p.M;
    const bool type_pa = (p.flags & SCCBF_FLAG_TYPE_PER_AGENT) != 0;
    const v4* obs4 = reinterpret_cast<const v4*>(p.obs);
    L.urv = reinterpret_cast<const v2*>(p.u_ref)[qp];
    L.sv = v4{T(0), T(0), T(0), T(0)};
    if (!QP_ONLY) L.sv = reinterpret_cast<const v4*>(p.state)[qp];
    const ObsAddr oad = obs_addr_(p, qp);
    const unsigned tstr = type_pa ? (unsigned)p.A : 1u;
    const unsigned long long toff = type_pa ? (unsigned long long)qp : 0ull;
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
        const int m = m0 + r * dm;
        const int mm = m < M ? m : M - 1;
        L.ty[r] = 0;
        // keep the issue order (the vmcnt counter retires loads in order, so the order
        // decides how early each row's arithmetic may start)
        __builtin_amdgcn_sched_barrier(0);
        if (QP_ONLY) {  // the host never launches this kernel with M = 0
            L.oa[r] = obs4[(long long)mm * p.A + qp];
            L.ob[r] = v4{T(0), T(0), T(0), T(0)};
        } else {
            const unsigned long long i0 = oad.at(mm);
            L.ty[r] = p.obs_type[(unsigned long long)(unsigned)mm * tstr + toff];
            if (NT) {
                L.oa[r] = nt_load4_<T>(&obs4[i0]);
                L.ob[r] = nt_load4_<T>(&obs4[i0 + oad.half]);
            } else {
                L.oa[r] = obs4[i0];
                L.ob[r] = obs4[i0 + oad.half];
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

// Rows of this lane (slots m0, m0 + dm, ...) from its loaded slots: unit normals in w-space.
// Returns non-zero if the state, the nominal input or a row is not finite.
template <typename T, int RPL, bool QP_ONLY>
__device__ inline int lane_rows_(const KArgs<T>& p, long long qp, int m0, int dm, bool live,
                                 const LaneLoads<T, RPL>& L, Ego<T>& e, T& ur0, T& ur1,
                                 T (&at0)[RPL], T (&at1)[RPL], T (&bt)[RPL], T* h_min = nullptr)
{
    using v4 = typename VT<T>::v4;
    const int M = p.M;
    int bad;
    make_ego_(p, L.sv, L.urv, QP_ONLY, e, ur0, ur1, bad);
    T chk = T(0);  // stays 0 while every row is finite (x * 0 is NaN for Inf / NaN)
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
        const int m = m0 + r * dm;
        T a0, a1, b, h;
        if (QP_ONLY) {
            a0 = L.oa[r].x; a1 = L.oa[r].y; b = L.oa[r].z; h = T(0);
            if (m >= M) { a0 = T(0); a1 = T(0); b = T(-1); }
        } else {
            const T o[8] = {L.oa[r].x, L.oa[r].y, L.oa[r].z, L.oa[r].w,
                            L.ob[r].x, L.ob[r].y, L.ob[r].z, L.ob[r].w};
            const int ty = m < M ? L.ty[r] : SCCBF_OBS_EMPTY;
            slot_row_(ty, o, e, p, a0, a1, b, h);
            if (p.rows_out && m < M && live)
                reinterpret_cast<v4*>(p.rows_out)[(long long)m * p.A + qp] = v4{a0, a1, b, h};
            if (h_min && ty >= 0 && ty <= SCCBF_OBS_RADIAL) *h_min = min_(*h_min, h);  // real slots only
        }
        chk += a0 * T(0) + a1 * T(0) + b * T(0);
        to_wspace_(a0, a1, b, e, p, at0[r], at1[r], bt[r]);
    }
    return bad | !(chk == T(0));
}

// Active-set iteration of a group of G lanes: lane g holds rows r G + g in registers; with
// LDS_PICK all rows of the group's QP are also in LDS at s_rows[c * plane + qslot + m].
template <typename T, int G, int RPL, bool LDS_PICK>
__device__ inline Pivot<T> active_set_(int M, int g, const T (&at0)[RPL], const T (&at1)[RPL],
                                       const T (&bt)[RPL], const T* s_rows, int plane, int qslot, int bad)
{
    // take the most violated row, move the optimum onto it
    Pivot<T> s;
    s.reset(bad ? SCCBF_STATUS_NONFINITE : SCCBF_STATUS_OPTIMAL);
    unsigned idx[RPL];  // 255 - row index: ties in violation go to the lowest index
#pragma unroll
    for (int r = 0; r < RPL; ++r) idx[r] = (unsigned)(255 - (r * G + g));
    const int cap = kCapPerRow * M + kCapBase;
    // One step: find the most violated row of the group, move the optimum onto it.
    // Returns false (wave-uniform) once no group of the wave has work left.
    auto step = [&](auto ns_max, bool first, int it) -> bool {
        // rows are unit-normal, so bt - a.w is a distance; below thr it is rounding
        const T thr = Lim<T>::viol_tol * (abs_(s.w0) + abs_(s.w1)) + Lim<T>::tiny_n2;
        // key = violation in the high bits, idx below: the maximum is the most violated
        // row and the keys of a group are distinct.  Keys are compared as signed integers,
        // so a row that is satisfied (negative violation) loses against any violated one.
        int key[RPL];
        int kbest = 0;
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            const T viol = first ? bt[r] : bt[r] - at0[r] * s.w0 - at1[r] * s.w1;
            key[r] = (int)((viol_bits(viol, thr) & ~0xFFu) | idx[r]);
            kbest = (r == 0 || key[r] > kbest) ? key[r] : kbest;  // always the key of a real slot
        }
        const int kmax = group_imax<G>(kbest);
        const bool go = (s.st == SCCBF_STATUS_OPTIMAL) & viol_key_above(kmax, thr);
        if (!__any(go)) return false;  // every group of the wave is done
        if (it >= cap) {               // no convergence: rows numerically inconsistent
            s.st = go ? SCCBF_STATUS_INFEASIBLE : s.st;
            return false;
        }
        T c0 = T(0), c1 = T(0), cb = T(0);
        if (LDS_PICK) {
            const int at = qslot + 255 - (kmax & 0xFF);
            c0 = s_rows[at];
            c1 = s_rows[plane + at];
            cb = s_rows[2 * plane + at];
        } else {
            // the lane that holds the winning row publishes it (the other lanes contribute 0)
#pragma unroll
            for (int r = 0; r < RPL; ++r) {
                const bool sel = key[r] == kmax;
                c0 = sel ? at0[r] : c0;
                c1 = sel ? at1[r] : c1;
                cb = sel ? bt[r] : cb;
            }
            if (G > 1) {
                c0 = group_bcast<G>(c0);
                c1 = group_bcast<G>(c1);
                cb = group_bcast<G>(cb);
            }
        }
        pivot_onto_<decltype(ns_max)::value>(s, c0, c1, cb, go);
        return true;
    };
    // wave-uniform trip count, lanes commit under `go`; the first step starts from w = 0
    // with no active row and the second one has at most one
    if (step(std::integral_constant<int, 0>(), true, 0) && step(std::integral_constant<int, 1>(), false, 1)) {
        for (int it = 2; step(std::integral_constant<int, 2>(), false, it); ++it) {}
    }
    return s;
}

// Rows of this lane from its loaded slots, then the active-set iteration of the group.
template <typename T, int G, int RPL, bool QP_ONLY>
__device__ inline void group_solve_(const KArgs<T>& p, long long qp, int g, const LaneLoads<T, RPL>& L)
{
    Ego<T> e;
    T ur0, ur1;
    T at0[RPL], at1[RPL], bt[RPL];
    int bad = lane_rows_<T, RPL, QP_ONLY>(p, qp, g, G, true, L, e, ur0, ur1, at0, at1, bt);
    bad = (int)group_or<G>((unsigned)bad);

    // The rows also go to LDS, row m of the group's QP at [component][qp slot][m], so that
    // the winning row of a step is one indexed read per component instead of a chain of
    // selects over the registers of every lane.  Each wavefront reads only what it wrote.
    constexpr int kMpad = G * RPL + 1;          // odd stride between QPs: fewer bank conflicts
    constexpr int kPlane = (kBlock / G) * kMpad;
    constexpr bool kLdsPick = (size_t)3 * kPlane * sizeof(T) <= 53 * 1024;  // 3 workgroups per CU at most
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
    if (g == 0) store_result_(p, qp, QP_ONLY, s.st, s.w0, s.w1, e, ur0, ur1);
}

// The arguments every wave needs before it can issue its first loads come first and one
// by one: the leading 16 dwords of the kernel arguments are preloaded into SGPRs at wave
// launch (-amdgpu-kernarg-preload-count), which a by-value struct does not qualify for.
template <typename T, int G, int RPL, bool QP_ONLY, bool NT>
__global__ __launch_bounds__(kBlock) void k_group(long long A, const T* obs, const T* state, const T* u_ref,
                                                  const uint8_t* obs_type, int M, unsigned flags,
                                                  const KArgs<T> rest)
{
    KArgs<T> p = rest;
    p.A = A; p.obs = obs; p.state = state; p.u_ref = u_ref; p.obs_type = obs_type; p.M = M; p.flags = flags;
    const int g = (int)(threadIdx.x % G);
    const long long qp = (long long)blockIdx.x * (kBlock / G) + threadIdx.x / G;
    if (qp >= p.A) return;  // groups are lane-aligned: a whole group leaves together
    LaneLoads<T, RPL> L;
    group_load_<T, RPL, QP_ONLY, NT>(p, qp, g, G, L);
    group_solve_<T, G, RPL, QP_ONLY>(p, qp, g, L);
}

// ------------------------------------------ one wavefront per QP, rows staged in LDS
template <typename T, int WPB, bool QP_ONLY>
__global__ __launch_bounds__(WPB * 64) void k_wave(const KArgs<T> p)
{
    using v4 = typename VT<T>::v4;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int M = p.M;
    T* s_at0 = reinterpret_cast<T*>(smem_raw) + (size_t)wv * 3 * M;
    T* s_at1 = s_at0 + M;
    T* s_bt = s_at1 + M;

    long long qp = (long long)blockIdx.x * WPB + wv;
    const bool live = qp < p.A;
    if (!live) qp = p.A - 1;  // keep the wave for the block barrier; its stores are masked

    const bool type_pa = (p.flags & SCCBF_FLAG_TYPE_PER_AGENT) != 0;
    const v4* obs4 = reinterpret_cast<const v4*>(p.obs);
    const ObsAddr oad = obs_addr_(p, qp);

    Ego<T> e;
    T ur0, ur1;
    int bad;
    load_ego_(p, qp, QP_ONLY, e, ur0, ur1, bad);

    for (int m = lane; m < M; m += 64) {
        T a0, a1, b, h;
        if (QP_ONLY) {
            const v4 rw = obs4[(long long)m * p.A + qp];
            a0 = rw.x; a1 = rw.y; b = rw.z; h = T(0);
        } else {
            const long long i0 = oad.at(m);
            const v4 oa = obs4[i0];
            const v4 ob = obs4[i0 + oad.half];
            const int ty = type_pa ? p.obs_type[(long long)m * p.A + qp] : p.obs_type[m];
            const T o[8] = {oa.x, oa.y, oa.z, oa.w, ob.x, ob.y, ob.z, ob.w};
            slot_row_(ty, o, e, p, a0, a1, b, h);
            if (p.rows_out && live)
                reinterpret_cast<v4*>(p.rows_out)[(long long)m * p.A + qp] = v4{a0, a1, b, h};
        }
        bad |= !(finite_(a0) && finite_(a1) && finite_(b));
        T t0, t1, tb;
        to_wspace_(a0, a1, b, e, p, t0, t1, tb);
        s_at0[m] = t0;
        s_at1[m] = t1;
        s_bt[m] = tb;
    }
    bad = __any(bad);
    __syncthreads();

    Pivot<T> s;
    s.reset(bad ? SCCBF_STATUS_NONFINITE : SCCBF_STATUS_OPTIMAL);
    const int cap = kCapPerRow * M + kCapBase;
    for (int it = 0;; ++it) {  // wave-uniform loop
        const T thr = Lim<T>::viol_tol * (abs_(s.w0) + abs_(s.w1)) + Lim<T>::tiny_n2;
        unsigned kbest = 0u;
        for (int m = lane; m < M; m += 64) {
            const T viol = s_bt[m] - (s_at0[m] * s.w0 + s_at1[m] * s.w1);
            const unsigned key = viol > thr ? ((viol_bits_pos(viol) & ~0x3FFu) | (unsigned)(1023 - m)) : 0u;
            kbest = key > kbest ? key : kbest;
        }
        const unsigned kmax = group_umax<64>(kbest);
        const bool go = (s.st == SCCBF_STATUS_OPTIMAL) & (kmax != 0u);
        if (!go) break;
        if (it >= cap) {
            s.st = SCCBF_STATUS_INFEASIBLE;
            break;
        }
        const int sel = 1023 - (int)(kmax & 0x3FFu);
        pivot_onto_(s, s_at0[sel], s_at1[sel], s_bt[sel], true);  // broadcast reads
    }
    const int st = s.st;
    const T w0 = s.w0, w1 = s.w1;
    if (lane == 0 && live) store_result_(p, qp, QP_ONLY, st, w0, w1, e, ur0, ur1);
}

// ------------------------------------- barrier value and partials only (no rows, no QP)
// One thread per (slot, agent); out is T[M][2][A][4] = (h, h_x, h_y, h_theta | h_v, h_t, 0, 0).
template <typename T>
__global__ __launch_bounds__(kBlock) void k_partials(const KArgs<T> p)
{
    using v4 = typename VT<T>::v4;
    const long long gid = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= p.A * p.M) return;
    const int m = (int)(gid / p.A);
    const long long a = gid % p.A;
    const bool type_pa = (p.flags & SCCBF_FLAG_TYPE_PER_AGENT) != 0;
    const v4* obs4 = reinterpret_cast<const v4*>(p.obs);
    const ObsAddr oad = obs_addr_(p, a);
    const long long i0 = oad.at(m);
    const v4 oa = obs4[i0];
    const v4 ob = obs4[i0 + oad.half];
    const int ty = type_pa ? p.obs_type[(long long)m * p.A + a] : p.obs_type[m];
    const v4 s = reinterpret_cast<const v4*>(p.state)[a];
    T sth, cth;
    sincos_(s.z, &sth, &cth);
    const T o[8] = {oa.x, oa.y, oa.z, oa.w, ob.x, ob.y, ob.z, ob.w};
    Partials<T> d;
    switch (ty) {
    case SCCBF_OBS_ELLIPSE: ellipse_(s.x, s.y, o, d); break;
    case SCCBF_OBS_CONE: cone_(s.x, s.y, s.z, s.w, cth, sth, o, p.zero_tol, d); break;
    case SCCBF_OBS_LANE: lane_(s.x, s.y, o, p.zero_tol, d); break;
    case SCCBF_OBS_RADIAL: radial_(s.x, s.y, s.w, o, d); break;
    default: d.h = T(1); d.hx = d.hy = d.hth = d.hv = d.ht = T(0); break;
    }
    v4* out = reinterpret_cast<v4*>(p.rows_out);
    out[(long long)m * 2 * p.A + a] = v4{d.h, d.hx, d.hy, d.hth};
    out[(long long)m * 2 * p.A + p.A + a] = v4{d.hv, d.ht, T(0), T(0)};
}

// ----------------------------------------- no rows at all: the nominal input is optimal
template <typename T, bool QP_ONLY>
__global__ __launch_bounds__(kBlock) void k_passthrough(const KArgs<T> p)
{
    using v4 = typename VT<T>::v4;
    using v2 = typename VT<T>::v2;
    const long long a = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (a >= p.A) return;
    const v2 ur = reinterpret_cast<const v2*>(p.u_ref)[a];
    bool ok = finite_(ur.x) && finite_(ur.y);
    if (!QP_ONLY) {
        const v4 s = reinterpret_cast<const v4*>(p.state)[a];
        ok = ok && finite_(s.x) && finite_(s.y) && finite_(s.z) && finite_(s.w);
    }
    reinterpret_cast<v2*>(p.u_out)[a] = ur;
    p.status[a] = ok ? SCCBF_STATUS_OPTIMAL : SCCBF_STATUS_NONFINITE;
}

// ------------------------------------------- nominal input: Stanley steering + PID speed
template <typename T> struct NArgs {
    T axle, k, ks, kp, ki, kd, dt, inv_dt, target_speed;
    unsigned flags;
    long long A;
    int N;
    const T* state;
    const T* traj;
    int* target_idx;
    T* pid_state;
    T* u_ref;
};

constexpr int kTrajChunk = 2048;  // way-points staged in LDS per pass
constexpr int kSub = 32;          // way-points per bounding box
constexpr int kSubs = kTrajChunk / kSub;
constexpr int kSubStride = kSub + 1;  // odd stride: groups scanning different runs hit different banks
constexpr int kNomLanes = 4;          // lanes per agent in the nominal and rollout kernels
constexpr int kNomBlock = 1024;       // threads per workgroup of the nominal kernel: 256 agents share one staging

template <typename T> __device__ inline T wrap_pi_(T a)
{
    // reference cbf/utils.py:93-106 subtracts / adds 2 pi until the angle is in [-pi, pi]
    constexpr T kPi = T(3.14159265358979323846), k2Pi = T(6.28318530717958647692);
    a -= k2Pi * rint(a * (T(1) / k2Pi));
    a = a > kPi ? a - k2Pi : a;
    a = a < -kPi ? a + k2Pi : a;
    return a;
}

// One pass (up to 2048 way-points) of the course in LDS: coordinates in runs of 32 and the
// bounding box of every run (four extra empty boxes for the four-at-a-time walk).
template <typename T> struct CoursePass {
    T x[kSubs * kSubStride], y[kSubs * kSubStride];
    T x0[kSubs + kNomLanes], x1[kSubs + kNomLanes], y0[kSubs + kNomLanes], y1[kSubs + kNomLanes];
};

// Every thread of a BLOCK-thread workgroup takes 2048 / BLOCK consecutive way-points of the
// pass [base, base + n); 32 * BLOCK / 2048 neighbouring threads make one run and reduce its
// box by shuffles.  Way-points past the end are infinitely far, runs past the end empty.
// The caller puts a workgroup barrier before and after.
template <typename T, int BLOCK>
__device__ inline void stage_course_(CoursePass<T>& c, const typename VT<T>::v4* traj4, int base, int n)
{
    using v2 = typename VT<T>::v2;
    constexpr int kPer = kTrajChunk / BLOCK;  // way-points per thread
    constexpr int kThr = kSub / kPer;         // threads per run
    static_assert(kPer * BLOCK == kTrajChunk && kThr * kPer == kSub && kThr <= 64, "block size");
    T x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY;
    const int r = threadIdx.x / kThr, j0 = (threadIdx.x % kThr) * kPer;
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const int i = threadIdx.x * kPer + j;
        T x = INFINITY, y = INFINITY;
        if (i < n) {
            const v2 w = *reinterpret_cast<const v2*>(traj4 + base + i);
            x = w.x; y = w.y;
            x0 = min_(x0, x); x1 = max_(x1, x);
            y0 = min_(y0, y); y1 = max_(y1, y);
        }
        c.x[r * kSubStride + j0 + j] = x;
        c.y[r * kSubStride + j0 + j] = y;
    }
#pragma unroll
    for (int m = 1; m < kThr; m *= 2) {
        x0 = min_(x0, __shfl_xor(x0, m, 64)); x1 = max_(x1, __shfl_xor(x1, m, 64));
        y0 = min_(y0, __shfl_xor(y0, m, 64)); y1 = max_(y1, __shfl_xor(y1, m, 64));
    }
    if (threadIdx.x % kThr == 0) {
        c.x0[r] = x0; c.x1[r] = x1; c.y0[r] = y0; c.y1[r] = y1;
    }
    if (threadIdx.x < kNomLanes) {
        c.x0[kSubs + threadIdx.x] = INFINITY; c.x1[kSubs + threadIdx.x] = -INFINITY;
        c.y0[kSubs + threadIdx.x] = INFINITY; c.y1[kSubs + threadIdx.x] = -INFINITY;
    }
}

// Nearest way-point of one staged pass to (fx, fy), by the four lanes of a group: they step
// through the boxes four at a time and scan (eight way-points each) only the runs whose box
// is not farther than lim, the best squared distance known so far (on entry: any upper bound
// of the answer).  best / bi: this lane's nearest way-point so far (strict "<" in index
// order, so the first index wins a tie).  Every group walks its own list of runs: a wavefront
// pays for the longest list among its groups, not for their union.
template <typename T>
__device__ inline void nearest_in_pass_(const CoursePass<T>& c, T fx, T fy, int g, int base, int runs,
                                        T& lim, T& best, int& bi)
{
    constexpr int G = kNomLanes;
    for (int r = 0; r < runs;) {  // r and lim are the same in the four lanes of a group
        // lane g looks at the box of run r + g (infinitely far when the run is empty)
        const T dx = max_(max_(c.x0[r + g] - fx, fx - c.x1[r + g]), T(0));
        const T dy = max_(max_(c.y0[r + g] - fy, fy - c.y1[r + g]), T(0));
        // (a few ulps of slack: the box distance and the way-point distance round independently)
        const unsigned hit = group_or<G>((dx * dx + dy * dy <= lim * (T(1) + T(8) * Lim<T>::eps)) ? 1u << g : 0u);
        if (hit == 0u) {
            r += G;
            continue;
        }
        r += __ffs(hit) - 1;  // the first run of the four that can hold the answer
#pragma unroll
        for (int j = 0; j < kSub / G; ++j) {  // lane g: way-points 8 g .. 8 g + 7 of the run
            const int jj = g * (kSub / G) + j;
            const T ex = fx - c.x[r * kSubStride + jj], ey = fy - c.y[r * kSubStride + jj];
            const T d = ex * ex + ey * ey;
            const bool lt = d < best;
            best = lt ? d : best;
            bi = lt ? base + r * kSub + jj : bi;
        }
        // tighten the bound with the group's best so far
        T m = best;
        m = min_(m, __shfl_xor(m, 1, 64));
        m = min_(m, __shfl_xor(m, 2, 64));
        lim = min_(lim, m);
        ++r;
    }
}

// the group's answer: least distance, then least index
template <typename T> __device__ inline int group_nearest_(T best, int bi, int N)
{
    T m = best;
    m = min_(m, __shfl_xor(m, 1, 64));
    m = min_(m, __shfl_xor(m, 2, 64));
    int c = best == m ? bi : 0x7fffffff;
    c = min(c, __shfl_xor(c, 1, 64));
    c = min(c, __shfl_xor(c, 2, 64));
    return min(c, N - 1);  // (non-finite state: no way-point compares nearer; any index will do)
}

// Stanley steering and PID speed control from the nearest way-point (xn, yn), the target
// way-point's (yaw, v_des) and the PID state ps = (previous error, integral), updated in place.
// Reference cbf/controllers.py:96-99, 118-119, 141-147, 173-180.
template <typename T>
__device__ inline void stanley_pid_(const NArgs<T>& p, T yaw, T v, T sth, T cth, T fx, T fy, T xn, T yn,
                                    T yaw_des, T v_des, bool pid, typename VT<T>::v2& ps, T& acc, T& delta)
{
    // cross-track error projected on the axle direction (-cos(yaw + pi/2), -sin(yaw + pi/2)) = (sin yaw, -cos yaw)
    const T err = (fx - xn) * sth - (fy - yn) * cth;
    const T theta_e = wrap_pi_(yaw_des - yaw);
    const T theta_d = atan2_(p.k * err, v + p.ks);
    delta = theta_e + theta_d;
    const T vt = (p.flags & SCCBF_NOMINAL_WAYPOINT_SPEED) ? v_des : p.target_speed;
    const T e = vt - v;
    acc = p.kp * e;
    if (pid) {
        const T de = (e - ps.x) * p.inv_dt;
        ps.y += p.dt * e;
        acc = p.kp * e + p.ki * ps.y + p.kd * de;
        ps.x = e;
    }
}

// Four lanes per agent.  The nearest way-point is searched over the whole course
// (controllers.py:89-93: np.argmin of np.hypot, first index on ties; squared distances order
// the same way), exactly but not exhaustively: the workgroup stages the course through LDS
// 2048 way-points at a time together with the bounding box of every run of 32, and the
// distance to last tick's target way-point bounds the answer from above (nearest_in_pass_).
template <typename T>
__global__ __launch_bounds__(kNomBlock) void k_nominal(const NArgs<T> p)
{
    using v4 = typename VT<T>::v4;
    using v2 = typename VT<T>::v2;
    constexpr int G = kNomLanes;
    __shared__ CoursePass<T> c;
    const int g = threadIdx.x % G;
    long long a = (long long)blockIdx.x * (kNomBlock / G) + threadIdx.x / G;
    const bool live = a < p.A;
    if (!live) a = p.A - 1;  // stays for the barriers; stores are masked
    const v4* traj4 = reinterpret_cast<const v4*>(p.traj);
    const v4 s = reinterpret_cast<const v4*>(p.state)[a];
    T sth, cth;
    sincos_(s.z, &sth, &cth);
    const T fx = s.x + p.axle * cth, fy = s.y + p.axle * sth;
    const int last = p.target_idx[a];
    // upper bound of the answer: the distance to the way-point targeted last tick
    T lim;
    {
        const v2 wl = *reinterpret_cast<const v2*>(traj4 + min(max(last, 0), p.N - 1));
        lim = (fx - wl.x) * (fx - wl.x) + (fy - wl.y) * (fy - wl.y);
    }
    T best = INFINITY;
    int bi = 0x7fffffff;
    for (int base = 0; base < p.N; base += kTrajChunk) {
        const int n = min(kTrajChunk, p.N - base);
        __syncthreads();
        stage_course_<T, kNomBlock>(c, traj4, base, n);
        __syncthreads();
        nearest_in_pass_(c, fx, fy, g, base, (n + kSub - 1) / kSub, lim, best, bi);
    }
    bi = group_nearest_(best, bi, p.N);
    const v4 wn = traj4[bi];
    const int idx = last >= bi ? last : bi;  // controllers.py:118-119
    const v4 wt = traj4[min(idx, p.N - 1)];
    v2 ps = v2{T(0), T(0)};
    if (p.pid_state) ps = reinterpret_cast<v2*>(p.pid_state)[a];
    T acc, delta;
    stanley_pid_(p, s.z, s.w, sth, cth, fx, fy, wn.x, wn.y, wt.z, wt.w, p.pid_state != nullptr, ps, acc, delta);
    if (live && g == 0) {
        if (p.pid_state) reinterpret_cast<v2*>(p.pid_state)[a] = ps;
        p.target_idx[a] = idx;
        reinterpret_cast<v2*>(p.u_ref)[a] = v2{acc, delta};
    }
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

template <typename T>
__global__ __launch_bounds__(kBlock) void k_vehicle_step(long long A, T* state, const T* u, T k_in,
                                                         T inv_lr, T max_steer, T dt, T* beta_out)
{
    using v4 = typename VT<T>::v4;
    using v2 = typename VT<T>::v2;
    const long long a = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (a >= A) return;
    v4 s = reinterpret_cast<v4*>(state)[a];
    const v2 in = reinterpret_cast<const v2*>(u)[a];
    const T beta = vehicle_step_<T>(s, in.x, in.y, k_in, inv_lr, max_steer, dt);
    reinterpret_cast<v4*>(state)[a] = s;
    if (beta_out) beta_out[a] = beta;
}

// ---------------------------------------------------- closed-loop rollout in one launch
template <typename T> struct RArgs {
    KArgs<T> f;   // filter: parameters, obs, obs_type, M, flags (state / u_ref / u_out / status unused)
    NArgs<T> n;   // nominal: parameters, traj, N, target_idx (in/out), pid_state (in/out)
    T* state;     // [A][4] in/out
    T k_in, inv_lr, max_steer, dt;
    int ticks;
    int motion;   // SCCBF_MOTION_*
    T* h_min;     // [A] optional
    int* n_active;      // [A] optional
    int* n_infeasible;  // [A] optional
    T* trace;     // [ticks][A][4] optional
};

// obstacle slots move between ticks (constant velocity): ellipse by (vx, vy), collision cone by
// v_o along theta_o + beta, radial by (cdot_x, cdot_y); lanes and empty slots stay
template <typename T>
__device__ inline void advance_slot_(int type, typename VT<T>::v4& oa, typename VT<T>::v4& ob, T dt)
{
    if (type == SCCBF_OBS_ELLIPSE) {
        oa.x += ob.y * dt;
        oa.y += ob.z * dt;
    } else if (type == SCCBF_OBS_CONE) {
        T so, co;
        sincos_(oa.z + ob.y, &so, &co);
        oa.x += oa.w * co * dt;
        oa.y += oa.w * so * dt;
    } else if (type == SCCBF_OBS_RADIAL) {
        oa.x += ob.x * dt;
        oa.y += ob.y * dt;
    }
}

// `ticks` control ticks of nominal controller -> filter -> vehicle step (-> obstacle motion)
// for every agent in ONE launch: four lanes per agent, the agent's state, its obstacle slots
// (four per lane), target index and PID state stay in registers, the course (at most 2048
// way-points) and its run boxes are staged in LDS once per workgroup, the rows of each tick go
// through the per-wavefront LDS table of the 4-lane filter kernel.  No HBM traffic inside the
// loop except the optional trace.
template <typename T, int RPL>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(SCCBF_ROLLOUT_WAVES, SCCBF_ROLLOUT_WAVES)))
void k_rollout(const RArgs<T> a)
{
    using v4 = typename VT<T>::v4;
    using v2 = typename VT<T>::v2;
    constexpr int G = kNomLanes;
    constexpr int kMpad = G * RPL + 1;
    constexpr int kPlane = (kBlock / G) * kMpad;
    __shared__ CoursePass<T> c;
    __shared__ T s_yaw[kTrajChunk];
    __shared__ T s_rows[3 * kPlane];
    const KArgs<T>& p = a.f;
    const NArgs<T>& np = a.n;
    const int g = threadIdx.x % G;
    long long ag = (long long)blockIdx.x * (kBlock / G) + threadIdx.x / G;
    const bool live = ag < p.A;
    if (!live) ag = p.A - 1;  // stays for the barrier; stores are masked
    const int qslot = (int)(threadIdx.x / G) * kMpad;
    const v4* traj4 = reinterpret_cast<const v4*>(np.traj);

    stage_course_<T, kBlock>(c, traj4, 0, np.N);
    for (int i = threadIdx.x; i < np.N; i += kBlock) s_yaw[i] = traj4[i].z;

    // this lane's obstacle slots (r G + g) and everything the agent carries from tick to tick
    LaneLoads<T, RPL> L;
    {
        KArgs<T> q = p;
        q.state = a.state;
        q.u_ref = np.u_ref;  // any readable (A, 2) array: overwritten below before use
        group_load_<T, RPL, false, false>(q, ag, g, G, L);
    }
    v4 st = L.sv;
    int tgt = np.target_idx[ag];
    v2 ps = v2{T(0), T(0)};
    if (np.pid_state) ps = reinterpret_cast<v2*>(np.pid_state)[ag];
    T h_min = INFINITY;
    int n_act = 0, n_inf = 0;
    __syncthreads();
    const int runs = (np.N + kSub - 1) / kSub;

    for (int tick = 0; tick < a.ticks; ++tick) {
        // nominal input
        T sth, cth;
        sincos_(st.z, &sth, &cth);
        const T fx = st.x + np.axle * cth, fy = st.y + np.axle * sth;
        const int lt = min(max(tgt, 0), np.N - 1);
        const T lx = c.x[(lt / kSub) * kSubStride + lt % kSub], ly = c.y[(lt / kSub) * kSubStride + lt % kSub];
        T lim = (fx - lx) * (fx - lx) + (fy - ly) * (fy - ly);
        T best = INFINITY;
        int bi = 0x7fffffff;
        nearest_in_pass_(c, fx, fy, g, 0, runs, lim, best, bi);
        bi = group_nearest_(best, bi, np.N);
        tgt = tgt >= bi ? tgt : bi;
        const int ti = min(tgt, np.N - 1);
        T v_des = np.target_speed;
        if (np.flags & SCCBF_NOMINAL_WAYPOINT_SPEED) v_des = traj4[ti].w;
        T acc, delta;
        stanley_pid_(np, st.z, st.w, sth, cth, fx, fy, c.x[(bi / kSub) * kSubStride + bi % kSub],
                     c.y[(bi / kSub) * kSubStride + bi % kSub], s_yaw[ti], v_des, np.pid_state != nullptr, ps,
                     acc, delta);
        // filter
        L.sv = st;
        L.urv = v2{acc, delta};
        Ego<T> e;
        T ur0, ur1;
        T at0[RPL], at1[RPL], bt[RPL];
        T hm = INFINITY;
        int bad = lane_rows_<T, RPL, false>(p, ag, g, G, false, L, e, ur0, ur1, at0, at1, bt, &hm);
        bad = (int)group_or<G>((unsigned)bad);
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            s_rows[qslot + r * G + g] = at0[r];
            s_rows[kPlane + qslot + r * G + g] = at1[r];
            s_rows[2 * kPlane + qslot + r * G + g] = bt[r];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const Pivot<T> s = active_set_<T, G, RPL, true>(p.M, g, at0, at1, bt, s_rows, kPlane, qslot, bad);
        __builtin_amdgcn_wave_barrier();  // the table is rewritten next tick
        const v2 u = result_u_(p, false, s.st, s.w0, s.w1, e, ur0, ur1);
        hm = min_(hm, __shfl_xor(hm, 1, 64));
        hm = min_(hm, __shfl_xor(hm, 2, 64));
        h_min = min_(h_min, hm);
        n_act += (s.st == SCCBF_STATUS_OPTIMAL) & ((s.w0 != T(0)) | (s.w1 != T(0)));
        n_inf += s.st == SCCBF_STATUS_INFEASIBLE;
        // vehicle step, then the obstacles move
        const T beta = vehicle_step_<T>(st, u.x, u.y, a.k_in, a.inv_lr, a.max_steer, a.dt);
        if (a.trace && live && g == 0)
            reinterpret_cast<v4*>(a.trace)[(long long)tick * p.A + ag] = v4{u.x, u.y, beta, (T)s.st};
        if (a.motion == SCCBF_MOTION_CONSTANT_VELOCITY) {
#pragma unroll
            for (int r = 0; r < RPL; ++r) advance_slot_<T>(L.ty[r], L.oa[r], L.ob[r], a.dt);
        }
    }
    if (live && g == 0) {
        reinterpret_cast<v4*>(a.state)[ag] = st;
        np.target_idx[ag] = tgt;
        if (np.pid_state) reinterpret_cast<v2*>(np.pid_state)[ag] = ps;
        if (a.h_min) a.h_min[ag] = h_min;
        if (a.n_active) a.n_active[ag] = n_act;
        if (a.n_infeasible) a.n_infeasible[ag] = n_inf;
    }
}

// ------------------------------------------------------------------- host side
template <typename T>
int fill_args(const sccbf_params* sp, int64_t A, int32_t M, KArgs<T>& k)
{
    if (!sp || A < 0 || M < 0) return SCCBF_ERR_ARG;
    if (A > 0x7fffffffLL) return SCCBF_ERR_ARG;  // the kernels keep strides of up to 2 A in 32 bits
    if (M > SCCBF_MAX_ROWS) return SCCBF_ERR_ROWS;
    if (sp->dynamics != SCCBF_DYN_DBM && sp->dynamics != SCCBF_DYN_KBM) return SCCBF_ERR_ARG;
    if ((sp->flags & SCCBF_FLAG_OBS_SHARED) && (sp->flags & SCCBF_FLAG_OBS_TILED)) return SCCBF_ERR_ARG;
    const double r00 = sp->R[0], r01 = 0.5 * (sp->R[1] + sp->R[2]), r11 = sp->R[3];
    if (!(r00 > 0.0) || !std::isfinite(r00) || !std::isfinite(r01) || !std::isfinite(r11))
        return SCCBF_ERR_WEIGHT;
    const double l00 = std::sqrt(r00), l10 = r01 / l00, t = r11 - l10 * l10;
    if (!(t > 0.0)) return SCCBF_ERR_WEIGHT;
    const double l11 = std::sqrt(t);
    if (sp->dynamics == SCCBF_DYN_DBM && !(sp->lr > 0.0 && sp->lf + sp->lr > 0.0)) return SCCBF_ERR_ARG;
    if (sp->dynamics == SCCBF_DYN_KBM && !(sp->L > 0.0)) return SCCBF_ERR_ARG;
    k.alpha = (T)sp->alpha; k.lr = (T)sp->lr; k.lf = (T)sp->lf; k.L = (T)sp->L;
    k.inv_lr = (T)(1.0 / sp->lr); k.inv_L = (T)(1.0 / sp->L);
    k.k_in = (T)(sp->lr / (sp->lf + sp->lr)); k.k_out = (T)((sp->lf + sp->lr) / sp->lr);
    k.l10 = (T)l10; k.inv_l00 = (T)(1.0 / l00); k.inv_l11 = (T)(1.0 / l11);
    k.zero_tol = (T)sp->zero_tol;
    k.dynamics = sp->dynamics;
    k.flags = sp->flags;
    k.A = A;
    k.M = M;
    return SCCBF_OK;
}

// lanes-per-QP when the caller leaves the choice to the library (measured on MI355X with
// 16 rows: 4 lanes from 32768 QPs up to 1M, 8 around 8192-16384, 16 below; the
// one-wavefront-per-QP kernel only when the rows do not fit 16 lanes x 16 registers)
int auto_lanes(int64_t A, int M)
{
    int g;
    if (A >= (1 << 15)) g = 4;
    else if (A >= (1 << 13)) g = 8;
    else g = 16;
    while (g < 16 && M > 16 * g) g *= 2;  // keep at most 16 rows per lane
    if (M > 16 * g) return 64;
    return g;
}

// Obstacle bytes per launch from which the slots are loaded non-temporally: measured
// break-even between 34 MB (cached loads 8% faster) and 67 MB (non-temporal 1-4% faster,
// 5-10% from 134 MB on).
constexpr double kNtBytes = 48.0 * 1024 * 1024;

template <typename T, bool QP_ONLY, int G, int RPL>
int launch_group(const KArgs<T>& k, hipStream_t s)
{
    const long long per_block = kBlock / G;
    const long long blocks = (k.A + per_block - 1) / per_block;
    if (blocks > 0x7fffffffLL) return SCCBF_ERR_ARG;
    // rows of the QP-only entry are a third of the bytes of obstacle slots: always cached loads
    const bool nt = !QP_ONLY && (double)k.A * k.M * 8 * sizeof(T) >= kNtBytes;
    if (nt)
        hipLaunchKernelGGL((k_group<T, G, RPL, QP_ONLY, !QP_ONLY>), dim3((unsigned)blocks), dim3(kBlock), 0, s, k.A,
                           k.obs, k.state, k.u_ref, k.obs_type, k.M, k.flags, k);
    else
        hipLaunchKernelGGL((k_group<T, G, RPL, QP_ONLY, false>), dim3((unsigned)blocks), dim3(kBlock), 0, s, k.A,
                           k.obs, k.state, k.u_ref, k.obs_type, k.M, k.flags, k);
    return hipGetLastError() == hipSuccess ? SCCBF_OK : SCCBF_ERR_LAUNCH;
}

template <typename T, bool QP_ONLY, int G>
int launch_group_rpl(const KArgs<T>& k, hipStream_t s)
{
    const int rpl = (k.M + G - 1) / G;
    if (rpl <= 1) return launch_group<T, QP_ONLY, G, 1>(k, s);
    if (rpl <= 2) return launch_group<T, QP_ONLY, G, 2>(k, s);
    if (rpl <= 4) return launch_group<T, QP_ONLY, G, 4>(k, s);
    if (rpl <= 8) return launch_group<T, QP_ONLY, G, 8>(k, s);
    if (rpl <= 16) return launch_group<T, QP_ONLY, G, 16>(k, s);
    return SCCBF_ERR_VARIANT;
}

template <typename T, bool QP_ONLY>
int launch_wave(const KArgs<T>& k, hipStream_t s)
{
    const size_t per_wave = (size_t)3 * k.M * sizeof(T);
    if (per_wave * 4 <= 48 * 1024) {
        const long long blocks = (k.A + 3) / 4;
        if (blocks > 0x7fffffffLL) return SCCBF_ERR_ARG;
        hipLaunchKernelGGL((k_wave<T, 4, QP_ONLY>), dim3((unsigned)blocks), dim3(256), per_wave * 4, s, k);
    } else {
        if (k.A > 0x7fffffffLL) return SCCBF_ERR_ARG;
        hipLaunchKernelGGL((k_wave<T, 1, QP_ONLY>), dim3((unsigned)k.A), dim3(64), per_wave, s, k);
    }
    return hipGetLastError() == hipSuccess ? SCCBF_OK : SCCBF_ERR_LAUNCH;
}

template <typename T, bool QP_ONLY>
int dispatch(const sccbf_params* sp, KArgs<T>& k, void* stream)
{
    if (k.A == 0) return SCCBF_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (k.M == 0) {
        const long long blocks = (k.A + kBlock - 1) / kBlock;
        if (blocks > 0x7fffffffLL) return SCCBF_ERR_ARG;
        hipLaunchKernelGGL((k_passthrough<T, QP_ONLY>), dim3((unsigned)blocks), dim3(kBlock), 0, s, k);
        return hipGetLastError() == hipSuccess ? SCCBF_OK : SCCBF_ERR_LAUNCH;
    }
    int lanes = sp->lanes_per_qp;
    if (lanes == 0) lanes = auto_lanes(k.A, k.M);
    switch (lanes) {
    case 1: return launch_group_rpl<T, QP_ONLY, 1>(k, s);
    case 2: return launch_group_rpl<T, QP_ONLY, 2>(k, s);
    case 4: return launch_group_rpl<T, QP_ONLY, 4>(k, s);
    case 8: return launch_group_rpl<T, QP_ONLY, 8>(k, s);
    case 16: return launch_group_rpl<T, QP_ONLY, 16>(k, s);
    case 64: return launch_wave<T, QP_ONLY>(k, s);
    default: return SCCBF_ERR_VARIANT;
    }
}

template <typename T>
int solve_impl(const sccbf_params* sp, int64_t A, int32_t M, const T* state, const T* u_ref,
               const T* obs, const uint8_t* obs_type, T* u_out, int32_t* status, T* rows_out,
               void* stream)
{
    KArgs<T> k;
    int rc = fill_args<T>(sp, A, M, k);
    if (rc != SCCBF_OK) return rc;
    if (A > 0 && (!state || !u_ref || !u_out || !status)) return SCCBF_ERR_ARG;
    if (A > 0 && M > 0 && (!obs || !obs_type)) return SCCBF_ERR_ARG;
    k.state = state; k.u_ref = u_ref; k.obs = obs; k.obs_type = obs_type;
    k.u_out = u_out; k.status = status; k.rows_out = rows_out;
    return dispatch<T, false>(sp, k, stream);
}

template <typename T>
int partials_impl(const sccbf_params* sp, int64_t A, int32_t M, const T* state, const T* obs,
                  const uint8_t* obs_type, T* out, void* stream)
{
    KArgs<T> k;
    int rc = fill_args<T>(sp, A, M, k);
    if (rc != SCCBF_OK) return rc;
    if (A == 0 || M == 0) return SCCBF_OK;
    if (!state || !obs || !obs_type || !out) return SCCBF_ERR_ARG;
    k.state = state; k.u_ref = nullptr; k.obs = obs; k.obs_type = obs_type;
    k.u_out = nullptr; k.status = nullptr; k.rows_out = out;
    const long long blocks = (A * M + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffLL) return SCCBF_ERR_ARG;
    hipLaunchKernelGGL((k_partials<T>), dim3((unsigned)blocks), dim3(kBlock), 0,
                       reinterpret_cast<hipStream_t>(stream), k);
    return hipGetLastError() == hipSuccess ? SCCBF_OK : SCCBF_ERR_LAUNCH;
}

template <typename T>
int qp_impl(const sccbf_params* sp, int64_t A, int32_t M, const T* rows, const T* u_nom,
            T* u_out, int32_t* status, void* stream)
{
    KArgs<T> k;
    sccbf_params q;
    if (!sp) return SCCBF_ERR_ARG;
    q = *sp;
    q.dynamics = SCCBF_DYN_DBM;  // unused by the QP stage; keep validation independent of it
    if (!(q.lr > 0.0)) q.lr = 1.0;
    if (!(q.lf + q.lr > 0.0)) q.lf = 1.0;
    q.flags = 0;
    int rc = fill_args<T>(&q, A, M, k);
    if (rc != SCCBF_OK) return rc;
    if (A > 0 && (!u_nom || !u_out || !status)) return SCCBF_ERR_ARG;
    if (A > 0 && M > 0 && !rows) return SCCBF_ERR_ARG;
    k.state = nullptr; k.u_ref = u_nom; k.obs = rows; k.obs_type = nullptr;
    k.u_out = u_out; k.status = status; k.rows_out = nullptr;
    return dispatch<T, true>(&q, k, stream);
}

template <typename T>
void fill_nominal_args(const sccbf_nominal_params* sp, int64_t A, int32_t N, NArgs<T>& k)
{
    k.axle = (T)sp->axle; k.k = (T)sp->k; k.ks = (T)sp->ks;
    k.kp = (T)sp->kp; k.ki = (T)sp->ki; k.kd = (T)sp->kd;
    k.dt = (T)sp->dt; k.inv_dt = (T)(sp->dt > 0.0 ? 1.0 / sp->dt : 0.0);
    k.target_speed = (T)sp->target_speed;
    k.flags = sp->flags;
    k.A = A; k.N = N;
}

template <typename T>
int nominal_impl(const sccbf_nominal_params* sp, int64_t A, int32_t N, const T* state, const T* traj,
                 int32_t* target_idx, T* pid_state, T* u_ref, void* stream)
{
    if (!sp || A < 0 || N < 1 || A > 0x7fffffffLL) return SCCBF_ERR_ARG;
    if (A == 0) return SCCBF_OK;
    if (!state || !traj || !target_idx || !u_ref) return SCCBF_ERR_ARG;
    if (pid_state && !(sp->dt > 0.0)) return SCCBF_ERR_ARG;
    NArgs<T> k;
    fill_nominal_args<T>(sp, A, N, k);
    k.state = state; k.traj = traj; k.target_idx = target_idx; k.pid_state = pid_state; k.u_ref = u_ref;
    const long long per_block = kNomBlock / kNomLanes;
    const long long blocks = (A + per_block - 1) / per_block;
    hipLaunchKernelGGL((k_nominal<T>), dim3((unsigned)blocks), dim3(kNomBlock), 0,
                       reinterpret_cast<hipStream_t>(stream), k);
    return hipGetLastError() == hipSuccess ? SCCBF_OK : SCCBF_ERR_LAUNCH;
}

template <typename T>
int vehicle_step_impl(int64_t A, T* state, const T* u, double lr, double lf, double max_steer, double dt,
                      T* beta_out, void* stream)
{
    if (A < 0 || A > 0x7fffffffLL || !(lr > 0.0) || !(lf + lr > 0.0) || !(max_steer >= 0.0)) return SCCBF_ERR_ARG;
    if (A == 0) return SCCBF_OK;
    if (!state || !u) return SCCBF_ERR_ARG;
    const long long blocks = (A + kBlock - 1) / kBlock;
    hipLaunchKernelGGL((k_vehicle_step<T>), dim3((unsigned)blocks), dim3(kBlock), 0,
                       reinterpret_cast<hipStream_t>(stream), (long long)A, state, u,
                       (T)(lr / (lf + lr)), (T)(1.0 / lr), (T)max_steer, (T)dt, beta_out);
    return hipGetLastError() == hipSuccess ? SCCBF_OK : SCCBF_ERR_LAUNCH;
}

template <typename T, int RPL>
int launch_rollout(const RArgs<T>& r, hipStream_t s)
{
    const long long per_block = kBlock / kNomLanes;
    const long long blocks = (r.f.A + per_block - 1) / per_block;
    if (blocks > 0x7fffffffLL) return SCCBF_ERR_ARG;
    hipLaunchKernelGGL((k_rollout<T, RPL>), dim3((unsigned)blocks), dim3(kBlock), 0, s, r);
    return hipGetLastError() == hipSuccess ? SCCBF_OK : SCCBF_ERR_LAUNCH;
}

template <typename T>
int rollout_impl(const sccbf_params* sp, const sccbf_nominal_params* np, const sccbf_rollout_params* rp,
                 int64_t A, int32_t M, int32_t N, T* state, const T* traj, int32_t* target_idx, T* pid_state,
                 const T* obs, const uint8_t* obs_type, T* h_min, int32_t* n_active, int32_t* n_infeasible,
                 T* trace, void* stream)
{
    if (!np || !rp) return SCCBF_ERR_ARG;
    RArgs<T> r;
    int rc = fill_args<T>(sp, A, M, r.f);
    if (rc != SCCBF_OK) return rc;
    if (N > kTrajChunk || M > 16 * kNomLanes) return SCCBF_ERR_ROWS;
    if (N < 1 || M < 1 || rp->ticks < 0 || !(rp->dt > 0.0) || !(rp->max_steer >= 0.0)) return SCCBF_ERR_ARG;
    if (rp->motion != SCCBF_MOTION_STATIC && rp->motion != SCCBF_MOTION_CONSTANT_VELOCITY) return SCCBF_ERR_ARG;
    if (sp->dynamics != SCCBF_DYN_DBM) return SCCBF_ERR_ARG;  // update_com is the DBM step
    if (sp->flags & SCCBF_FLAG_OBS_SHARED) return SCCBF_ERR_ARG;  // moving obstacles are per agent
    if (A == 0 || rp->ticks == 0) return SCCBF_OK;
    if (!state || !traj || !target_idx || !obs || !obs_type) return SCCBF_ERR_ARG;
    if (pid_state && !(np->dt > 0.0)) return SCCBF_ERR_ARG;
    fill_nominal_args<T>(np, A, N, r.n);
    r.f.state = state; r.f.u_ref = nullptr; r.f.obs = obs; r.f.obs_type = obs_type;
    r.f.u_out = nullptr; r.f.status = nullptr; r.f.rows_out = nullptr;
    r.n.state = state; r.n.traj = traj; r.n.target_idx = target_idx; r.n.pid_state = pid_state;
    r.n.u_ref = state;  // placeholder for the loader's nominal-input read (overwritten every tick)
    r.state = state;
    r.k_in = (T)(sp->lr / (sp->lf + sp->lr)); r.inv_lr = (T)(1.0 / sp->lr);
    r.max_steer = (T)rp->max_steer; r.dt = (T)rp->dt;
    r.ticks = rp->ticks; r.motion = rp->motion;
    r.h_min = h_min; r.n_active = n_active; r.n_infeasible = n_infeasible; r.trace = trace;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int rpl = (M + kNomLanes - 1) / kNomLanes;
    if (rpl <= 1) return launch_rollout<T, 1>(r, s);
    if (rpl <= 2) return launch_rollout<T, 2>(r, s);
    if (rpl <= 4) return launch_rollout<T, 4>(r, s);
    if (rpl <= 8) return launch_rollout<T, 8>(r, s);
    return launch_rollout<T, 16>(r, s);
}

}  // namespace

extern "C" {

int sccbf_version(void) { return SCCBF_VERSION; }

const char* sccbf_strerror(int code)
{
    switch (code) {
    case SCCBF_OK: return "ok";
    case SCCBF_ERR_ARG: return "invalid argument (null pointer, negative size or bad enum)";
    case SCCBF_ERR_ROWS: return "too many constraint rows (M > SCCBF_MAX_ROWS)";
    case SCCBF_ERR_WEIGHT: return "QP weight R is not symmetric positive definite";
    case SCCBF_ERR_LAUNCH: return "kernel launch failed";
    case SCCBF_ERR_VARIANT: return "lanes_per_qp cannot hold M rows (use 0 or 64)";
    case SCCBF_ERR_NO_DEVICE: return "no usable HIP device";
    default: return "unknown error";
    }
}

void sccbf_default_params(sccbf_params* p)
{
    if (!p) return;
    p->alpha = 1.0;
    p->lr = 1.45;
    p->lf = 1.45;
    p->L = 2.9;
    p->R[0] = 1.0; p->R[1] = 0.0; p->R[2] = 0.0; p->R[3] = 1.0;
    p->zero_tol = 1e-3;
    p->dynamics = SCCBF_DYN_DBM;
    p->lanes_per_qp = 0;
    p->flags = 0;
    p->reserved = 0;
}

int sccbf_solve_f32(const sccbf_params* p, int64_t A, int32_t M, const float* state,
                    const float* u_ref, const float* obs, const uint8_t* obs_type, float* u_out,
                    int32_t* status, float* rows_out, void* stream)
{
    return solve_impl<float>(p, A, M, state, u_ref, obs, obs_type, u_out, status, rows_out, stream);
}

int sccbf_solve_f64(const sccbf_params* p, int64_t A, int32_t M, const double* state,
                    const double* u_ref, const double* obs, const uint8_t* obs_type, double* u_out,
                    int32_t* status, double* rows_out, void* stream)
{
    return solve_impl<double>(p, A, M, state, u_ref, obs, obs_type, u_out, status, rows_out, stream);
}

int sccbf_qp_f32(const sccbf_params* p, int64_t A, int32_t M, const float* rows, const float* u_nom,
                 float* u_out, int32_t* status, void* stream)
{
    return qp_impl<float>(p, A, M, rows, u_nom, u_out, status, stream);
}

int sccbf_qp_f64(const sccbf_params* p, int64_t A, int32_t M, const double* rows,
                 const double* u_nom, double* u_out, int32_t* status, void* stream)
{
    return qp_impl<double>(p, A, M, rows, u_nom, u_out, status, stream);
}

int sccbf_partials_f32(const sccbf_params* p, int64_t A, int32_t M, const float* state,
                       const float* obs, const uint8_t* obs_type, float* out, void* stream)
{
    return partials_impl<float>(p, A, M, state, obs, obs_type, out, stream);
}

int sccbf_partials_f64(const sccbf_params* p, int64_t A, int32_t M, const double* state,
                       const double* obs, const uint8_t* obs_type, double* out, void* stream)
{
    return partials_impl<double>(p, A, M, state, obs, obs_type, out, stream);
}

void sccbf_default_nominal_params(sccbf_nominal_params* p)
{
    if (!p) return;
    p->axle = 2.0;
    p->k = 0.5;
    p->ks = 0.01;
    p->kp = 1.0;
    p->ki = 0.0;
    p->kd = 0.0;
    p->dt = 0.1;
    p->target_speed = 30.0 / 3.6;
    p->flags = 0;
    p->reserved = 0;
}

int sccbf_nominal_f32(const sccbf_nominal_params* p, int64_t A, int32_t N, const float* state,
                      const float* traj, int32_t* target_idx, float* pid_state, float* u_ref, void* stream)
{
    return nominal_impl<float>(p, A, N, state, traj, target_idx, pid_state, u_ref, stream);
}

int sccbf_nominal_f64(const sccbf_nominal_params* p, int64_t A, int32_t N, const double* state,
                      const double* traj, int32_t* target_idx, double* pid_state, double* u_ref, void* stream)
{
    return nominal_impl<double>(p, A, N, state, traj, target_idx, pid_state, u_ref, stream);
}

int sccbf_vehicle_step_f32(int64_t A, float* state, const float* u, double lr, double lf, double max_steer,
                           double dt, float* beta_out, void* stream)
{
    return vehicle_step_impl<float>(A, state, u, lr, lf, max_steer, dt, beta_out, stream);
}

int sccbf_vehicle_step_f64(int64_t A, double* state, const double* u, double lr, double lf, double max_steer,
                           double dt, double* beta_out, void* stream)
{
    return vehicle_step_impl<double>(A, state, u, lr, lf, max_steer, dt, beta_out, stream);
}

void sccbf_default_rollout_params(sccbf_rollout_params* p)
{
    if (!p) return;
    p->max_steer = 30.0 * 3.14159265358979323846 / 180.0;
    p->dt = 0.1;
    p->ticks = 0;
    p->motion = SCCBF_MOTION_STATIC;
}

int sccbf_rollout_f32(const sccbf_params* p, const sccbf_nominal_params* np, const sccbf_rollout_params* rp,
                      int64_t A, int32_t M, int32_t N, float* state, const float* traj, int32_t* target_idx,
                      float* pid_state, const float* obs, const uint8_t* obs_type, float* h_min,
                      int32_t* n_active, int32_t* n_infeasible, float* trace, void* stream)
{
    return rollout_impl<float>(p, np, rp, A, M, N, state, traj, target_idx, pid_state, obs, obs_type, h_min,
                               n_active, n_infeasible, trace, stream);
}

int sccbf_rollout_f64(const sccbf_params* p, const sccbf_nominal_params* np, const sccbf_rollout_params* rp,
                      int64_t A, int32_t M, int32_t N, double* state, const double* traj, int32_t* target_idx,
                      double* pid_state, const double* obs, const uint8_t* obs_type, double* h_min,
                      int32_t* n_active, int32_t* n_infeasible, double* trace, void* stream)
{
    return rollout_impl<double>(p, np, rp, A, M, N, state, traj, target_idx, pid_state, obs, obs_type, h_min,
                                n_active, n_infeasible, trace, stream);
}

}  // extern "C"
