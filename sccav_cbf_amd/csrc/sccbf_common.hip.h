// sccbf_common.hip.h - build-time constants, kernel argument block, math shims, cross-lane helpers
// Part of the sccbf library (one translation unit, see sccbf.hip); device code for gfx950.
#pragma once
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
#define SCCBF_ROLLOUT_WAVES 4  // wavefronts per SIMD the fp32 rollout (up to 16 slots per agent) is held to
#endif
#ifndef SCCBF_ROLLOUT_WAVES_F64
#define SCCBF_ROLLOUT_WAVES_F64 2  // and the fp64 one: 256 registers per lane (built without machine LICM, see sccbf.hip)
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
    // vehicle step on the filtered input after the solve (null: none), sccbf_solve_step_*
    T* step_state;
    T* step_beta;
    T step_max_steer, step_dt;
};

// ------------------------------------------------------------------ math shims
// fp32: the hardware sin/cos unit (argument in revolutions, reduced to [-1/2, 1/2] so that any
// magnitude is in its range and a small angle of either sign keeps its relative accuracy: reduced
// to [0, 1) instead, a heading just below zero would sit next to 1 and lose it, 14 % of its sine
// at -1e-6 rad; the angle error is |x| 2^-24, 2e-7 rad for a heading in [-pi, pi]); 1-ulp rcp /
// sqrt / rsq instead of the IEEE division and denormal-safe sqrt sequences.  Library sincosf/tanf would pull a Payne-Hanek reduction with a scratch
// table into every call site.  fp64 keeps the full-accuracy library forms.
__device__ inline void sincos_(float x, float* s, float* c)
{
    const float t = x * 0.15915494309189535f;
    const float r = t - __builtin_rintf(t);
    *s = __builtin_amdgcn_sinf(r);
    *c = __builtin_amdgcn_cosf(r);
}
#ifndef SCCBF_F64_LIBM
#define SCCBF_F64_LIBM 0  // 1: double precision through the device library's sincos / tan / sqrt / division (the
                          // reference point of the forms below; measured 25 % more instructions per launch)
#endif
#if SCCBF_F64_LIBM
__device__ inline void sincos_(double x, double* s, double* c) { sincos(x, s, c); }
__device__ inline double sqrt_(double x) { return sqrt(x); }
__device__ inline double rsqrt_(double x) { return 1.0 / sqrt(x); }
__device__ inline double rcp_(double x) { return 1.0 / x; }
__device__ inline double div_(double a, double b) { return a / b; }
__device__ inline double tan_(double x) { return tan(x); }
#else
// fp64 without the device library's general-purpose forms: those carry a Payne-Hanek branch for huge arguments
// (sincos, tan), scaling against overflow and denormals and a class fix-up (sqrt, division) in every inlined copy.
// Here: the same hardware estimates (v_rcp_f64, v_rsq_f64) refined by Newton steps to the last bit or two, and
// sine / cosine by a three-step Cody-Waite reduction with fused multiply-adds and the classic minimax kernels on
// [-pi/4, pi/4] (relative error 1e-16 for a heading of ordinary size; the reduction keeps an absolute error of
// |x| 2^-87, 1e-11 rad at |x| = 1e15).  Accuracy against the correctly rounded forms: 1-2 ulp; the fp64 parity bar
// of 1e-8 against the oracle is seven orders above that.  0, Inf and NaN: 1 / 0 and 1 / Inf give NaN where IEEE
// gives Inf and 0 (every caller guards its divisor or reports the row as non-finite either way); sqrt_(0) = 0.
__device__ inline double rcp_(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    y = fma(fma(-x, y, 1.0), y, y);
    y = fma(fma(-x, y, 1.0), y, y);
    return y;
}
__device__ inline double div_(double a, double b)
{
    const double y = rcp_(b);
    const double q = a * y;
    return fma(fma(-b, q, a), y, q);  // one correction with the residual: the quotient to the last bit or two
}
__device__ inline double rsqrt_(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    y = fma(y * 0.5, fma(-x * y, y, 1.0), y);
    y = fma(y * 0.5, fma(-x * y, y, 1.0), y);
    return y;
}
__device__ inline double sqrt_(double x)
{
    const double y = rsqrt_(x);
    double g = x * y;
    g = fma(fma(-g, g, x), 0.5 * y, g);
    return x > 0.0 ? g : (x == 0.0 ? 0.0 : __builtin_nan(""));  // rsq(0) is Inf: 0 Inf = NaN
}
__device__ inline void sincos_(double x, double* s, double* c)
{
    // x = k pi/2 + r, |r| <= pi/4: pi/2 in three pieces of 33 bits and the rest, so that k times a piece is exact
    const double k = rint(x * 0.63661977236758134308);
    double r = fma(-k, 1.57079632673412561417e+00, x);
    r = fma(-k, 6.07710050630396597660e-11, r);
    r = fma(-k, 2.02226624871116645580e-21, r);
    r = fma(-k, 8.47842766036889956997e-32, r);
    const double z = r * r;
    double ps = 1.58969099521155010221e-10;
    ps = fma(ps, z, -2.50507602534068634195e-08);
    ps = fma(ps, z, 2.75573137070700676789e-06);
    ps = fma(ps, z, -1.98412698298579493134e-04);
    ps = fma(ps, z, 8.33333333332248946124e-03);
    ps = fma(ps, z, -1.66666666666666324348e-01);
    const double sr = fma(r * z, ps, r);
    double pc = -1.13596475577881948265e-11;
    pc = fma(pc, z, 2.08757232129817482790e-09);
    pc = fma(pc, z, -2.75573143513906633035e-07);
    pc = fma(pc, z, 2.48015872894767294178e-05);
    pc = fma(pc, z, -1.38888888888741095749e-03);
    pc = fma(pc, z, 4.16666666666666019037e-02);
    const double cr = fma(z * z, pc, fma(-0.5, z, 1.0));
    // quadrant k mod 4 (k may be beyond the integers of 32 bits)
    const int q = (int)fma(-4.0, rint(k * 0.25), k) & 3;
    const double ss = (q & 1) ? cr : sr, cc = (q & 1) ? sr : cr;
    *s = (q & 2) ? -ss : ss;
    *c = ((q + 1) & 2) ? -cc : cc;
}
__device__ inline double tan_(double x)
{
    double s, c;
    sincos_(x, &s, &c);
    return div_(s, c);
}
#endif
__device__ inline float sqrt_(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ inline float rsqrt_(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ inline float rcp_(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ inline float div_(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ inline float tan_(float x)
{
    float s, c;
    sincos_(x, &s, &c);
    return s * __builtin_amdgcn_rcpf(c);
}
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
    static constexpr float big = 3.0e38f;      // finite, beyond every bound a row can state
};
template <> struct Lim<double> {
    static constexpr double eps = 2.220446049250313e-16;
    static constexpr double tiny_n2 = 1e-300;
    static constexpr double viol_tol = 1e-12;
    static constexpr double par_tol = 1e-14;
    static constexpr double feas_tol = 1e-10;
    static constexpr double lane_xtol = 1e-8;
    static constexpr double big = 1.0e300;
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
__device__ inline bool viol_key_above(int key, float thr, unsigned idx_mask = 0xFFu)
{
    return __uint_as_float((unsigned)key & ~idx_mask) > thr;
}
__device__ inline bool viol_key_above(int key, double, unsigned idx_mask = 0xFFu) { return key > (int)idx_mask; }

}  // namespace
