// sccbf_ingest.hip.h - in front of the filter: bounding boxes -> obstacle slots, lane points -> lane slots
// Part of the sccbf library (one translation unit, see sccbf.hip); device code for gfx950.
#pragma once
#include "sccbf_rows.hip.h"

namespace {

template <typename T> struct IArgs {
    long long A;
    int M;
    unsigned flags;      // SCCBF_FLAG_TYPE_PER_AGENT (kind), SCCBF_FLAG_OBS_TILED / _AGENT_MAJOR (layout of obs)
    const T* bbox;       // [M][2][A][4] = loc_x, loc_y, ext_x, ext_y | yaw, speed, -, -
    const uint8_t* kind; // [M] or [M][A]
    T buffer;
    T* obs;
};

// index of the first 4-vector of slot m of agent a in obs, and the distance to the second
template <typename T> __device__ inline ObsAddr slot_addr_(long long A, int M, unsigned flags, long long a)
{
    KArgs<T> k;
    k.A = A; k.M = M; k.flags = flags & (SCCBF_FLAG_OBS_TILED | SCCBF_FLAG_OBS_AGENT_MAJOR);
    return obs_addr_(k, a);
}

// One thread per (slot, agent).  Ellipse2D.from_bounding_box, reference cbf/obstacles.py:320-331
// (a = extent.x, b = extent.y, centre, theta = yaw, buffer on both axes, no velocity);
// CollisionCone2D.from_bounding_box, 534-543 (a = hypot(extent.x, extent.y) + buffer,
// s_obs = [x, y, 0, velocity]).  Any other kind leaves an all-zero slot.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_ingest(const IArgs<T> p)
{
    using v4 = typename VT<T>::v4;
    const long long gid = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= p.A * p.M) return;
    const int m = (int)(gid / p.A);
    const long long a = gid % p.A;
    const v4* in = reinterpret_cast<const v4*>(p.bbox);
    const v4 b0 = in[(long long)m * 2 * p.A + a];
    const v4 b1 = in[(long long)m * 2 * p.A + p.A + a];
    const int kind = (p.flags & SCCBF_FLAG_TYPE_PER_AGENT) ? p.kind[(long long)m * p.A + a] : p.kind[m];
    v4 oa = v4{T(0), T(0), T(0), T(0)}, ob = oa;
    if (kind == SCCBF_OBS_ELLIPSE) {
        oa = v4{b0.x, b0.y, b0.z + p.buffer, b0.w + p.buffer};
        ob = v4{b1.x, T(0), T(0), T(0)};
    } else if (kind == SCCBF_OBS_CONE) {
        oa = v4{b0.x, b0.y, T(0), b1.y};
        ob = v4{box_cone_radius_(b0.z, b0.w, p.buffer), T(0), T(0), T(0)};
    } else if (kind == SCCBF_OBS_ELLIPSE_PREP) {  // the same ellipse, constants evaluated (in double)
        const double a = (double)(b0.z + p.buffer), b = (double)(b0.w + p.buffer);
        oa = v4{b0.x, b0.y, (T)(1.0 / (a * a)), (T)(1.0 / (b * b))};
        ob = v4{(T)cos((double)b1.x), (T)sin((double)b1.x), T(0), T(0)};
    } else if (kind == SCCBF_OBS_CONE_PREP) {     // theta_o = 0, beta = 0: the velocity is (speed, 0)
        oa = v4{b0.x, b0.y, b1.y, T(0)};
        ob = v4{box_cone_radius_(b0.z, b0.w, p.buffer), T(0), T(0), T(0)};
    }
    const ObsAddr ad = slot_addr_<T>(p.A, p.M, p.flags, a);
    v4* out = reinterpret_cast<v4*>(p.obs);
    out[ad.at(m)] = oa;
    out[ad.at(m) + ad.half] = ob;
}

// As-written ellipse / cone slots -> prepared slots, in place; one thread per (slot, agent).  The
// constants are evaluated in double whatever T is: this runs once per obstacle update, not per tick.
template <typename T> struct PArgs {
    long long A;
    int M;
    unsigned flags;
    T* obs;
    uint8_t* type;
};

template <typename T>
__global__ __launch_bounds__(kBlock) void k_prepare_obs(const PArgs<T> p)
{
    using v4 = typename VT<T>::v4;
    const long long gid = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= p.A * p.M) return;
    const int m = (int)(gid / p.A);
    const long long a = gid % p.A;
    const int kind = (p.flags & SCCBF_FLAG_TYPE_PER_AGENT) ? p.type[(long long)m * p.A + a] : p.type[m];
    if (kind != SCCBF_OBS_ELLIPSE && kind != SCCBF_OBS_CONE) return;
    KArgs<T> k;
    k.A = p.A; k.M = p.M; k.flags = p.flags & (SCCBF_FLAG_OBS_TILED | SCCBF_FLAG_OBS_SHARED | SCCBF_FLAG_OBS_AGENT_MAJOR);
    const ObsAddr ad = obs_addr_(k, a);
    v4* obs = reinterpret_cast<v4*>(p.obs);
    v4 oa = obs[ad.at(m)], ob = obs[ad.at(m) + ad.half];
    if (kind == SCCBF_OBS_ELLIPSE) {
        const double ea = (double)oa.z, eb = (double)oa.w, th = (double)ob.x;
        oa.z = (T)(1.0 / (ea * ea));
        oa.w = (T)(1.0 / (eb * eb));
        ob = v4{(T)cos(th), (T)sin(th), ob.y, ob.z};
    } else {
        const double ang = (double)oa.z + (double)ob.y, vo = (double)oa.w;
        oa.z = (T)(vo * cos(ang));
        oa.w = (T)(vo * sin(ang));
    }
    obs[ad.at(m)] = oa;
    obs[ad.at(m) + ad.half] = ob;
}

// after k_prepare_obs (same stream): the kinds of the slots it rewrote
__global__ __launch_bounds__(kBlock) void k_prepare_types(long long n, uint8_t* type)
{
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int k = type[i];
    if (k == SCCBF_OBS_ELLIPSE) type[i] = SCCBF_OBS_ELLIPSE_PREP;
    if (k == SCCBF_OBS_CONE) type[i] = SCCBF_OBS_CONE_PREP;
}

template <typename T> struct FArgs {
    long long A;
    int K, degree, m, M;
    unsigned flags;
    const T* px;     // [K][A]
    const T* py;     // [K][A]
    const T* sigma;  // [K][A] or null (every point 10, reference cbf/obstacles.py:737-738)
    T buffer;
    T* obs;
};

// This lane's usable points of agent a, 32 / G ... 8 loads in flight at a time.
template <int kFitLanes, typename T, typename F>
__device__ inline void for_points_(const FArgs<T>& p, long long a, int g, F&& fn)
{
    constexpr int C = kFitLanes == 1 ? 8 : 4;
    for (int k0 = g; k0 < p.K; k0 += C * kFitLanes) {
        T xs[C], ys[C], ss[C];
#pragma unroll
        for (int i = 0; i < C; ++i) {
            const int k = min(k0 + i * kFitLanes, p.K - 1);
            xs[i] = p.px[(long long)k * p.A + a];
            ys[i] = p.py[(long long)k * p.A + a];
            ss[i] = p.sigma ? p.sigma[(long long)k * p.A + a] : T(10);  // reference cbf/obstacles.py:737-738
        }
#pragma unroll
        for (int i = 0; i < C; ++i) {
            const double x = (double)xs[i], y = (double)ys[i], sg = (double)ss[i];
            if (k0 + i * kFitLanes >= p.K || !(sg > 0.0) || !isfinite(sg) || !isfinite(x) || !isfinite(y)) continue;
            fn(x, y, sg);
        }
    }
}

// sum / min / max over the lanes of an agent; every lane ends with the same bits
__device__ inline double fit_shfl_(double v, int m)
{
    const long long b = __double_as_longlong(v);
    const int lo = __shfl_xor((int)b, m, 64), hi = __shfl_xor((int)(b >> 32), m, 64);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
template <int G> __device__ inline double fit_sum_(double v)
{
    if (G >= 2) v += fit_shfl_(v, 1);
    if (G >= 4) v += fit_shfl_(v, 2);
    return v;
}

// PolyLane.fit_polynomial_curve, reference cbf/obstacles.py:715-773: the polynomial of the given
// degree that minimises sum(((p(x_i) - y_i) / sigma_i)^2) (what scipy.optimize.curve_fit minimises
// there for a model that is linear in its coefficients).  The lanes of an agent share its points
// (partial sums meet by shuffles after every pass); the arithmetic is fp64 throughout.
// The fit is built on the polynomials orthogonal under the points' own weighted inner product
// (three-term recurrence in t = (x - centre) / half-range, one pass over the points per degree),
// whose error grows with the condition of the problem and not with its square as the normal
// equations' does.  The expansion coefficients are then corrected once from the residuals of the
// first fit (a fixed point weighing 1e6 times its neighbours makes every level of the basis cancel
// against it, which costs the first fit up to eleven digits and the corrected one none that
// matter); the expansion gives the coefficients of the powers of x.  Points with
// sigma <= 0 or a non-finite value are skipped; if the points do not determine the polynomial
// (fewer distinct abscissae than coefficients: the next orthogonal polynomial vanishes on all of
// them, up to its own rounding) the coefficients are NaN and the filter reports the agent as
// non-finite.
// N = degree + 1 coefficients (every loop over them unrolls, nothing is indexed at run time);
// kFitLanes lanes per agent (lane g takes the points g, g + kFitLanes, ...); kHeld points per lane
// stay in registers for all passes (the host launches this form for K <= kHeld * kFitLanes), or with
// kHeld = 0 the lists are read again on every pass.
template <typename T, int N, int kFitLanes, int kHeld>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(kHeld == 16 ? 2 : 3, 8))) void k_lane_fit(const FArgs<T> p)
{
    using v4 = typename VT<T>::v4;
    static_assert(kFitLanes == 1 || kFitLanes == 4, "none or two shuffle rounds");
    static_assert(N >= 1 && N <= 4, "constant to cubic");
    const int g = threadIdx.x % kFitLanes;
    const long long a = (long long)blockIdx.x * (kBlock / kFitLanes) + threadIdx.x / kFitLanes;
    if (a >= p.A) return;  // the four lanes of an agent leave together
    // a held point that does not count has w = 0
    constexpr bool held = kHeld > 0;
    constexpr int kSlots = held ? kHeld : 1;
    double ht[kSlots], hy[kSlots], hw[kSlots];
    double xmin = INFINITY, xmax = -INFINITY;
    int cnt = 0;
    if constexpr (held) {
#pragma unroll
        for (int i = 0; i < kSlots; ++i) {
            const int k = g + i * kFitLanes, kk = min(k, p.K - 1);
            const double x = (double)p.px[(long long)kk * p.A + a], y = (double)p.py[(long long)kk * p.A + a];
            const double sg = p.sigma ? (double)p.sigma[(long long)kk * p.A + a] : 10.0;
            const bool use = k < p.K && sg > 0.0 && isfinite(sg) && isfinite(x) && isfinite(y);
            ht[i] = x;
            hy[i] = use ? y : 0.0;
            hw[i] = use ? 1.0 / (sg * sg) : 0.0;
            xmin = use ? fmin(xmin, x) : xmin;
            xmax = use ? fmax(xmax, x) : xmax;
            cnt += use;
        }
    } else {
        for_points_<kFitLanes>(p, a, g, [&](double x, double, double) {
            xmin = fmin(xmin, x);
            xmax = fmax(xmax, x);
            ++cnt;
        });
    }
    if (kFitLanes == 4) {
        xmin = fmin(xmin, fit_shfl_(xmin, 1)); xmin = fmin(xmin, fit_shfl_(xmin, 2));
        xmax = fmax(xmax, fit_shfl_(xmax, 1)); xmax = fmax(xmax, fit_shfl_(xmax, 2));
        cnt += __shfl_xor(cnt, 1, 64); cnt += __shfl_xor(cnt, 2, 64);
    }
    const double xc = 0.5 * (xmin + xmax);
    double hs = 0.5 * (xmax - xmin);
    if (!(hs > 0.0)) hs = 1.0;
    const double ihs = 1.0 / hs;
    if constexpr (held) {
#pragma unroll
        for (int i = 0; i < kSlots; ++i) ht[i] = hw[i] != 0.0 ? (ht[i] - xc) * ihs : 0.0;
    }
    auto each = [&](auto&& fn) {  // fn(t, y, w) for this lane's points
        if constexpr (held) {
#pragma unroll
            for (int i = 0; i < kSlots; ++i) fn(ht[i], hy[i], hw[i]);
        } else {
            for_points_<kFitLanes>(p, a, g, [&](double x, double y, double sg) { fn((x - xc) * ihs, y, 1.0 / (sg * sg)); });
        }
    };
    // P_0 = 1, P_{j+1} = (t - al_j) P_j - be_j P_{j-1};  fit = sum_j cf_j P_j
    double al[N], be[N], cf[N], nrm[N];
    double prev_norm = 1.0;
    bool ok = cnt >= N;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double nn = 0.0, ta = 0.0, mm = 0.0;
        each([&](double t, double, double w) {
            double pm = 0.0, pj = 1.0, qm = 0.0, qj = 1.0;  // P_{i-1}, P_i and bounds of the terms they sum
#pragma unroll
            for (int i = 0; i < j; ++i) {
                const double pn = (t - al[i]) * pj - be[i] * pm;
                const double qn = (fabs(t) + fabs(al[i])) * qj + be[i] * qm;
                pm = pj; pj = pn; qm = qj; qj = qn;
            }
            nn += w * pj * pj;
            ta += w * t * pj * pj;
            mm += w * qj * qj;
        });
        nn = fit_sum_<kFitLanes>(nn); ta = fit_sum_<kFitLanes>(ta); mm = fit_sum_<kFitLanes>(mm);
        // P_j is its own rounding error on every point: the points have fewer than j + 1 distinct abscissae
        if (!(nn > 1e-26 * mm)) { ok = false; nn = 1.0; }
        al[j] = ta / nn;
        be[j] = j ? nn / prev_norm : 0.0;
        nrm[j] = nn;
        cf[j] = 0.0;
        prev_norm = nn;
    }
    // cf_j = <y, P_j> / <P_j, P_j>, then once more with the residuals y - fit in place of y
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        double gs[N];
#pragma unroll
        for (int j = 0; j < N; ++j) gs[j] = 0.0;
        each([&](double t, double y, double w) {
            double P[N];
            P[0] = 1.0;
#pragma unroll
            for (int j = 1; j < N; ++j) P[j] = (t - al[j - 1]) * P[j - 1] - (j > 1 ? be[j - 1] * P[j > 1 ? j - 2 : 0] : 0.0);
            double r = y;
            if (pass) {
#pragma unroll
                for (int j = 0; j < N; ++j) r -= cf[j] * P[j];
            }
            const double wr = w * r;
#pragma unroll
            for (int j = 0; j < N; ++j) gs[j] += wr * P[j];
        });
#pragma unroll
        for (int j = 0; j < N; ++j) cf[j] += fit_sum_<kFitLanes>(gs[j]) / nrm[j];
    }
    // coefficients of the powers of t, then of x:  p(x) = sum_j d_j ((x - xc) / hs)^j
    double Pm[N], Pj[N], d[N];
#pragma unroll
    for (int k = 0; k < N; ++k) { Pm[k] = 0.0; Pj[k] = k ? 0.0 : 1.0; d[k] = 0.0; }
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double Pn[N];
#pragma unroll
        for (int k = 0; k < N; ++k) {
            d[k] += cf[j] * Pj[k];
            Pn[k] = (k ? Pj[k ? k - 1 : 0] : 0.0) - al[j] * Pj[k] - be[j] * Pm[k];
        }
#pragma unroll
        for (int k = 0; k < N; ++k) { Pm[k] = Pj[k]; Pj[k] = Pn[k]; }
    }
    double c[4] = {0, 0, 0, 0};
    constexpr double binom[4][4] = {{1, 0, 0, 0}, {1, 1, 0, 0}, {1, 2, 1, 0}, {1, 3, 3, 1}};
    double sc = 1.0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const double e = d[j] * sc;
        sc *= ihs;
        double pw = 1.0;  // (-xc)^(j - k), k from j down to 0
#pragma unroll
        for (int k = j; k >= 0; --k) {
            c[k] += e * binom[j][k] * pw;
            pw *= -xc;
        }
    }
    if (!ok) c[0] = c[1] = c[2] = c[3] = NAN;
    if (g != 0) return;
    const ObsAddr ad = slot_addr_<T>(p.A, p.M, p.flags, a);
    v4* out = reinterpret_cast<v4*>(p.obs);
    out[ad.at(p.m)] = v4{(T)c[0], (T)c[1], (T)c[2], (T)c[3]};
    out[ad.at(p.m) + ad.half] = v4{p.buffer, T(0), T(0), T(0)};
}

}  // namespace
