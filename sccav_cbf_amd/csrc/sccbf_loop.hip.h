// sccbf_loop.hip.h - nominal controller, vehicle step and the closed-loop rollout kernel
// Part of the sccbf library (one translation unit, see sccbf.hip); device code for gfx950.
#pragma once
#include "sccbf_ext.hip.h"

namespace {

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
constexpr int kSup = 4;               // runs per outer bounding box (128 way-points)
constexpr int kSups = kSubs / kSup;
constexpr int kNomLanes = 4;          // lanes per agent in the nominal and rollout kernels
constexpr int kNomBlock = 1024;       // threads per workgroup of the nominal kernel: 256 agents share one staging

template <typename T> __device__ inline T wrap_pi_(T a)
{
    // reference cbf/utils.py:93-106 subtracts / adds 2 pi until the angle is in [-pi, pi]
    return wrap_angle_(a);
}

// One pass (up to 2048 way-points) of the course in LDS: coordinates in runs of 32, the
// bounding box of every run and the box of every four runs.
template <typename T> struct CoursePass {
    T x[kSubs * kSubStride], y[kSubs * kSubStride];
    T x0[kSubs], x1[kSubs], y0[kSubs], y1[kSubs];
    T X0[kSups], X1[kSups], Y0[kSups], Y1[kSups];
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
    static_assert(kPer * BLOCK == kTrajChunk && kThr * kPer == kSub && kThr * kSup <= 64, "block size");
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
    // the four runs of an outer box sit in kSup * kThr neighbouring lanes of one wavefront
#pragma unroll
    for (int m = kThr; m < kThr * kSup; m *= 2) {
        x0 = min_(x0, __shfl_xor(x0, m, 64)); x1 = max_(x1, __shfl_xor(x1, m, 64));
        y0 = min_(y0, __shfl_xor(y0, m, 64)); y1 = max_(y1, __shfl_xor(y1, m, 64));
    }
    if (threadIdx.x % (kThr * kSup) == 0) {
        c.X0[r / kSup] = x0; c.X1[r / kSup] = x1; c.Y0[r / kSup] = y0; c.Y1[r / kSup] = y1;
    }
}

// Squared length of (ex, ey), in ONE fixed sequence of roundings (ey * ey, then a fused multiply-add)
// for way-points and boxes alike.  With it a box's distance is never above the distance of a
// way-point inside the box, in floating point as in exact arithmetic: subtraction, square and the
// fused sum are monotone, so no slack is needed between the two and equal values mean ties.
template <typename T> __device__ inline T len2_(T ex, T ey)
{
#pragma clang fp contract(off)
    const T ey2 = ey * ey;
    return __builtin_fma(ex, ex, ey2);
}
template <> __device__ inline float len2_<float>(float ex, float ey)
{
#pragma clang fp contract(off)
    const float ey2 = ey * ey;
    return __builtin_fmaf(ex, ex, ey2);
}

// Squared distance from (fx, fy) to a bounding box (infinite for the box of an empty run).
template <typename T> __device__ inline T box_d2_(T x0, T x1, T y0, T y1, T fx, T fy)
{
    const T dx = max_(max_(x0 - fx, fx - x1), T(0));
    const T dy = max_(max_(y0 - fy, fy - y1), T(0));
    return len2_(dx, dy);
}

// Can a box at squared distance d2 whose way-points start at index i0 still change the answer?
// lim: least squared distance known, ai: some way-point with an index <= ai lies at that distance
// (INT_MAX: none known).
// Ties go to the lowest index, so a box that can only tie behind that index is out (the box that
// holds it stays in: the answer is taken from the way-points scanned).
template <typename T> __device__ inline bool box_in_(T d2, int i0, T lim, int ai)
{
    return d2 < lim || (d2 == lim && i0 <= ai);
}
// fp32: squared distances are not negative, so their bit patterns order as they do, and the pair
// (distance, index) compares as one 64-bit integer
__device__ inline unsigned long long dist_key_(float d2, int i)
{
    return ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)i;
}
template <> __device__ inline bool box_in_<float>(float d2, int i0, float lim, int ai)
{
    return dist_key_(d2, i0) <= dist_key_(lim, ai);
}
// is (m, i) a better bound than (lim, ai)?
template <typename T> __device__ inline bool bound_lt_(T m, int i, T lim, int ai)
{
    return m < lim || (m == lim && i < ai);
}
template <> __device__ inline bool bound_lt_<float>(float m, int i, float lim, int ai)
{
    return dist_key_(m, i) < dist_key_(lim, ai);
}

// Nearest way-point of one staged pass to (fx, fy), by the four lanes of a group.  lim is the
// best squared distance known so far (on entry: any upper bound of the answer) and ai an index at or
// below which a way-point is known to lie at that distance (INT_MAX if none is known).  First the group lists the runs that can hold
// the answer: the outer boxes that can still change it (four per lane and step), then the run boxes
// inside those (one per lane and outer box); then it scans the listed runs in index order, eight
// way-points per lane, tightening lim as it goes and skipping runs that have fallen behind the
// bound.  An agent so far from the course that every way-point lies at the same computed distance
// scans one run: every other box can only tie with a higher index and fails its test.
// best / bi: this lane's nearest way-point so far (strict "<" in index order, so the first index
// wins a tie).  Every group works through its own list and the groups of a wavefront scan their
// k-th runs together: a wavefront pays for the longest list among its groups, not for their union.
template <typename T>
__device__ inline void nearest_in_pass_(const CoursePass<T>& c, T fx, T fy, int g, int base, int runs,
                                        T& lim, int& ai, T& best, int& bi)
{
    constexpr int G = kNomLanes;
    static_assert(kSup == G && kSups <= 32 && kSubs <= 64, "one run box per lane inside an outer box; masks");
    const int sups = (runs + kSup - 1) / kSup;
    // Every outer box holds a way-point, so the answer is no farther than the far corner of any of
    // them: for an agent that has left the course far behind (its last target says nothing any
    // more) this keeps the list to the stretch of the course that faces it.
    constexpr int kMine = kSups / G;  // outer boxes per lane
    T d2s[kMine];
    T far = INFINITY;
#pragma unroll
    for (int i = 0; i < kMine; ++i) {
        const int so = i * G + g;  // boxes of runs past the end are empty: infinitely far
        const T X0 = c.X0[so], X1 = c.X1[so], Y0 = c.Y0[so], Y1 = c.Y1[so];
        d2s[i] = so < sups ? box_d2_(X0, X1, Y0, Y1, fx, fy) : T(INFINITY);
        const T ex = max_(fx - X0, X1 - fx), ey = max_(fy - Y0, Y1 - fy);
        far = min_(far, so < sups ? len2_(ex, ey) : T(INFINITY));
    }
    far = min_(far, __shfl_xor(far, 1, 64));
    far = min_(far, __shfl_xor(far, 2, 64));
    ai = far < lim ? 0x7fffffff : ai;  // some way-point lies within `far`, which one is not known
    lim = min_(lim, far);
    unsigned smask = 0u;
#pragma unroll
    for (int i = 0; i < kMine; ++i)
        smask |= box_in_(d2s[i], base + (i * G + g) * kSup * kSub, lim, ai) ? 1u << (i * G + g) : 0u;
    smask = group_or<G>(smask);
    unsigned rlo = 0u, rhi = 0u;  // bit r: run r is listed
    while (smask != 0u) {
        const int rr = (__ffs(smask) - 1) * kSup + g;
        smask &= smask - 1u;
        const bool in = box_in_(box_d2_(c.x0[rr], c.x1[rr], c.y0[rr], c.y1[rr], fx, fy), base + rr * kSub, lim, ai);
        rlo |= (in && rr < 32) ? 1u << rr : 0u;
        rhi |= (in && rr >= 32) ? 1u << (rr - 32) : 0u;
    }
    unsigned long long rmask = ((unsigned long long)group_or<G>(rhi) << 32) | group_or<G>(rlo);
    const T listed = lim;  // rmask, lim and ai are the same in the four lanes of a group
    const int listed_ai = ai;
    while (rmask != 0ull) {
        const int r = __ffsll((long long)rmask) - 1;
        rmask &= rmask - 1ull;
        // the bound may have moved since the boxes were tested
        if ((lim < listed || ai < listed_ai) &&
            !box_in_(box_d2_(c.x0[r], c.x1[r], c.y0[r], c.y1[r], fx, fy), base + r * kSub, lim, ai))
            continue;
#pragma unroll
        for (int j = 0; j < kSub / G; ++j) {  // lane g: way-points 8 g .. 8 g + 7 of the run
            const int jj = g * (kSub / G) + j;
            const T d = len2_(fx - c.x[r * kSubStride + jj], fy - c.y[r * kSubStride + jj]);
            const bool lt = d < best;
            best = lt ? d : best;
            bi = lt ? base + r * kSub + jj : bi;
        }
        // tighten the bound with the group's best so far
        T m = best;
        m = min_(m, __shfl_xor(m, 1, 64));
        m = min_(m, __shfl_xor(m, 2, 64));
        // Runs are scanned in index order and every scanned run that reached the bound has pulled ai
        // down to its own end, so a way-point at distance m <= lim with the end of this run below ai
        // lies in this run: its end bounds the index of a way-point at the new lim.
        const int rend = base + r * kSub + kSub - 1;
        const bool upd = bound_lt_(m, rend, lim, ai);
        lim = upd ? m : lim;
        ai = upd ? rend : ai;
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
                                    T yaw_des, T v_des, bool pid, typename VT<T>::v2& ps, T& acc, T& delta,
                                    T* v_track = nullptr)
{
    // cross-track error projected on the axle direction (-cos(yaw + pi/2), -sin(yaw + pi/2)) = (sin yaw, -cos yaw)
    const T err = (fx - xn) * sth - (fy - yn) * cth;
    const T theta_e = wrap_pi_(yaw_des - yaw);
    const T theta_d = atan2_(p.k * err, v + p.ks);
    delta = theta_e + theta_d;
    const T vt = (p.flags & SCCBF_NOMINAL_WAYPOINT_SPEED) ? v_des : p.target_speed;
    if (v_track) *v_track = vt;
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
    int ai = min(max(last, 0), p.N - 1);  // a way-point known to lie at distance lim
    {
        const v2 wl = *reinterpret_cast<const v2*>(traj4 + ai);
        lim = len2_(fx - wl.x, fy - wl.y);
    }
    T best = INFINITY;
    int bi = 0x7fffffff;
    for (int base = 0; base < p.N; base += kTrajChunk) {
        const int n = min(kTrajChunk, p.N - base);
        __syncthreads();
        stage_course_<T, kNomBlock>(c, traj4, base, n);
        __syncthreads();
        nearest_in_pass_(c, fx, fy, g, base, (n + kSub - 1) / kSub, lim, ai, best, bi);
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

// ------------------------------------------------------- vehicle step (update_com): vehicle_step_ in sccbf_rows.hip.h
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

template <typename T>
__global__ __launch_bounds__(kBlock) void k_vehicle_step_kin(long long A, T* state, const T* u, int mode, T inv_L,
                                                             T max_steer, T dt)
{
    using v4 = typename VT<T>::v4;
    using v2 = typename VT<T>::v2;
    const long long a = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (a >= A) return;
    v4 s = reinterpret_cast<v4*>(state)[a];
    const v2 in = reinterpret_cast<const v2*>(u)[a];
    kin_step_<T>(s, mode, in.x, in.y, inv_L, max_steer, dt);
    reinterpret_cast<v4*>(state)[a] = s;
}

// ---------------------------------------------------- closed-loop rollout in one launch
template <typename T> struct RArgs {
    KArgs<T> f;   // filter: parameters, obs, obs_type, M, flags (state / u_ref / u_out / status unused)
    NArgs<T> n;   // nominal: parameters, traj, N, target_idx (in/out), pid_state (in/out)
    T* state;     // [A][4] in/out
    T k_in, inv_lr, max_steer, dt, inv_dt;
    T seek_gain, seek_v_min;
    T* slip;      // [A][2] in/out, SADBM only: beta, beta_ref_last
    int ticks;
    int motion;   // SCCBF_MOTION_*
    T* h_min;     // [A] optional
    int* n_active;      // [A] optional
    int* n_infeasible;  // [A] optional
    T* trace;     // [ticks][A][4] optional
};

// obstacle slots move between ticks (constant velocity): ellipse by (vx, vy), collision cone by
// v_o along theta_o + beta, radial by (cdot_x, cdot_y); lanes, empty slots and ellipse boxes stay
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
    } else if (type == SCCBF_OBS_ELLIPSE_PREP) {
        oa.x += ob.z * dt;
        oa.y += ob.w * dt;
    } else if (type == SCCBF_OBS_CONE_PREP) {
        oa.x += oa.z * dt;
        oa.y += oa.w * dt;
    } else if (type == SCCBF_OBS_CONE_BOX) {  // theta_o = 0: along x at the box's speed; an ellipse box is at rest
        oa.x += ob.y * dt;
    }
}

// seeker law, reference test_scripts/radial_dynamic_obstacles.py:210-234: turn toward the ego at
// (x, y), speed max(gain * distance, v_min) along the new yaw, then move.  cos/sin of
// atan2(dy, dx) are dx/d and dy/d; atan2(0, 0) = 0 gives the direction (1, 0).
template <typename T>
__device__ inline void seek_slot_(int type, typename VT<T>::v4& oa, typename VT<T>::v4& ob, T x, T y, T gain,
                                  T v_min, T dt)
{
    if (type != SCCBF_OBS_ELLIPSE && type != SCCBF_OBS_RADIAL && type != SCCBF_OBS_ELLIPSE_PREP) {
        advance_slot_<T>(type, oa, ob, dt);
        return;
    }
    const T dx = x - oa.x, dy = y - oa.y;
    const T d2 = dx * dx + dy * dy;
    const T inv = d2 > T(0) ? rsqrt_(d2) : T(0);
    const T v = max_(gain * (d2 * inv), v_min);
    const T vx = d2 > T(0) ? v * dx * inv : v, vy = v * dy * inv;
    if (type == SCCBF_OBS_ELLIPSE) {
        ob.x = atan2_(dy, dx);
        ob.y = vx;
        ob.z = vy;
    } else if (type == SCCBF_OBS_ELLIPSE_PREP) {  // the new heading's cosine and sine, directly
        ob.x = d2 > T(0) ? dx * inv : T(1);
        ob.y = dy * inv;
        ob.z = vx;
        ob.w = vy;
    } else {
        ob.x = vx;
        ob.y = vy;
    }
    oa.x += vx * dt;
    oa.y += vy * dt;
}

// `ticks` control ticks of nominal controller -> filter -> vehicle step (-> obstacle motion)
// for every agent in ONE launch: four lanes per agent, the agent's state, its obstacle slots
// (RPL per lane), target index, PID and slip state stay in registers, the course and its run boxes
// are staged in LDS (once per workgroup when it has at most 2048 way-points, else pass by pass on
// every tick), the rows of each tick go through the per-wavefront LDS table of the 4-lane filter
// kernel.  No HBM traffic inside the loop except the optional trace (and the course passes of a
// long course).
// EXT = false: the bicycle models whose rows lane_rows_ forms (DBM: u = (a, delta), vehicle step
// update_com; KBM: u = (v, delta) with v the tracked speed, step update_by_vel);
// EXT = true: the models of k_group_ext (DUM: u = (a, omega), omega_ref = v tan(delta_ref) / L, unicycle
// step; SADBM: u = (a, delta) in and out, slip state per agent, every cone takes the agent's beta, step
// update_com with the filter's delta).
// G lanes hold an agent's slots and solve its QP: 4 (up to 64 slots) or a whole wavefront (up to 1024);
// the nearest way-point is always searched by groups of four lanes (with G = 64 the sixteen groups of
// a wavefront do the same search and find the same answer).
template <typename T, int RPL, bool EXT, int G = kNomLanes>
__global__ __launch_bounds__(kBlock)
__attribute__((amdgpu_waves_per_eu(
    (RPL <= 4 && G == kNomLanes) ? (sizeof(T) == 4 ? SCCBF_ROLLOUT_WAVES : SCCBF_ROLLOUT_WAVES_F64) : 1, SCCBF_ROLLOUT_WAVES)))
void k_rollout(const RArgs<T> a)
{
    using v4 = typename VT<T>::v4;
    using v2 = typename VT<T>::v2;
    static_assert(G == kNomLanes || G == 64, "lanes per agent");
    constexpr int kMpad = G * RPL + 1;
    constexpr int kPlane = (kBlock / G) * kMpad;
    __shared__ CoursePass<T> c;
    __shared__ T s_yaw[kTrajChunk];
    __shared__ T s_rows[3 * kPlane];
    const KArgs<T>& p = a.f;
    const NArgs<T>& np = a.n;
    const int g = threadIdx.x % G;
    const int gn = threadIdx.x % kNomLanes;  // lane of the nominal controller's group of four
    long long ag = (long long)blockIdx.x * (kBlock / G) + threadIdx.x / G;
    const bool live = ag < p.A;
    if (!live) ag = p.A - 1;  // stays for the barrier; stores are masked
    const int qslot = (int)(threadIdx.x / G) * kMpad;
    const v4* traj4 = reinterpret_cast<const v4*>(np.traj);

    const bool course = np.N > 0;            // else the nominal input is zero
    const bool one_pass = np.N <= kTrajChunk;  // the whole course fits the staging area
    if (course && one_pass) {
        stage_course_<T, kBlock>(c, traj4, 0, np.N);
        for (int i = threadIdx.x; i < np.N; i += kBlock) s_yaw[i] = traj4[i].z;
    }

    // this lane's obstacle slots (r G + g) and everything the agent carries from tick to tick
    LaneLoads<T, RPL> L;
    {
        KArgs<T> q = p;
        q.state = a.state;
        q.u_ref = a.state;  // any readable (A, 2) array: overwritten below before use
        group_load_<T, RPL, false, false>(q, ag, g, G, L);
    }
    v4 st = L.sv;
    int tgt = course ? np.target_idx[ag] : 0;
    v2 ps = v2{T(0), T(0)};
    if (course && np.pid_state) ps = reinterpret_cast<v2*>(np.pid_state)[ag];
    const bool sadbm = EXT && p.dynamics == SCCBF_DYN_SADBM;
    v2 sl = v2{T(0), T(0)};  // SADBM: beta, beta_ref_last
    if (sadbm) sl = reinterpret_cast<const v2*>(a.slip)[ag];
    T h_min = INFINITY;
    int n_act = 0, n_inf = 0;
    __syncthreads();

    for (int tick = 0; tick < a.ticks; ++tick) {
        // nominal input
        T acc = T(0), delta = T(0), v_track = T(0);
        if (course) {
            T sth, cth;
            sincos_(st.z, &sth, &cth);
            const T fx = st.x + np.axle * cth, fy = st.y + np.axle * sth;
            const int lt = min(max(tgt, 0), np.N - 1);
            T lim, best = INFINITY;
            int ai = lt, bi = 0x7fffffff;
            if (one_pass) {
                const T lx = c.x[(lt / kSub) * kSubStride + lt % kSub], ly = c.y[(lt / kSub) * kSubStride + lt % kSub];
                lim = len2_(fx - lx, fy - ly);
                nearest_in_pass_(c, fx, fy, gn, 0, (np.N + kSub - 1) / kSub, lim, ai, best, bi);
            } else {
                const v2 wl = *reinterpret_cast<const v2*>(traj4 + lt);
                lim = len2_(fx - wl.x, fy - wl.y);
                for (int base = 0; base < np.N; base += kTrajChunk) {
                    const int n = min(kTrajChunk, np.N - base);
                    __syncthreads();
                    stage_course_<T, kBlock>(c, traj4, base, n);
                    __syncthreads();
                    nearest_in_pass_(c, fx, fy, gn, base, (n + kSub - 1) / kSub, lim, ai, best, bi);
                }
            }
            bi = group_nearest_(best, bi, np.N);
            tgt = tgt >= bi ? tgt : bi;
            const int ti = min(tgt, np.N - 1);
            T xn, yn, yaw_des, v_des = np.target_speed;
            if (one_pass) {
                xn = c.x[(bi / kSub) * kSubStride + bi % kSub];
                yn = c.y[(bi / kSub) * kSubStride + bi % kSub];
                yaw_des = s_yaw[ti];
                if (np.flags & SCCBF_NOMINAL_WAYPOINT_SPEED) v_des = traj4[ti].w;
            } else {
                const v2 wn = *reinterpret_cast<const v2*>(traj4 + bi);
                const v4 wt = traj4[ti];
                xn = wn.x; yn = wn.y; yaw_des = wt.z; v_des = wt.w;
            }
            stanley_pid_(np, st.z, st.w, sth, cth, fx, fy, xn, yn, yaw_des, v_des, np.pid_state != nullptr, ps,
                         acc, delta, &v_track);
        }
        // filter
        Ego<T> e;
        T ur0, ur1;
        T at0[RPL], at1[RPL], bt[RPL];
        T hm = INFINITY;
        int bad;
        T beta_ref = T(0);
        if (!EXT) {
            L.sv = st;
            L.urv = p.dynamics == SCCBF_DYN_KBM ? v2{v_track, delta} : v2{acc, delta};
            bad = lane_rows_<T, RPL, false>(p, ag, g, G, false, L, e, ur0, ur1, at0, at1, bt, &hm);
        } else {
            // as k_group_ext forms them
            e.x = st.x; e.y = st.y; e.th = st.z; e.v = st.w;
            ur0 = acc;
            ur1 = sadbm ? delta : st.w * tan_(delta) * p.inv_L;
            bad = !(finite_(ur0) && finite_(ur1) && finite_(e.x) && finite_(e.y) && finite_(e.th) && finite_(e.v) &&
                    finite_(sl.x) && finite_(sl.y));
            sincos_(e.th, &e.sth, &e.cth);
            e.derive(p);
            T cthb = e.cth, sthb = e.sth, sb = T(0);
            e.un0 = ur0;
            e.un1 = ur1;
            if (sadbm) {
                beta_ref = atan_(tan_(ur1) * p.k_in);
                e.un1 = (beta_ref - sl.y) * a.inv_dt;
                T cb;
                sincos_(e.th + sl.x, &sthb, &cthb);
                sincos_(sl.x, &sb, &cb);
            }
            T chk = T(0);
#pragma unroll
            for (int r = 0; r < RPL; ++r) {
                const int m = g + r * G;
                T o[8] = {L.oa[r].x, L.oa[r].y, L.oa[r].z, L.oa[r].w, L.ob[r].x, L.ob[r].y, L.ob[r].z, L.ob[r].w};
                const int ty = m < p.M ? L.ty[r] : SCCBF_OBS_EMPTY;
                T a0, a1, b, h;
                ext_slot_row_<(RPL <= 4)>(ty, o, e, p, sadbm, sadbm, sl.x, cthb, sthb, sb, a0, a1, b, h);
                if (ty >= 0 && ty <= SCCBF_OBS_LANE6) hm = min_(hm, h);
                chk += a0 * T(0) + a1 * T(0) + b * T(0);
                to_wspace_(a0, a1, b, e, p, at0[r], at1[r], bt[r]);
            }
            bad |= !(chk == T(0));
        }
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
#pragma unroll
        for (int m = 1; m < G; m *= 2) hm = min_(hm, __shfl_xor(hm, m, 64));
        h_min = min_(h_min, hm);
        n_act += (s.st == SCCBF_STATUS_OPTIMAL) & ((s.w0 != T(0)) | (s.w1 != T(0)));
        n_inf += s.st == SCCBF_STATUS_INFEASIBLE;
        // the filter's answer, the vehicle step, then the obstacles move
        v2 u;
        T beta = T(0);
        if (!EXT) {
            u = result_u_(p, false, s.st, s.w0, s.w1, e, ur0, ur1);
            if (p.dynamics == SCCBF_DYN_KBM)
                kin_step_<T>(st, SCCBF_STEP_KIN_VEL, u.x, u.y, p.inv_L, a.max_steer, a.dt);
            else
                beta = vehicle_step_<T>(st, u.x, u.y, a.k_in, a.inv_lr, a.max_steer, a.dt);
        } else {
            T u0 = e.un0, u1 = e.un1;
            if (s.st == SCCBF_STATUS_OPTIMAL) {
                const T y1 = s.w1 * p.inv_l11;
                const T y0 = (s.w0 - p.l10 * y1) * p.inv_l00;
                u0 += y0;
                u1 += y1;
            }
            u = v2{u0, u1};
            if (sadbm) {
                if (s.st == SCCBF_STATUS_NONFINITE) {
                    u = v2{ur0, ur1};  // inputs pass through, the slip state stays
                } else {
                    sl.x += u1 * a.dt;                       // reference cbf/cbf.py:421
                    sl.y = beta_ref;
                    u.y = atan_(tan_(sl.x) * p.k_out);       // cbf.py:430
                }
                beta = vehicle_step_<T>(st, u.x, u.y, a.k_in, a.inv_lr, a.max_steer, a.dt);
            } else {
                kin_step_<T>(st, SCCBF_STEP_UNICYCLE, u.x, u.y, p.inv_L, a.max_steer, a.dt);
            }
        }
        if (a.trace && live && g == 0)
            reinterpret_cast<v4*>(a.trace)[(long long)tick * p.A + ag] = v4{u.x, u.y, beta, (T)s.st};
        if (a.motion == SCCBF_MOTION_CONSTANT_VELOCITY) {
#pragma unroll
            for (int r = 0; r < RPL; ++r) advance_slot_<T>(L.ty[r], L.oa[r], L.ob[r], a.dt);
        } else if (a.motion == SCCBF_MOTION_SEEKER) {
#pragma unroll
            for (int r = 0; r < RPL; ++r)
                seek_slot_<T>(L.ty[r], L.oa[r], L.ob[r], st.x, st.y, a.seek_gain, a.seek_v_min, a.dt);
        }
    }
    if (live && g == 0) {
        reinterpret_cast<v4*>(a.state)[ag] = st;
        if (course) np.target_idx[ag] = tgt;
        if (course && np.pid_state) reinterpret_cast<v2*>(np.pid_state)[ag] = ps;
        if (sadbm) reinterpret_cast<v2*>(a.slip)[ag] = sl;
        if (a.h_min) a.h_min[ag] = h_min;
        if (a.n_active) a.n_active[ag] = n_act;
        if (a.n_infeasible) a.n_infeasible[ag] = n_inf;
    }
}

}  // namespace
