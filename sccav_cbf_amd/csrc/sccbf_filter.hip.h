// sccbf_filter.hip.h - the filter kernels: k_group, k_wave, k_partials, k_passthrough
// Part of the sccbf library (one translation unit, see sccbf.hip); device code for gfx950.
#pragma once
#include "sccbf_rows.hip.h"

namespace {

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
            slot_row_<(RPL <= 4)>(ty, o, e, p, a0, a1, b, h);
            if (p.rows_out && m < M && live)
                reinterpret_cast<v4*>(p.rows_out)[(long long)m * p.A + qp] = v4{a0, a1, b, h};
            if (h_min && ty >= 0 && ty <= SCCBF_OBS_LANE6) *h_min = min_(*h_min, h);  // real slots only
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
    // the row index rides in the low bits of the key: 8 bits for up to 256 rows per QP, else 10
    constexpr unsigned kIdxMask = G * RPL > 256 ? 0x3FFu : 0xFFu;
    static_assert(G * RPL <= 1024, "row index field");
    unsigned idx[RPL];  // mask - row index: ties in violation go to the lowest index
#pragma unroll
    for (int r = 0; r < RPL; ++r) idx[r] = kIdxMask - (unsigned)(r * G + g);
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
            key[r] = (int)((viol_bits(viol, thr) & ~kIdxMask) | idx[r]);
            kbest = (r == 0 || key[r] > kbest) ? key[r] : kbest;  // always the key of a real slot
        }
        const int kmax = group_imax<G>(kbest);
        const bool go = (s.st == SCCBF_STATUS_OPTIMAL) & viol_key_above(kmax, thr, kIdxMask);
        if (!__any(go)) return false;  // every group of the wave is done
        if (it >= cap) {               // no convergence: rows numerically inconsistent
            s.st = go ? SCCBF_STATUS_INFEASIBLE : s.st;
            return false;
        }
        T c0 = T(0), c1 = T(0), cb = T(0);
        if (LDS_PICK) {
            const int at = qslot + (int)(kIdxMask - ((unsigned)kmax & kIdxMask));
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

// Wavefronts per SIMD the register allocation of a k_group instantiation is held to (at least): the fp64 kernels with up
// to four rows per lane sit at 126 ... 130 registers, two either side of the 128 that four wavefronts allow, and the
// allocator finds the two when asked (nothing spilled: tests/test_isa.py); the others are left alone.
#ifndef SCCBF_EXP_NO_WAVES_ATTR
template <typename T, int RPL> constexpr int kGroupWaves = (sizeof(T) == 8 && RPL <= 4) ? 4 : 1;
#else
template <typename T, int RPL> constexpr int kGroupWaves = 1;
#endif

// The arguments every wave needs before it can issue its first loads come first and one
// by one: the leading 16 dwords of the kernel arguments are preloaded into SGPRs at wave
// launch (-amdgpu-kernarg-preload-count), which a by-value struct does not qualify for.
template <typename T, int G, int RPL, bool QP_ONLY, bool NT>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(kGroupWaves<T, RPL>))) void k_group(long long A, const T* obs, const T* state, const T* u_ref,
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

// ------------------------------------------ rows staged in LDS only: any M up to SCCBF_MAX_ROWS
// Table of one QP in LDS: row m at words 3 m, 3 m + 1, 3 m + 2 (at0, at1, bt), M rounded up to whole turns of
// the G lanes that scan it; the rows past M are (0, 0, -max): never violated.  Words between the tables of two
// QPs of a workgroup: odd, so that the groups of a wavefront scan different banks.
__host__ __device__ inline int wave_table_rows(int M, int G) { return (M + G - 1) / G * G; }
__host__ __device__ inline int wave_table_stride(int M, int G) { return (3 * wave_table_rows(M, G)) | 1; }

// The active-set iteration of a group of G lanes (16, 32 or the whole wavefront) on the rows it has staged
// in LDS: every lane scans rows g, g + G, ... (the same number for every lane, no lane masked: the offsets
// of an unrolled turn are immediates of the reads, and the loop counter is scalar), the most violated row
// comes from a max over the group, the pivot reads it by index (broadcast reads).  Keys as in active_set_:
// violation in the high bits, 1023 - row below, compared as signed integers; the threshold is applied once, to
// the winner.  The trip count is wave-uniform; the groups of a wavefront commit under `go`.
// FIRST_GIVEN: first_key is the winner of the scan at w = 0 (where a row's violation is its bt), which the threads that
// formed the rows have already taken (wave_first_key_), and the first turn of the loop needs no scan.
template <typename T, int G, bool FIRST_GIVEN = false>
__device__ inline Pivot<T> wave_active_set_(int M, int g, const T* s_rows, int bad, int first_key = 0)
{
    Pivot<T> s;
    s.reset(bad ? SCCBF_STATUS_NONFINITE : SCCBF_STATUS_OPTIMAL);
    const int cap = kCapPerRow * M + kCapBase;
    const int turns = (M + G - 1) / G;
    const T* mine = s_rows + 3 * g;
    for (int it = 0;; ++it) {  // wave-uniform loop
        const T thr = Lim<T>::viol_tol * (abs_(s.w0) + abs_(s.w1)) + Lim<T>::tiny_n2;
        int kmax = first_key;
        if (!FIRST_GIVEN || it > 0) {
            int kbest = (int)0x80000000u;
#pragma unroll 4
            for (int i = 0; i < turns; ++i) {
                const T* r = mine + 3 * G * i;
                const T viol = r[2] - (r[0] * s.w0 + r[1] * s.w1);
                const int key = (int)((viol_bits(viol, thr) & ~0x3FFu) | (unsigned)(1023 - g - G * i));
                kbest = key > kbest ? key : kbest;
            }
            kmax = group_imax<G>(kbest);
        }
        const bool go = (s.st == SCCBF_STATUS_OPTIMAL) & viol_key_above(kmax, thr, 0x3FFu);
        if (!__any(go)) break;  // every group of the wavefront is done
        if (it >= cap) {
            s.st = go ? SCCBF_STATUS_INFEASIBLE : s.st;
            break;
        }
        const T* w = s_rows + 3 * (go ? 1023 - (kmax & 0x3FF) : 0);
        pivot_onto_(s, w[0], w[1], w[2], go);  // broadcast reads
    }
    return s;
}

// The forming side of the same table: row m of the QP whose table starts at `table`.
template <typename T> __device__ inline void wave_table_put(T* table, int m, T t0, T t1, T tb)
{
    table[3 * m] = t0;
    table[3 * m + 1] = t1;
    table[3 * m + 2] = tb;
}
// The scan key of row m at w = 0, as wave_active_set_ forms it (violation = bt, threshold of a zero optimum).
template <typename T> __device__ inline int wave_first_key_(int m, T tb)
{
    return (int)((viol_bits(tb, (T)Lim<T>::tiny_n2) & ~0x3FFu) | (unsigned)(1023 - m));
}
// Thread `rsub` of the G that form a QP's rows fills its share of the padding (fewer than G rows).
template <typename T, int G> __device__ inline void wave_table_pad(T* table, int M, int rsub)
{
    const int m = M + rsub;
    if (m < wave_table_rows(M, G)) wave_table_put(table, m, T(0), T(0), -Lim<T>::big);
}

// The slots of a lane (g, g + G, ...) come kWaveBatch at a time, and the next batch is on its way while the
// rows of the current one are formed: with the few wavefronts per CU that the row tables leave room for, a
// wavefront has to overlap its own loads and arithmetic.  row(m, oa, ob, ty) forms and stores row m.
// Two slots per batch (measured against 1, 4 and 8 at 129 ... 1024 rows): the row code is inlined once per
// slot of the two batches, four copies, the kernels stay within 96 registers (five wavefronts per SIMD) and
// spill nothing; four per batch take 128 registers and are 8 % slower, eight (80 KB of code, more than the
// instruction cache holds) three times slower.
// fp64: 195 registers with two slots per batch (two wavefronts per SIMD), fewer with one: the kernels come in both
// depths (template parameter NB) and the host launches the shallow one where the LDS tables leave room for more than
// two wavefronts per SIMD (257 rows: 339 -> 283 us, 512: 507 -> 440), the deep one where they do not and a wavefront
// has to cover its own latency (1024 rows: 1072 against 1263 us).
#ifndef SCCBF_WAVE_BATCH
#define SCCBF_WAVE_BATCH 2
#endif
template <typename T> constexpr int kWaveBatch = SCCBF_WAVE_BATCH;
template <typename T> constexpr int kWaveBatchShallow = sizeof(T) == 8 ? 1 : kWaveBatch<T>;
template <typename T, int NB> struct SlotBatch {
    typename VT<T>::v4 oa[NB], ob[NB];
    int ty[NB];
};

template <typename T, int G, bool QP_ONLY, int NB>
__device__ inline void batch_load_(const KArgs<T>& p, const ObsAddr& oad, long long qp, bool type_pa, int m0,
                                   SlotBatch<T, NB>& B)
{
    using v4 = typename VT<T>::v4;
    const v4* obs4 = reinterpret_cast<const v4*>(p.obs);
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int m = min(m0 + k * G, p.M - 1);  // past the end: a slot that exists, its row is dropped
        if (QP_ONLY) {
            B.oa[k] = obs4[(long long)m * p.A + qp];
        } else {
            const long long at = oad.at(m);
            B.oa[k] = obs4[at];
            B.ob[k] = obs4[at + oad.half];
            B.ty[k] = type_pa ? p.obs_type[(long long)m * p.A + qp] : p.obs_type[m];
        }
    }
}

template <typename T, int G, bool QP_ONLY, int NB, typename RowFn>
__device__ inline void staged_rows_(const KArgs<T>& p, long long qp, int g, RowFn row)
{
    const bool type_pa = (p.flags & SCCBF_FLAG_TYPE_PER_AGENT) != 0;
    const ObsAddr oad = obs_addr_(p, qp);
    constexpr int kStep = NB * G;
    const int M = p.M;
    SlotBatch<T, NB> b0, b1;
    batch_load_<T, G, QP_ONLY, NB>(p, oad, qp, type_pa, g, b0);
    for (int base = 0; base < M; base += 2 * kStep) {  // wave-uniform
        if (base + kStep < M) batch_load_<T, G, QP_ONLY, NB>(p, oad, qp, type_pa, base + kStep + g, b1);
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int m = base + k * G + g;
            if (m < M) row(m, b0.oa[k], b0.ob[k], b0.ty[k]);
        }
        if (base + kStep >= M) break;
        if (base + 2 * kStep < M) batch_load_<T, G, QP_ONLY, NB>(p, oad, qp, type_pa, base + 2 * kStep + g, b0);
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int m = base + kStep + k * G + g;
            if (m < M) row(m, b1.oa[k], b1.ob[k], b1.ty[k]);
        }
    }
}

// G lanes per QP in the active-set iteration (G = 16, 32 or 64: a whole wavefront), QPB QPs per workgroup:
// at least the agents that share a 128-byte line of an obstacle plane (eight in fp32, four in fp64).
// While the rows are formed the threads are laid out the other way round, agents fastest: thread t forms
// rows t / QPB, t / QPB + G, ... of agent t % QPB, so a load instruction reads whole lines (QPB agents of a
// few consecutive slots), consecutive slots of one kind keep the kind branch uniform, and every row goes into
// its agent's table in LDS.  After the workgroup barrier thread t scans the table of agent t / G with the
// other G - 1 lanes of its group; the answer goes back through LDS to the thread that holds that agent's
// state (t < QPB), which stores it.  A CU should hold at least two workgroups: while one iterates, the other
// loads; one alone leaves the memory idle for a quarter of the time.
template <typename T> struct WaveTables {
    T* rows;      // QPB tables of wave_table_stride(M, G) values
    T* answer;    // QPB x (w0, w1)
    int* status;  // QPB
    int* bad;     // QPB
    int* first;   // QPB: the largest scan key of the QP's rows at w = 0 (ds_max by the threads that form them)
    __device__ WaveTables(unsigned char* raw, int M, int qpb, int G)
    {
        rows = reinterpret_cast<T*>(raw);
        answer = rows + (size_t)qpb * wave_table_stride(M, G);
        status = reinterpret_cast<int*>(answer + 2 * qpb);
        bad = status + qpb;
        first = bad + qpb;
    }
};
template <typename T> __host__ __device__ inline size_t wave_lds_bytes(int M, int qpb, int G)
{
    return ((size_t)qpb * wave_table_stride(M, G) + 2 * qpb) * sizeof(T) + 3 * qpb * sizeof(int);
}

// First agent of a workgroup.  PAIRED: the workgroup takes half of the agents that share a line, and the
// workgroup with the other half is the one eight further on in the grid: workgroups are dealt to the XCDs in
// turn, so the two sit on the same XCD, start about together, and the second finds the line in that XCD's L2.
// (Placement is not promised: if it differs, the line is fetched twice and nothing else changes.)  The grid
// is a multiple of 16 workgroups.
template <int QPB, bool PAIRED> __device__ inline long long wave_first_agent(unsigned block = blockIdx.x)
{
    if (!PAIRED) return (long long)block * QPB;
    const unsigned xcd = block & 7u, slot = block >> 3;
    return ((long long)(slot >> 1) * 8 + xcd) * (2 * QPB) + (slot & 1u) * QPB;
}

// The work of one workgroup on the QPB agents from wave_first_agent(block) on; smem_raw: the workgroup's dynamic
// LDS.  A workgroup may call it again for another `block`: the tables it reads after its last barrier
// (answer, status) are rewritten only after the next call's second barrier.
template <typename T, int QPB, bool QP_ONLY, int G, bool PAIRED, int NB = kWaveBatch<T>>
__device__ inline void wave_body_(const KArgs<T>& p, unsigned block, unsigned char* smem_raw)
{
    using v4 = typename VT<T>::v4;
    static_assert(QPB * G % 64 == 0, "whole wavefronts");
    constexpr int RS = G;  // rows of an agent formed side by side
    const int M = p.M;
    const WaveTables<T> tab(smem_raw, M, QPB, G);
    const int stride = wave_table_stride(M, G);
    const int qa = (int)(threadIdx.x % QPB), rsub = (int)(threadIdx.x / QPB);  // forming rows
    const int qs = (int)(threadIdx.x / G), g = (int)(threadIdx.x % G);         // solving

    long long qp = wave_first_agent<QPB, PAIRED>(block) + qa;
    const bool live = qp < p.A;
    if (!live) qp = p.A - 1;  // the thread stays for the barriers; its stores are masked
    if (threadIdx.x < QPB) {
        tab.bad[threadIdx.x] = 0;
        tab.first[threadIdx.x] = (int)0x80000000u;
    }

    Ego<T> e;
    T ur0, ur1;
    int bad;
    load_ego_(p, qp, QP_ONLY, e, ur0, ur1, bad);
    __syncthreads();

    T* table = tab.rows + (size_t)qa * stride;
    T chk = T(0);  // stays 0 while every row is finite
    int kfirst = (int)0x80000000u;
    staged_rows_<T, RS, QP_ONLY, NB>(p, qp, rsub, [&](int m, const v4& oa, const v4& ob, int ty) {
        T a0, a1, b, h;
        if (QP_ONLY) {
            a0 = oa.x; a1 = oa.y; b = oa.z; h = T(0);
        } else {
            const T o[8] = {oa.x, oa.y, oa.z, oa.w, ob.x, ob.y, ob.z, ob.w};
            slot_row_(ty, o, e, p, a0, a1, b, h);
            if (p.rows_out && live) reinterpret_cast<v4*>(p.rows_out)[(long long)m * p.A + qp] = v4{a0, a1, b, h};
        }
        chk += a0 * T(0) + a1 * T(0) + b * T(0);
        T t0, t1, tb;
        to_wspace_(a0, a1, b, e, p, t0, t1, tb);
        wave_table_put(table, m, t0, t1, tb);
        const int key = wave_first_key_(m, tb);
        kfirst = key > kfirst ? key : kfirst;
    });
    wave_table_pad<T, G>(table, M, rsub);
    atomicMax(&tab.first[qa], kfirst);
    if (bad | !(chk == T(0))) tab.bad[qa] = 1;
    __syncthreads();
    const Pivot<T> s = wave_active_set_<T, G, true>(M, g, tab.rows + (size_t)qs * stride, tab.bad[qs], tab.first[qs]);
    if (g == 0) {
        tab.answer[2 * qs] = s.w0;
        tab.answer[2 * qs + 1] = s.w1;
        tab.status[qs] = s.st;
    }
    __syncthreads();
    if (rsub == 0 && live)
        store_result_(p, qp, QP_ONLY, tab.status[qa], tab.answer[2 * qa], tab.answer[2 * qa + 1], e, ur0, ur1);
}

// Wavefronts per SIMD the register allocation of this family is held to (at least), fp32: five (96 registers) while
// four-wavefront workgroups (16 or 32 lanes per QP) are what fills a CU; the workgroups of 64 lanes per QP come two
// or three to a CU by their LDS tables whatever the registers up to 128, so those kernels may use them.  fp64 is left alone.
#ifndef SCCBF_WAVE_WAVES_F32
#define SCCBF_WAVE_WAVES_F32 5
#endif
template <typename T, int G = 32> constexpr int kWaveWaves = sizeof(T) == 4 ? (G == 64 ? 4 : SCCBF_WAVE_WAVES_F32) : 1;

template <typename T, int QPB, bool QP_ONLY, int G, bool PAIRED = false, int NB = kWaveBatch<T>>
__global__ __launch_bounds__(QPB * G) __attribute__((amdgpu_waves_per_eu(kWaveWaves<T, G>))) void k_wave(const KArgs<T> p)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    wave_body_<T, QPB, QP_ONLY, G, PAIRED, NB>(p, blockIdx.x, smem_raw);
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
    Ego<T> e;
    e.x = s.x; e.y = s.y; e.th = s.z; e.v = s.w; e.cth = cth; e.sth = sth;
    if (!slot_partials_(ty, o, e, p, d)) {
        d.h = T(1);
        d.hx = d.hy = d.hth = d.hv = d.ht = T(0);
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

}  // namespace
