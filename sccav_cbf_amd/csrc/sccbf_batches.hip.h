// sccbf_batches.hip.h - K independent batches of the filter solved by ONE grid
// Part of the sccbf library (one translation unit, see sccbf.hip); device code for gfx950.
//
// A 65536-agent x 16-row launch is one round of four wavefronts per SIMD: all of them load, then all of
// them compute, and the launch ends before memory and arithmetic have anything to overlap with.  The same
// kernel over 2^20 agents runs at the rate the memory system sustains.  Callers whose agents arrive as
// separate batches (the scenarios of a Monte-Carlo sweep, the clients of a serving tick: every batch with
// its own arrays) get that launch shape here: grid.y = batch, grid.x = the workgroups of the largest batch,
// and every workgroup reads its batch's pointers and size from a table of sccbf_batch records in device
// memory (one 64-byte scalar load).  The arithmetic is that of k_group / k_wave, instruction for instruction.
#pragma once
#include "sccbf_filter.hip.h"

namespace {

static_assert(sizeof(sccbf_batch) == 64, "one batch record = one 64-byte scalar load");

// This workgroup's batch: its record -> the per-batch fields of the argument block.  False for a record that
// the kernels must not touch (no agents, more agents than the grid was sized for, or a pointer missing): its
// workgroups leave.  On entry p.A holds the size the grid was made for.
// A pointer read from the table is a number as far as the compiler knows; said to be a global address (which a device
// pointer of the caller's is), the loads and stores through it are global ones, counted by vmcnt alone, and the kernel
// keeps the counted waits of k_group (as generic "flat" accesses they would all be waited for together).
template <typename P> __device__ inline P* global_ptr_(const void* p)
{
    typedef __attribute__((address_space(1))) P* gptr;
    return (P*)(gptr)(unsigned long long)p;
}

template <typename T> __device__ inline bool batch_args_(const sccbf_batch* table, int M, KArgs<T>& p)
{
    const long long a_max = p.A;
    const sccbf_batch d = table[blockIdx.y];  // uniform over the workgroup: scalar loads
    p.A = d.A;
    p.state = global_ptr_<const T>(d.state);
    p.u_ref = global_ptr_<const T>(d.u_ref);
    p.obs = global_ptr_<const T>(d.obs);
    p.obs_type = global_ptr_<const uint8_t>(d.obs_type);
    p.u_out = global_ptr_<T>(d.u_out);
    p.status = global_ptr_<int>(d.status);
    p.rows_out = global_ptr_<T>(d.rows_out);
    const bool slots = M == 0 || (d.obs && d.obs_type);
    return d.A > 0 && d.A <= a_max && d.state && d.u_ref && d.u_out && d.status && slots;
}

// k_group over a table of batches: straight-line code like k_group itself (a loop over several turns per
// workgroup, for batches larger than the grid, costs 35 registers per lane and three of seven wavefronts per SIMD).
template <typename T, int G, int RPL, bool NT>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(kGroupWaves<T, RPL>))) void k_group_batches(const sccbf_batch* table, int M, unsigned flags,
                                                          const KArgs<T> rest)
{
    KArgs<T> p = rest;
    p.M = M; p.flags = flags;
    if (!batch_args_(table, M, p)) return;
    const int g = (int)(threadIdx.x % G);
    const long long qp = (long long)blockIdx.x * (kBlock / G) + threadIdx.x / G;
    if (qp >= p.A) return;  // groups are lane-aligned: a whole group leaves together
    LaneLoads<T, RPL> L;
    group_load_<T, RPL, false, NT>(p, qp, g, G, L);
    group_solve_<T, G, RPL, false>(p, qp, g, L);
}

template <typename T, int QPB, int G, bool PAIRED, int NB = kWaveBatch<T>>
__global__ __launch_bounds__(QPB * G) __attribute__((amdgpu_waves_per_eu(kWaveWaves<T, G>))) void k_wave_batches(const sccbf_batch* table, int M, unsigned flags,
                                                          const KArgs<T> rest)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    KArgs<T> p = rest;
    p.M = M; p.flags = flags;
    if (!batch_args_(table, M, p)) return;
    // the grid is sized like k_wave's for the largest batch (a multiple of 16 workgroups when PAIRED): the
    // workgroups past this batch's last agent leave together, whole pairs of them
    if (wave_first_agent<QPB, PAIRED>(blockIdx.x & ~(PAIRED ? 15u : 0u)) >= p.A) return;
    wave_body_<T, QPB, false, G, PAIRED, NB>(p, blockIdx.x, smem_raw);
}

// no rows at all: the nominal input is optimal
template <typename T>
__global__ __launch_bounds__(kBlock) void k_passthrough_batches(const sccbf_batch* table, const KArgs<T> rest)
{
    using v4 = typename VT<T>::v4;
    using v2 = typename VT<T>::v2;
    KArgs<T> p = rest;
    if (!batch_args_(table, 0, p)) return;
    const long long a = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (a >= p.A) return;
    const v2 ur = reinterpret_cast<const v2*>(p.u_ref)[a];
    const v4 s = reinterpret_cast<const v4*>(p.state)[a];
    const bool ok = finite_(ur.x) && finite_(ur.y) && finite_(s.x) && finite_(s.y) && finite_(s.z) && finite_(s.w);
    reinterpret_cast<v2*>(p.u_out)[a] = ur;
    p.status[a] = ok ? SCCBF_STATUS_OPTIMAL : SCCBF_STATUS_NONFINITE;
}

}  // namespace
