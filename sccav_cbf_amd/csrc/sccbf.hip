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
//   k_wave<T, WPB, G>   G = 16, 32 or 64 lanes per QP (64 = one wavefront per QP), rows staged
//                       in LDS only, any M; the most violated row is found by a max over the group.
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
//
// Files: sccbf_common.hip.h (constants, math, cross-lane helpers), sccbf_qp.hip.h (pivot),
// sccbf_rows.hip.h (barriers and rows), sccbf_filter.hip.h (filter kernels), sccbf_loop.hip.h
// (nominal controller, vehicle step, rollout), sccbf_ext.hip.h (unicycle-input and steer-rate
// models), sccbf_ingest.hip.h (bounding boxes and lane points -> slots); this file: host-side
// dispatch and the C ABI.
#include <atomic>
#include <tuple>
#include <utility>

#include "sccbf_loop.hip.h"
#include "sccbf_ext.hip.h"
#include "sccbf_ingest.hip.h"
#include "sccbf_batches.hip.h"
#include "sccbf_minviol.hip.h"

// The library is built from this one source in three parts that compile side by side (_build.py):
// the fp32 entry points, the fp64 ones, and sccbf_rollout_f64 on its own, because that part is compiled
// without the machine-level hoisting of loop invariants: with it, the 50 double-precision constants of the
// library tan / atan / atan2 / sincos are parked in registers across the tick loop, 358 registers per lane
// and one wavefront per SIMD; without it 256 and two, with nothing spilled.  The fourth part holds the
// entry points that solve several batches in one launch, in both precisions.  No macro: everything.
#if defined(SCCBF_ONLY_F32)
#define SCCBF_HAS_F32 1
#define SCCBF_HAS_F64 0
#define SCCBF_HAS_ROLLOUT_F64 0
#define SCCBF_HAS_BATCHES 0
#elif defined(SCCBF_ONLY_F64)
#define SCCBF_HAS_F32 0
#define SCCBF_HAS_F64 1
#define SCCBF_HAS_ROLLOUT_F64 0
#define SCCBF_HAS_BATCHES 0
#elif defined(SCCBF_ONLY_ROLLOUT_F64)
#define SCCBF_HAS_F32 0
#define SCCBF_HAS_F64 0
#define SCCBF_HAS_ROLLOUT_F64 1
#define SCCBF_HAS_BATCHES 0
#elif defined(SCCBF_ONLY_BATCHES)
#define SCCBF_HAS_F32 0
#define SCCBF_HAS_F64 0
#define SCCBF_HAS_ROLLOUT_F64 0
#define SCCBF_HAS_BATCHES 1
#else
#define SCCBF_HAS_F32 1
#define SCCBF_HAS_F64 1
#define SCCBF_HAS_ROLLOUT_F64 1
#define SCCBF_HAS_BATCHES 1
#endif

namespace {

// ------------------------------------------------------------------- host side
// Every kernel is launched through hipLaunchKernel, whose return value is this launch's own verdict.  The runtime's
// "last error" is never read: it may hold an error that another library left unread on this thread, which is that
// library's to report, not this one's to consume.
thread_local int t_launch_rc = SCCBF_OK;

template <typename... P, typename... Args, size_t... I>
void launch_impl_(void (*kernel)(P...), dim3 grid, dim3 block, size_t lds, hipStream_t s, std::index_sequence<I...>,
                  Args&&... args)
{
    std::tuple<std::remove_cv_t<P>...> vals(static_cast<std::remove_cv_t<P>>(args)...);
    void* ptrs[sizeof...(P)] = {static_cast<void*>(&std::get<I>(vals))...};
    t_launch_rc = hipLaunchKernel(reinterpret_cast<const void*>(kernel), grid, block, ptrs, lds, s) == hipSuccess
                      ? SCCBF_OK : SCCBF_ERR_LAUNCH;
}

template <typename... P, typename... Args>
void launch_(void (*kernel)(P...), dim3 grid, dim3 block, size_t lds, hipStream_t s, Args&&... args)
{
    static_assert(sizeof...(P) == sizeof...(Args), "one argument per kernel parameter");
    launch_impl_(kernel, grid, block, lds, s, std::index_sequence_for<P...>(), std::forward<Args>(args)...);
}

// SCCBF_OK or SCCBF_ERR_LAUNCH of the calling thread's latest launch_()
inline int launched_() { return t_launch_rc; }

// obs is laid out one way: at most one of the layout flags
constexpr uint32_t kObsLayoutFlags = SCCBF_FLAG_OBS_SHARED | SCCBF_FLAG_OBS_TILED | SCCBF_FLAG_OBS_AGENT_MAJOR;
inline bool one_obs_layout_(uint32_t flags)
{
    const uint32_t l = flags & kObsLayoutFlags;
    return (l & (l - 1)) == 0;
}

template <typename T>
int fill_args(const sccbf_params* sp, int64_t A, int32_t M, KArgs<T>& k)
{
    if (!sp || A < 0 || M < 0) return SCCBF_ERR_ARG;
    if (A > 0x7fffffffLL) return SCCBF_ERR_ARG;  // the kernels keep strides of up to 2 A in 32 bits
    if (M > SCCBF_MAX_ROWS) return SCCBF_ERR_ROWS;
    if (sp->dynamics != SCCBF_DYN_DBM && sp->dynamics != SCCBF_DYN_KBM && sp->dynamics != SCCBF_DYN_DUM &&
        sp->dynamics != SCCBF_DYN_SADBM)
        return SCCBF_ERR_ARG;
    if (!one_obs_layout_(sp->flags)) return SCCBF_ERR_ARG;
    const double r00 = sp->R[0], r01 = 0.5 * (sp->R[1] + sp->R[2]), r11 = sp->R[3];
    if (!(r00 > 0.0) || !std::isfinite(r00) || !std::isfinite(r01) || !std::isfinite(r11))
        return SCCBF_ERR_WEIGHT;
    const double l00 = std::sqrt(r00), l10 = r01 / l00, t = r11 - l10 * l10;
    if (!(t > 0.0)) return SCCBF_ERR_WEIGHT;
    const double l11 = std::sqrt(t);
    if ((sp->dynamics == SCCBF_DYN_DBM || sp->dynamics == SCCBF_DYN_SADBM) &&
        !(sp->lr > 0.0 && sp->lf + sp->lr > 0.0))
        return SCCBF_ERR_ARG;
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
    k.step_state = nullptr; k.step_beta = nullptr; k.step_max_steer = T(0); k.step_dt = T(0);
    return SCCBF_OK;
}

// lanes-per-QP when the caller leaves the choice to the library (measured on MI355X: with 16 rows
// 4 lanes from 32768 QPs up to 1M, 8 around 8192-16384, 16 below; with 32 ... 128 rows and 65536 QPs
// 8 lanes; from there on the kernels that keep the rows in LDS only are faster than 16 lanes x 16
// registers: 98 against 116 us at 200 rows, 116 against 130 at 256)
// From per-agent lists 16 lanes read runs of 16 consecutive slots of an agent: 65 ... 128 rows, 65536 QPs:
// 44 against 57 us at 96 rows, 53 against 63 at 128 (from tiles 8 lanes: 50 and 59).
// fp64: at most 8 rows per lane (16 rows of doubles need 365 registers: 8 lanes x 16 rows took 629 us at 128 rows where
// 16 lanes x 8 rows take 124).
int auto_lanes(int64_t A, int M, uint32_t flags, int scalar_bytes)
{
    if (M > 128) return 64;  // rows in LDS only; wave_lanes() then picks the group
    if ((flags & SCCBF_FLAG_OBS_AGENT_MAJOR) && M > 64) return 16;
    int g;
    if (A >= (1 << 15)) g = M <= 16 ? 4 : 8;
    else if (A >= (1 << 13)) g = 8;
    else g = 16;
    const int rows_per_lane = scalar_bytes == 8 ? 8 : 16;
    while (g < 16 && M > rows_per_lane * g) g *= 2;
    return g;
}

// Obstacle bytes per launch from which the slots are loaded non-temporally when the caller has
// not said that they are read once (SCCBF_FLAG_OBS_STREAMED): a batch that is read again on the
// next tick is better served from the caches up to about the size of the Infinity Cache; batches
// read once gain from 34 MB on (2-6 % at 65536 agents x 16 slots), 5-10 % from 134 MB on.
constexpr double kNtBytes = 48.0 * 1024 * 1024;

template <typename T, bool QP_ONLY, int G, int RPL>
int launch_group(const KArgs<T>& k, hipStream_t s)
{
    const long long per_block = kBlock / G;
    const long long blocks = (k.A + per_block - 1) / per_block;
    if (blocks > 0x7fffffffLL) return SCCBF_ERR_ARG;
    // rows of the QP-only entry are a third of the bytes of obstacle slots: always cached loads
    const bool nt = !QP_ONLY && ((k.flags & SCCBF_FLAG_OBS_STREAMED) || (double)k.A * k.M * 8 * sizeof(T) >= kNtBytes);
    if (nt)
        launch_((k_group<T, G, RPL, QP_ONLY, !QP_ONLY>), dim3((unsigned)blocks), dim3(kBlock), 0, s, k.A,
                           k.obs, k.state, k.u_ref, k.obs_type, k.M, k.flags, k);
    else
        launch_((k_group<T, G, RPL, QP_ONLY, false>), dim3((unsigned)blocks), dim3(kBlock), 0, s, k.A,
                           k.obs, k.state, k.u_ref, k.obs_type, k.M, k.flags, k);
    return launched_();
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

// The kernels that keep the rows in LDS only: G lanes per QP, QPB QPs per workgroup = the agents that share a
// 128-byte line of an obstacle plane or a few of those, never a part of one (a line fetched by two
// workgroups is fetched into the L2 of two XCDs).  G = 16: 16 QPs (256 threads), G = 32: 8 QPs (256 threads),
// G = 64: 8 QPs in fp32 (512 threads), 4 in fp64 (256 threads).  A workgroup's row tables may take the whole
// 160 KB of a CU's LDS.
constexpr size_t kLdsPerCu = 160 * 1024;
template <typename T, int G> struct WaveShape {
    static constexpr int kQpb = G == 16 ? 16 : (G == 32 ? 8 : (sizeof(T) == 4 ? 8 : 4));
};

template <typename T> size_t wave_lds(int M, int qpb, int G) { return wave_lds_bytes<T>(M, qpb, G); }

// lanes per QP when the library chooses (measured at 65536 agents, fp32, 129 ... 1024 rows, both layouts): 16 lanes
// while five workgroups fit a CU (the kernels run five wavefronts per SIMD; up to 160 fp32 rows: 69 against 74 us at
// 160 rows, 95 against 89 at 200), 32 while four do (up to 416: 147 against 164 us at 350 rows), else a wavefront per
// QP, whose workgroups have twice the wavefronts for the same tables
// fit16: workgroups of 16-lane groups a CU must hold for 16 lanes to be chosen: five for k_wave, four for k_wave_ext, which runs
// four wavefronts per SIMD (unicycle model: 88 against 94 us at 200 rows)
// fp64 (two or three wavefronts per SIMD by the registers): 32 lanes while two workgroups fit (up to 416 rows: 194 against
// 253 us at 200 rows, 380 against 399 at 400 for 64 lanes), else 64.
template <typename T> int wave_lanes(int M, int fit16 = 5)
{
    if (sizeof(T) == 8) return wave_lds<T>(M, WaveShape<T, 32>::kQpb, 32) * 2 <= kLdsPerCu ? 32 : 64;
    if (wave_lds<T>(M, WaveShape<T, 16>::kQpb, 16) * fit16 <= kLdsPerCu) return 16;
    if (wave_lds<T>(M, WaveShape<T, 32>::kQpb, 32) * 4 <= kLdsPerCu) return 32;
    return 64;
}

// more than 64 KB of dynamic LDS has to be granted per kernel, once per device
template <auto Kernel> bool grant_lds(size_t lds)
{
    if (lds <= 64 * 1024) return true;
    if (lds > kLdsPerCu) return false;
    // the only state the library keeps: which devices have the attribute (setting it twice from two
    // threads at once is harmless, the flag only saves the call)
    static std::atomic<bool> done[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (!done[dev].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)kLdsPerCu) != hipSuccess)
            return false;
        done[dev].store(true, std::memory_order_release);
    }
    return true;
}

// grid of a kernel whose workgroups take QPB agents each; PAIRED: see wave_first_agent()
template <int QPB, bool PAIRED> long long wave_blocks(long long A)
{
    if (!PAIRED) return (A + QPB - 1) / QPB;
    const long long groups = (A + 2 * QPB - 1) / (2 * QPB);  // of 2 QPB agents: a line's worth
    return (groups + 7) / 8 * 16;
}

// two workgroups of a line's worth of tables (eight fp32 agents, four fp64) no longer fit a CU: half of
// them per workgroup, paired on an XCD
template <typename T> bool wave_paired(int M) { return wave_lds<T>(M, WaveShape<T, 64>::kQpb, 64) * 2 > kLdsPerCu; }

// fp64: the kernels that stage one slot per load batch instead of two use fewer registers (three wavefronts per SIMD
// instead of two): taken where a workgroup's LDS tables leave room for more than eight wavefronts per CU, else the
// deeper ones, whose wavefronts cover more of their own latency.  fp32 has one depth.
template <typename T> bool wave_shallow(size_t lds, int threads)
{
    return sizeof(T) == 8 && (kLdsPerCu / lds) * (size_t)(threads / 64) > 8;
}

template <typename T, bool QP_ONLY, int G, int QPB, bool PAIRED>
int launch_wave_shape(const KArgs<T>& k, hipStream_t s)
{
    const size_t lds = wave_lds<T>(k.M, QPB, G);
    const long long blocks = wave_blocks<QPB, PAIRED>(k.A);
    if (blocks > 0x7fffffffLL) return SCCBF_ERR_ARG;
    if (wave_shallow<T>(lds, QPB * G)) {
        constexpr int kNb = kWaveBatchShallow<T>;
        if (!grant_lds<k_wave<T, QPB, QP_ONLY, G, PAIRED, kNb>>(lds)) return SCCBF_ERR_VARIANT;
        launch_((k_wave<T, QPB, QP_ONLY, G, PAIRED, kNb>), dim3((unsigned)blocks), dim3(QPB * G), lds, s, k);
        return launched_();
    }
    if (!grant_lds<k_wave<T, QPB, QP_ONLY, G, PAIRED>>(lds)) return SCCBF_ERR_VARIANT;
    launch_((k_wave<T, QPB, QP_ONLY, G, PAIRED>), dim3((unsigned)blocks), dim3(QPB * G), lds, s, k);
    return launched_();
}

template <typename T, bool QP_ONLY, int G>
int launch_wave(const KArgs<T>& k, hipStream_t s)
{
    constexpr int kQpb = WaveShape<T, G>::kQpb;
    if (G == 64 && wave_paired<T>(k.M)) return launch_wave_shape<T, QP_ONLY, G, (G == 64 ? kQpb / 2 : kQpb), G == 64>(k, s);
    return launch_wave_shape<T, QP_ONLY, G, kQpb, false>(k, s);
}

template <typename T, bool QP_ONLY>
int dispatch(const sccbf_params* sp, KArgs<T>& k, void* stream)
{
    if (k.A == 0) return SCCBF_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (k.M == 0) {
        const long long blocks = (k.A + kBlock - 1) / kBlock;
        if (blocks > 0x7fffffffLL) return SCCBF_ERR_ARG;
        launch_((k_passthrough<T, QP_ONLY>), dim3((unsigned)blocks), dim3(kBlock), 0, s, k);
        return launched_();
    }
    int lanes = sp->lanes_per_qp;
    if (lanes == 0) lanes = auto_lanes(k.A, k.M, k.flags, (int)sizeof(T));
    if (lanes == 64 && sp->lanes_per_qp == 0) lanes = wave_lanes<T>(k.M);
    switch (lanes) {
    case 1: return launch_group_rpl<T, QP_ONLY, 1>(k, s);
    case 2: return launch_group_rpl<T, QP_ONLY, 2>(k, s);
    case 4: return launch_group_rpl<T, QP_ONLY, 4>(k, s);
    case 8: return launch_group_rpl<T, QP_ONLY, 8>(k, s);
    case 16: return k.M <= 128 ? launch_group_rpl<T, QP_ONLY, 16>(k, s) : launch_wave<T, QP_ONLY, 16>(k, s);
    case 32: return launch_wave<T, QP_ONLY, 32>(k, s);
    case 64: return launch_wave<T, QP_ONLY, 64>(k, s);
    default: return SCCBF_ERR_VARIANT;
    }
}

// unicycle-input and steer-rate models: 4, 8 or 16 lanes per QP with the rows in registers
template <typename T, int G>
int launch_ext_rpl(const KArgs<T>& k, const XArgs<T>& x, hipStream_t s)
{
    const long long per_block = kBlock / G;
    const long long blocks = (k.A + per_block - 1) / per_block;
    if (blocks > 0x7fffffffLL) return SCCBF_ERR_ARG;
    const int rpl = (k.M + G - 1) / G;
    const dim3 grid((unsigned)blocks), block(kBlock);
    if (rpl <= 1) launch_((k_group_ext<T, G, 1>), grid, block, 0, s, k, x);
    else if (rpl <= 2) launch_((k_group_ext<T, G, 2>), grid, block, 0, s, k, x);
    else if (rpl <= 4) launch_((k_group_ext<T, G, 4>), grid, block, 0, s, k, x);
    else if (rpl <= 8) launch_((k_group_ext<T, G, 8>), grid, block, 0, s, k, x);
    else if (rpl <= 16) launch_((k_group_ext<T, G, 16>), grid, block, 0, s, k, x);
    else return SCCBF_ERR_ROWS;
    return launched_();
}

template <typename T, int G, int QPB, bool PAIRED>
int launch_wave_ext_shape(const KArgs<T>& k, const XArgs<T>& x, hipStream_t s)
{
    const size_t lds = wave_lds<T>(k.M, QPB, G);
    const long long blocks = wave_blocks<QPB, PAIRED>(k.A);
    if (blocks > 0x7fffffffLL) return SCCBF_ERR_ARG;
    if (wave_shallow<T>(lds, QPB * G)) {
        constexpr int kNb = kWaveBatchShallow<T>;
        if (!grant_lds<k_wave_ext<T, QPB, G, PAIRED, kNb>>(lds)) return SCCBF_ERR_VARIANT;
        launch_((k_wave_ext<T, QPB, G, PAIRED, kNb>), dim3((unsigned)blocks), dim3(QPB * G), lds, s, k, x);
        return launched_();
    }
    if (!grant_lds<k_wave_ext<T, QPB, G, PAIRED>>(lds)) return SCCBF_ERR_VARIANT;
    launch_((k_wave_ext<T, QPB, G, PAIRED>), dim3((unsigned)blocks), dim3(QPB * G), lds, s, k, x);
    return launched_();
}

template <typename T, int G>
int launch_wave_ext(const KArgs<T>& k, const XArgs<T>& x, hipStream_t s)
{
    constexpr int kQpb = WaveShape<T, G>::kQpb;
    if (G == 64 && wave_paired<T>(k.M)) return launch_wave_ext_shape<T, G, (G == 64 ? kQpb / 2 : kQpb), G == 64>(k, x, s);
    return launch_wave_ext_shape<T, G, kQpb, false>(k, x, s);
}

template <typename T>
int dispatch_ext(const sccbf_params* sp, const KArgs<T>& k, const XArgs<T>& x, void* stream)
{
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int lanes = sp->lanes_per_qp;
    if (lanes == 0) {  // as for the bicycle models (auto_lanes)
        if (k.M > 128) lanes = wave_lanes<T>(k.M, 4);
        else if (((k.flags & SCCBF_FLAG_OBS_AGENT_MAJOR) || sizeof(T) == 8) && k.M > 64) lanes = 16;   // fp64: at most 8 rows per lane
        else if (k.A >= (1 << 13)) lanes = k.M <= 16 ? 4 : 8;
        else lanes = 16;
    }
    if (lanes == 4 && k.M <= 64) return launch_ext_rpl<T, 4>(k, x, s);
    if (lanes == 8 && k.M <= 128) return launch_ext_rpl<T, 8>(k, x, s);
    if (lanes == 16) return k.M <= 128 ? launch_ext_rpl<T, 16>(k, x, s) : launch_wave_ext<T, 16>(k, x, s);
    if (lanes == 32) return launch_wave_ext<T, 32>(k, x, s);
    if (lanes == 64) return launch_wave_ext<T, 64>(k, x, s);
    return SCCBF_ERR_VARIANT;
}

// SCCBF_FLAG_INFEASIBLE_MIN_VIOLATION: second pass over the agents the filter has just reported infeasible
// transform: 0 the model's output map (DBM / KBM steering), 1 none (DUM, slip coordinates), 2 none and no state (QP stage)
template <typename T>
int launch_min_violation(const KArgs<T>& k, const T* rows, int transform, void* stream)
{
    const long long blocks = (k.A + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffLL) return SCCBF_ERR_ARG;
    launch_((k_min_violation<T>), dim3((unsigned)blocks), dim3(kBlock), 0, reinterpret_cast<hipStream_t>(stream), k, rows,
            transform);
    return launched_();
}

// sccbf_prewarm: the attribute that lets a workgroup take more than 64 KB of LDS, for every kernel of this part of the
// library that may ask for it (the kernels that keep the rows in LDS only), on the current device
template <typename T, int G> bool prewarm_wave_()
{
    constexpr int kQpb = WaveShape<T, G>::kQpb;
    constexpr int kHalf = G == 64 ? kQpb / 2 : kQpb;
    constexpr int kNb = kWaveBatchShallow<T>;   // fp64: the second depth of every kernel (fp32: the same instantiation again)
    bool ok = grant_lds<k_wave<T, kQpb, false, G, false>>(kLdsPerCu) & grant_lds<k_wave<T, kQpb, true, G, false>>(kLdsPerCu) &
              grant_lds<k_wave_ext<T, kQpb, G, false>>(kLdsPerCu) &
              grant_lds<k_wave<T, kQpb, false, G, false, kNb>>(kLdsPerCu) & grant_lds<k_wave<T, kQpb, true, G, false, kNb>>(kLdsPerCu) &
              grant_lds<k_wave_ext<T, kQpb, G, false, kNb>>(kLdsPerCu);
    if (G == 64)
        ok &= grant_lds<k_wave<T, kHalf, false, G, G == 64>>(kLdsPerCu) & grant_lds<k_wave<T, kHalf, true, G, G == 64>>(kLdsPerCu) &
              grant_lds<k_wave_ext<T, kHalf, G, G == 64>>(kLdsPerCu) &
              grant_lds<k_wave<T, kHalf, false, G, G == 64, kNb>>(kLdsPerCu) &
              grant_lds<k_wave<T, kHalf, true, G, G == 64, kNb>>(kLdsPerCu) & grant_lds<k_wave_ext<T, kHalf, G, G == 64, kNb>>(kLdsPerCu);
    return ok;
}
template <typename T> int prewarm_part_()
{
    const bool ok = prewarm_wave_<T, 16>() & prewarm_wave_<T, 32>() & prewarm_wave_<T, 64>();
    return ok ? SCCBF_OK : SCCBF_ERR_NO_DEVICE;
}

template <typename T>
int solve_impl(const sccbf_params* sp, int64_t A, int32_t M, const T* state, const T* u_ref,
               const T* obs, const uint8_t* obs_type, T* u_out, int32_t* status, T* rows_out,
               void* stream)
{
    KArgs<T> k;
    int rc = fill_args<T>(sp, A, M, k);
    if (rc != SCCBF_OK) return rc;
    if (sp->dynamics == SCCBF_DYN_SADBM) return SCCBF_ERR_ARG;  // needs the slip state: sccbf_solve_sa_*
    if (A > 0 && (!state || !u_ref || !u_out || !status)) return SCCBF_ERR_ARG;
    if (A > 0 && M > 0 && (!obs || !obs_type)) return SCCBF_ERR_ARG;
    k.state = state; k.u_ref = u_ref; k.obs = obs; k.obs_type = obs_type;
    k.u_out = u_out; k.status = status; k.rows_out = rows_out;
    const bool min_viol = (sp->flags & SCCBF_FLAG_INFEASIBLE_MIN_VIOLATION) && A > 0 && M > 0;
    if (min_viol && !rows_out) return SCCBF_ERR_ARG;  // the second pass works on the rows the filter writes
    if (sp->dynamics == SCCBF_DYN_DUM && A > 0 && M > 0) {
        if (sp->flags & SCCBF_FLAG_CONE_BETA_STATE) return SCCBF_ERR_ARG;
        XArgs<T> x;
        x.slip = nullptr; x.dt = T(0); x.inv_dt = T(0);
        rc = dispatch_ext<T>(sp, k, x, stream);
        return (rc == SCCBF_OK && min_viol) ? launch_min_violation<T>(k, rows_out, 1, stream) : rc;
    }
    rc = dispatch<T, false>(sp, k, stream);
    if (rc != SCCBF_OK || !min_viol) return rc;
    return launch_min_violation<T>(k, rows_out, (sp->flags & SCCBF_FLAG_SLIP_IO) ? 1 : 0, stream);
}

template <typename T>
int vehicle_step_impl(int64_t A, T* state, const T* u, double lr, double lf, double max_steer, double dt,
                      T* beta_out, void* stream);

// sccbf_solve_* followed by sccbf_vehicle_step_* on the filtered input, in the same launch
template <typename T>
int solve_step_impl(const sccbf_params* sp, double max_steer, double dt, int64_t A, int32_t M, T* state,
                    const T* u_ref, const T* obs, const uint8_t* obs_type, T* u_out, int32_t* status,
                    T* beta_out, void* stream)
{
    KArgs<T> k;
    int rc = fill_args<T>(sp, A, M, k);
    if (rc != SCCBF_OK) return rc;
    if (sp->dynamics != SCCBF_DYN_DBM || (sp->flags & SCCBF_FLAG_SLIP_IO)) return SCCBF_ERR_ARG;  // the step takes (a, delta)
    if (sp->flags & SCCBF_FLAG_INFEASIBLE_MIN_VIOLATION) return SCCBF_ERR_ARG;  // a second pass: sccbf_solve_* only
    if (!(max_steer >= 0.0) || !std::isfinite(dt)) return SCCBF_ERR_ARG;
    if (A == 0) return SCCBF_OK;
    if (!state || !u_ref || !u_out || !status) return SCCBF_ERR_ARG;
    if (M > 0 && (!obs || !obs_type)) return SCCBF_ERR_ARG;
    k.state = state; k.u_ref = u_ref; k.obs = obs; k.obs_type = obs_type;
    k.u_out = u_out; k.status = status; k.rows_out = nullptr;
    if (M == 0) {  // nothing to filter: the pass-through kernel, then the step
        rc = dispatch<T, false>(sp, k, stream);
        if (rc != SCCBF_OK) return rc;
        return vehicle_step_impl<T>(A, state, u_out, sp->lr, sp->lf, max_steer, dt, beta_out, stream);
    }
    k.step_state = state; k.step_beta = beta_out; k.step_max_steer = (T)max_steer; k.step_dt = (T)dt;
    return dispatch<T, false>(sp, k, stream);
}

template <typename T>
int solve_sa_impl(const sccbf_params* sp, double dt, int64_t A, int32_t M, const T* state, T* slip,
                  const T* u_ref, const T* obs, const uint8_t* obs_type, T* u_out, int32_t* status,
                  T* rows_out, void* stream)
{
    if (!sp) return SCCBF_ERR_ARG;
    sccbf_params q = *sp;
    q.dynamics = SCCBF_DYN_SADBM;
    KArgs<T> k;
    int rc = fill_args<T>(&q, A, M, k);
    if (rc != SCCBF_OK) return rc;
    if (M < 1 || !(dt > 0.0) || !std::isfinite(dt) || (q.flags & SCCBF_FLAG_SLIP_IO)) return SCCBF_ERR_ARG;
    if (q.flags & SCCBF_FLAG_INFEASIBLE_MIN_VIOLATION) return SCCBF_ERR_ARG;  // the slip state has been advanced already
    if (A == 0) return SCCBF_OK;
    if (!state || !slip || !u_ref || !u_out || !status || !obs || !obs_type) return SCCBF_ERR_ARG;
    k.state = state; k.u_ref = u_ref; k.obs = obs; k.obs_type = obs_type;
    k.u_out = u_out; k.status = status; k.rows_out = rows_out;
    XArgs<T> x;
    x.slip = slip; x.dt = (T)dt; x.inv_dt = (T)(1.0 / dt);
    return dispatch_ext<T>(&q, k, x, stream);
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
    launch_((k_partials<T>), dim3((unsigned)blocks), dim3(kBlock), 0,
                       reinterpret_cast<hipStream_t>(stream), k);
    return launched_();
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
    const bool min_viol = (q.flags & SCCBF_FLAG_INFEASIBLE_MIN_VIOLATION) && A > 0 && M > 0;
    q.flags = 0;
    int rc = fill_args<T>(&q, A, M, k);
    if (rc != SCCBF_OK) return rc;
    if (A > 0 && (!u_nom || !u_out || !status)) return SCCBF_ERR_ARG;
    if (A > 0 && M > 0 && !rows) return SCCBF_ERR_ARG;
    k.state = nullptr; k.u_ref = u_nom; k.obs = rows; k.obs_type = nullptr;
    k.u_out = u_out; k.status = status; k.rows_out = nullptr;
    rc = dispatch<T, true>(&q, k, stream);
    return (rc == SCCBF_OK && min_viol) ? launch_min_violation<T>(k, rows, 2, stream) : rc;
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
    launch_((k_nominal<T>), dim3((unsigned)blocks), dim3(kNomBlock), 0,
                       reinterpret_cast<hipStream_t>(stream), k);
    return launched_();
}

template <typename T>
int vehicle_step_impl(int64_t A, T* state, const T* u, double lr, double lf, double max_steer, double dt,
                      T* beta_out, void* stream)
{
    if (A < 0 || A > 0x7fffffffLL || !(lr > 0.0) || !(lf + lr > 0.0) || !(max_steer >= 0.0) || !std::isfinite(dt))
        return SCCBF_ERR_ARG;
    if (A == 0) return SCCBF_OK;
    if (!state || !u) return SCCBF_ERR_ARG;
    const long long blocks = (A + kBlock - 1) / kBlock;
    launch_((k_vehicle_step<T>), dim3((unsigned)blocks), dim3(kBlock), 0,
                       reinterpret_cast<hipStream_t>(stream), (long long)A, state, u,
                       (T)(lr / (lf + lr)), (T)(1.0 / lr), (T)max_steer, (T)dt, beta_out);
    return launched_();
}

template <typename T, int RPL, int G>
int launch_rollout(const RArgs<T>& r, bool ext, hipStream_t s)
{
    const long long per_block = kBlock / G;
    const long long blocks = (r.f.A + per_block - 1) / per_block;
    if (blocks > 0x7fffffffLL) return SCCBF_ERR_ARG;
    if (ext) launch_((k_rollout<T, RPL, true, G>), dim3((unsigned)blocks), dim3(kBlock), 0, s, r);
    else launch_((k_rollout<T, RPL, false, G>), dim3((unsigned)blocks), dim3(kBlock), 0, s, r);
    return launched_();
}

template <typename T>
int rollout_impl(const sccbf_params* sp, const sccbf_nominal_params* np, const sccbf_rollout_params* rp,
                 int64_t A, int32_t M, int32_t N, T* state, const T* traj, int32_t* target_idx, T* pid_state,
                 T* slip, const T* obs, const uint8_t* obs_type, T* h_min, int32_t* n_active,
                 int32_t* n_infeasible, T* trace, void* stream)
{
    if (!np || !rp) return SCCBF_ERR_ARG;
    RArgs<T> r;
    int rc = fill_args<T>(sp, A, M, r.f);
    if (rc != SCCBF_OK) return rc;
    if (N < 0 || M < 1 || rp->ticks < 0 || !(rp->dt > 0.0) || !(rp->max_steer >= 0.0)) return SCCBF_ERR_ARG;
    if (rp->motion != SCCBF_MOTION_STATIC && rp->motion != SCCBF_MOTION_CONSTANT_VELOCITY &&
        rp->motion != SCCBF_MOTION_SEEKER)
        return SCCBF_ERR_ARG;
    if (rp->motion == SCCBF_MOTION_SEEKER && !(rp->seek_gain >= 0.0 && rp->seek_v_min >= 0.0)) return SCCBF_ERR_ARG;
    if (sp->flags & (SCCBF_FLAG_SLIP_IO | SCCBF_FLAG_CONE_BETA_STATE | SCCBF_FLAG_INFEASIBLE_MIN_VIOLATION))
        return SCCBF_ERR_ARG;  // works in the caller's coordinates; no second pass inside the tick loop
    if (sp->flags & SCCBF_FLAG_OBS_SHARED) return SCCBF_ERR_ARG;  // moving obstacles are per agent
    const bool sadbm = sp->dynamics == SCCBF_DYN_SADBM;
    if ((sp->dynamics == SCCBF_DYN_KBM || sp->dynamics == SCCBF_DYN_DUM) && !(sp->L > 0.0)) return SCCBF_ERR_ARG;
    if (A == 0 || rp->ticks == 0) return SCCBF_OK;
    if (!state || !obs || !obs_type) return SCCBF_ERR_ARG;
    if (N > 0 && (!traj || !target_idx)) return SCCBF_ERR_ARG;
    if (sadbm && !slip) return SCCBF_ERR_ARG;
    if (N == 0) pid_state = nullptr;
    if (pid_state && !(np->dt > 0.0)) return SCCBF_ERR_ARG;
    fill_nominal_args<T>(np, A, N, r.n);
    r.f.state = state; r.f.u_ref = nullptr; r.f.obs = obs; r.f.obs_type = obs_type;
    r.f.u_out = nullptr; r.f.status = nullptr; r.f.rows_out = nullptr;
    r.n.state = state; r.n.traj = traj; r.n.target_idx = target_idx; r.n.pid_state = pid_state;
    r.n.u_ref = nullptr;
    r.state = state;
    r.slip = sadbm ? slip : nullptr;
    r.k_in = (T)(sp->lr / (sp->lf + sp->lr)); r.inv_lr = (T)(1.0 / sp->lr);
    r.max_steer = (T)rp->max_steer; r.dt = (T)rp->dt; r.inv_dt = (T)(1.0 / rp->dt);
    r.seek_gain = (T)rp->seek_gain; r.seek_v_min = (T)rp->seek_v_min;
    r.ticks = rp->ticks; r.motion = rp->motion;
    r.h_min = h_min; r.n_active = n_active; r.n_infeasible = n_infeasible; r.trace = trace;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool ext = sadbm || sp->dynamics == SCCBF_DYN_DUM;
    // four lanes per agent hold up to 64 slots, a whole wavefront per agent up to 1024
    if (M <= 16 * kNomLanes) {
        const int rpl = (M + kNomLanes - 1) / kNomLanes;
        if (rpl <= 1) return launch_rollout<T, 1, kNomLanes>(r, ext, s);
        if (rpl <= 2) return launch_rollout<T, 2, kNomLanes>(r, ext, s);
        if (rpl <= 4) return launch_rollout<T, 4, kNomLanes>(r, ext, s);
        if (rpl <= 8) return launch_rollout<T, 8, kNomLanes>(r, ext, s);
        return launch_rollout<T, 16, kNomLanes>(r, ext, s);
    }
    const int rpl = (M + 63) / 64;
    if (rpl <= 2) return launch_rollout<T, 2, 64>(r, ext, s);
    if (rpl <= 4) return launch_rollout<T, 4, 64>(r, ext, s);
    if (rpl <= 8) return launch_rollout<T, 8, 64>(r, ext, s);
    return launch_rollout<T, 16, 64>(r, ext, s);
}

template <typename T>
int vehicle_step_kin_impl(int64_t A, T* state, const T* u, int32_t mode, double L, double max_steer, double dt,
                          void* stream)
{
    if (A < 0 || A > 0x7fffffffLL || !(max_steer >= 0.0) || !std::isfinite(dt)) return SCCBF_ERR_ARG;
    if (mode != SCCBF_STEP_KIN_ACC && mode != SCCBF_STEP_KIN_VEL && mode != SCCBF_STEP_UNICYCLE) return SCCBF_ERR_ARG;
    if (mode != SCCBF_STEP_UNICYCLE && !(L > 0.0)) return SCCBF_ERR_ARG;
    if (A == 0) return SCCBF_OK;
    if (!state || !u) return SCCBF_ERR_ARG;
    const long long blocks = (A + kBlock - 1) / kBlock;
    launch_((k_vehicle_step_kin<T>), dim3((unsigned)blocks), dim3(kBlock), 0,
                       reinterpret_cast<hipStream_t>(stream), (long long)A, state, u, (int)mode,
                       (T)(L > 0.0 ? 1.0 / L : 0.0), (T)max_steer, (T)dt);
    return launched_();
}

template <typename T>
int ingest_impl(int64_t A, int32_t M, const T* bbox, const uint8_t* kind, double buffer, uint32_t flags,
                T* obs, void* stream)
{
    if (A < 0 || M < 0 || A > 0x7fffffffLL || M > SCCBF_MAX_ROWS || !std::isfinite(buffer)) return SCCBF_ERR_ARG;
    if (flags & ~(SCCBF_FLAG_TYPE_PER_AGENT | SCCBF_FLAG_OBS_TILED | SCCBF_FLAG_OBS_AGENT_MAJOR)) return SCCBF_ERR_ARG;
    if (!one_obs_layout_(flags)) return SCCBF_ERR_ARG;
    if (A == 0 || M == 0) return SCCBF_OK;
    if (!bbox || !kind || !obs) return SCCBF_ERR_ARG;
    IArgs<T> k;
    k.A = A; k.M = M; k.flags = flags; k.bbox = bbox; k.kind = kind; k.buffer = (T)buffer; k.obs = obs;
    const long long blocks = (A * M + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffLL) return SCCBF_ERR_ARG;
    launch_((k_ingest<T>), dim3((unsigned)blocks), dim3(kBlock), 0,
                       reinterpret_cast<hipStream_t>(stream), k);
    return launched_();
}

template <typename T>
int prepare_impl(int64_t A, int32_t M, T* obs, uint8_t* obs_type, uint32_t flags, void* stream)
{
    if (A < 0 || M < 0 || A > 0x7fffffffLL || M > SCCBF_MAX_ROWS) return SCCBF_ERR_ARG;
    if (flags & ~(SCCBF_FLAG_TYPE_PER_AGENT | kObsLayoutFlags)) return SCCBF_ERR_ARG;
    if (!one_obs_layout_(flags) || ((flags & SCCBF_FLAG_OBS_SHARED) && (flags & SCCBF_FLAG_TYPE_PER_AGENT)))
        return SCCBF_ERR_ARG;
    if (A == 0 || M == 0) return SCCBF_OK;
    if (!obs || !obs_type) return SCCBF_ERR_ARG;
    PArgs<T> k;
    k.A = (flags & SCCBF_FLAG_OBS_SHARED) ? 1 : A;   // one table: one "agent"
    k.M = M; k.flags = flags; k.obs = obs; k.type = obs_type;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const long long blocks = (k.A * M + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffLL) return SCCBF_ERR_ARG;
    launch_((k_prepare_obs<T>), dim3((unsigned)blocks), dim3(kBlock), 0, st, k);
    if (launched_() != SCCBF_OK) return SCCBF_ERR_LAUNCH;  // the kinds are rewritten only over converted slots
    const long long n = (flags & SCCBF_FLAG_TYPE_PER_AGENT) ? (long long)M * A : M;
    launch_(k_prepare_types, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, n, obs_type);
    return launched_();
}

template <typename T>
int lane_fit_impl(int64_t A, int32_t K, int32_t degree, const T* px, const T* py, const T* sigma, double buffer,
                  int32_t m, int32_t M, uint32_t flags, T* obs, void* stream)
{
    if (A < 0 || A > 0x7fffffffLL || K < 1 || degree < 0 || degree > 3 || !std::isfinite(buffer)) return SCCBF_ERR_ARG;
    if (M < 1 || M > SCCBF_MAX_ROWS || m < 0 || m >= M) return SCCBF_ERR_ARG;
    if ((flags & ~(SCCBF_FLAG_OBS_TILED | SCCBF_FLAG_OBS_AGENT_MAJOR)) || !one_obs_layout_(flags)) return SCCBF_ERR_ARG;
    if (A == 0) return SCCBF_OK;
    if (!px || !py || !obs) return SCCBF_ERR_ARG;
    FArgs<T> k;
    k.A = A; k.K = K; k.degree = degree; k.m = m; k.M = M; k.flags = flags;
    k.px = px; k.py = py; k.sigma = sigma; k.buffer = (T)buffer; k.obs = obs;
    // Lanes per agent and points held per lane, by the length of the lists (65536 agents, cubic, us per
    // launch): up to 8 points one lane holds them all; up to 32 / 64 four lanes hold 8 / 16 each; beyond
    // that the points are read again on every pass, by four lanes in fp32 and by one in fp64.
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    auto launch = [&](auto lanes, auto heldc) {
        constexpr int G = decltype(lanes)::value, H = decltype(heldc)::value;
        const long long per_block = kBlock / G;
        const dim3 grid((unsigned)((A + per_block - 1) / per_block)), block(kBlock);
        switch (degree) {
        case 0: launch_((k_lane_fit<T, 1, G, H>), grid, block, 0, st, k); break;
        case 1: launch_((k_lane_fit<T, 2, G, H>), grid, block, 0, st, k); break;
        case 2: launch_((k_lane_fit<T, 3, G, H>), grid, block, 0, st, k); break;
        default: launch_((k_lane_fit<T, 4, G, H>), grid, block, 0, st, k); break;
        }
    };
    using std::integral_constant;
    if (K <= 8) launch(integral_constant<int, 1>{}, integral_constant<int, 8>{});
    else if (K <= 32) launch(integral_constant<int, 4>{}, integral_constant<int, 8>{});
    else if (K <= 64) launch(integral_constant<int, 4>{}, integral_constant<int, 16>{});
    else if (sizeof(T) == 4) launch(integral_constant<int, 4>{}, integral_constant<int, 0>{});
    else launch(integral_constant<int, 1>{}, integral_constant<int, 0>{});
    return launched_();
}


// ------------------------------------------------------------- several batches, one launch
template <typename T, int G, int RPL>
int launch_group_batches(const KArgs<T>& k, const sccbf_batch* table, int K, long long a_max, bool nt, hipStream_t s)
{
    const long long per_block = kBlock / G;
    const long long bx = (a_max + per_block - 1) / per_block;
    if (bx > 0x7fffffffLL) return SCCBF_ERR_ARG;
    const dim3 grid((unsigned)bx, (unsigned)K);
    if (nt) launch_((k_group_batches<T, G, RPL, true>), grid, dim3(kBlock), 0, s, table, k.M, k.flags, k);
    else launch_((k_group_batches<T, G, RPL, false>), grid, dim3(kBlock), 0, s, table, k.M, k.flags, k);
    return launched_();
}

template <typename T, int G>
int launch_group_batches_rpl(const KArgs<T>& k, const sccbf_batch* table, int K, long long a_max, bool nt, hipStream_t s)
{
    const int rpl = (k.M + G - 1) / G;
    if (rpl <= 1) return launch_group_batches<T, G, 1>(k, table, K, a_max, nt, s);
    if (rpl <= 2) return launch_group_batches<T, G, 2>(k, table, K, a_max, nt, s);
    if (rpl <= 4) return launch_group_batches<T, G, 4>(k, table, K, a_max, nt, s);
    if (rpl <= 8) return launch_group_batches<T, G, 8>(k, table, K, a_max, nt, s);
    if (rpl <= 16) return launch_group_batches<T, G, 16>(k, table, K, a_max, nt, s);
    return SCCBF_ERR_VARIANT;
}

template <typename T, int G, int QPB, bool PAIRED>
int launch_wave_batches_shape(const KArgs<T>& k, const sccbf_batch* table, int K, long long a_max, hipStream_t s)
{
    const size_t lds = wave_lds<T>(k.M, QPB, G);
    const long long bx = wave_blocks<QPB, PAIRED>(a_max);
    if (bx > 0x7fffffffLL) return SCCBF_ERR_ARG;
    if (wave_shallow<T>(lds, QPB * G)) {
        constexpr int kNb = kWaveBatchShallow<T>;
        if (!grant_lds<k_wave_batches<T, QPB, G, PAIRED, kNb>>(lds)) return SCCBF_ERR_VARIANT;
        launch_((k_wave_batches<T, QPB, G, PAIRED, kNb>), dim3((unsigned)bx, (unsigned)K), dim3(QPB * G), lds, s, table,
                           k.M, k.flags, k);
        return launched_();
    }
    if (!grant_lds<k_wave_batches<T, QPB, G, PAIRED>>(lds)) return SCCBF_ERR_VARIANT;
    launch_((k_wave_batches<T, QPB, G, PAIRED>), dim3((unsigned)bx, (unsigned)K), dim3(QPB * G), lds, s, table,
                       k.M, k.flags, k);
    return launched_();
}

template <typename T, int G>
int launch_wave_batches(const KArgs<T>& k, const sccbf_batch* table, int K, long long a_max, hipStream_t s)
{
    constexpr int kQpb = WaveShape<T, G>::kQpb;
    if (G == 64 && wave_paired<T>(k.M))
        return launch_wave_batches_shape<T, G, (G == 64 ? kQpb / 2 : kQpb), G == 64>(k, table, K, a_max, s);
    return launch_wave_batches_shape<T, G, kQpb, false>(k, table, K, a_max, s);
}

// The launch of K batches whose records already lie in device memory at `table`.
template <typename T>
int batches_launch(const sccbf_params* sp, int32_t K, int32_t M, int64_t a_max, const sccbf_batch* table, void* stream)
{
    if (K < 0 || K > SCCBF_MAX_BATCHES || a_max < 0) return SCCBF_ERR_ARG;
    KArgs<T> k;
    int rc = fill_args<T>(sp, a_max, M, k);
    if (rc != SCCBF_OK) return rc;
    // the models of sccbf_solve_* whose kernels take nothing but these arrays; no shared-nothing flags beyond its own
    if (sp->dynamics != SCCBF_DYN_DBM && sp->dynamics != SCCBF_DYN_KBM) return SCCBF_ERR_ARG;
    if (sp->flags & (SCCBF_FLAG_CONE_BETA_STATE | SCCBF_FLAG_INFEASIBLE_MIN_VIOLATION)) return SCCBF_ERR_ARG;
    if (K == 0 || a_max == 0) return SCCBF_OK;
    if (!table) return SCCBF_ERR_ARG;
    k.state = nullptr; k.u_ref = nullptr; k.obs = nullptr; k.obs_type = nullptr;
    k.u_out = nullptr; k.status = nullptr; k.rows_out = nullptr;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (M == 0) {
        const long long bx = (a_max + kBlock - 1) / kBlock;
        if (bx > 0x7fffffffLL) return SCCBF_ERR_ARG;
        launch_((k_passthrough_batches<T>), dim3((unsigned)bx, (unsigned)K), dim3(kBlock), 0, s, table, k);
        return launched_();
    }
    // lanes per QP as for one launch over all the agents (the batches fill the GPU together)
    const double total = (double)a_max * K;
    const int64_t a_all = total > 2147483647.0 ? 0x7fffffffLL : (int64_t)total;
    int lanes = sp->lanes_per_qp;
    if (lanes == 0) lanes = auto_lanes(a_all, M, sp->flags, (int)sizeof(T));
    if (lanes == 64 && sp->lanes_per_qp == 0) lanes = wave_lanes<T>(M);
    const bool nt = (sp->flags & SCCBF_FLAG_OBS_STREAMED) || total * M * 8 * sizeof(T) >= kNtBytes;
    switch (lanes) {
    case 1: return launch_group_batches_rpl<T, 1>(k, table, K, a_max, nt, s);
    case 2: return launch_group_batches_rpl<T, 2>(k, table, K, a_max, nt, s);
    case 4: return launch_group_batches_rpl<T, 4>(k, table, K, a_max, nt, s);
    case 8: return launch_group_batches_rpl<T, 8>(k, table, K, a_max, nt, s);
    case 16: return M <= 128 ? launch_group_batches_rpl<T, 16>(k, table, K, a_max, nt, s)
                             : launch_wave_batches<T, 16>(k, table, K, a_max, s);
    case 32: return launch_wave_batches<T, 32>(k, table, K, a_max, s);
    case 64: return launch_wave_batches<T, 64>(k, table, K, a_max, s);
    default: return SCCBF_ERR_VARIANT;
    }
}

template <typename T>
int solve_batches_impl(const sccbf_params* sp, int32_t K, int32_t M, int64_t a_max, const sccbf_batch* table,
                       void* stream)
{
    return batches_launch<T>(sp, K, M, a_max, table, stream);
}

#if SCCBF_HAS_BATCHES
// ------------------------------------------- the batch table kept by the library (sccbf_solve_batches_static)
// Every GPU the library runs on holds its own copy of this array, at its own address, until the process ends.
// In host code the name is only a handle: the address comes from hipGetSymbolAddress, once per device.
static __device__ __align__(256) sccbf_batch scratch[SCCBF_MAX_STATIC_BATCHES];

sccbf_batch* static_table_()
{
    static std::atomic<sccbf_batch*> table[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    sccbf_batch* t = table[dev].load(std::memory_order_acquire);
    if (!t) {
        void* q = nullptr;
        if (hipGetSymbolAddress(&q, HIP_SYMBOL(scratch)) != hipSuccess || !q) return nullptr;
        t = static_cast<sccbf_batch*>(q);
        table[dev].store(t, std::memory_order_release);
    }
    return t;
}

// Records travel to the table as kernel arguments (2 KB per launch): no copy engine, no staging buffer, nothing
// that waits for the stream; the fill is ordered on the stream like the launch that reads the table.
constexpr int kFillBatches = 32;
struct BatchChunk {
    sccbf_batch b[kFillBatches];
};
__global__ __launch_bounds__(kFillBatches * 8) void k_table_fill(sccbf_batch* dst, const BatchChunk c, int n)
{
    const int i = (int)threadIdx.x;  // one 8-byte field each
    if (i >= n * 8) return;
    const sccbf_batch& r = c.b[i / 8];
    const unsigned long long f[8] = {(unsigned long long)r.A,
                                     (unsigned long long)r.state, (unsigned long long)r.u_ref,
                                     (unsigned long long)r.obs, (unsigned long long)r.obs_type,
                                     (unsigned long long)r.u_out, (unsigned long long)r.status,
                                     (unsigned long long)r.rows_out};
    reinterpret_cast<unsigned long long*>(dst)[i] = f[i % 8];
}

template <typename T, int G> bool prewarm_wave_batches_()
{
    constexpr int kQpb = WaveShape<T, G>::kQpb;
    constexpr int kNb = kWaveBatchShallow<T>;
    bool ok = grant_lds<k_wave_batches<T, kQpb, G, false>>(kLdsPerCu) & grant_lds<k_wave_batches<T, kQpb, G, false, kNb>>(kLdsPerCu);
    if (G == 64)
        ok &= grant_lds<k_wave_batches<T, (G == 64 ? kQpb / 2 : kQpb), G, G == 64>>(kLdsPerCu) &
              grant_lds<k_wave_batches<T, (G == 64 ? kQpb / 2 : kQpb), G, G == 64, kNb>>(kLdsPerCu);
    return ok;
}

template <typename T>
int solve_batches_static_impl(const sccbf_params* sp, int32_t K, int32_t M, const sccbf_batch* host, void* stream)
{
    if (K < 0) return SCCBF_ERR_ARG;
    if (K > SCCBF_MAX_STATIC_BATCHES) return SCCBF_ERR_BATCHES;
    if (K > 0 && !host) return SCCBF_ERR_ARG;
    int64_t a_max = 0;
    for (int i = 0; i < K; ++i) {  // the records are host memory here: everything the kernel would skip is refused
        const sccbf_batch& r = host[i];
        if (r.A < 0 || r.A > 0x7fffffffLL) return SCCBF_ERR_ARG;
        if (r.A > 0 && (!r.state || !r.u_ref || !r.u_out || !r.status)) return SCCBF_ERR_ARG;
        if (r.A > 0 && M > 0 && (!r.obs || !r.obs_type)) return SCCBF_ERR_ARG;
        a_max = r.A > a_max ? r.A : a_max;
    }
    // arguments first: a call that is going to be refused must not touch the table
    int rc = batches_launch<T>(sp, 0, M, a_max, nullptr, stream);
    if (rc != SCCBF_OK || K == 0 || a_max == 0) return rc;
    sccbf_batch* table = static_table_();
    if (!table) return SCCBF_ERR_NO_DEVICE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    for (int at = 0; at < K; at += kFillBatches) {
        BatchChunk c;
        const int n = K - at < kFillBatches ? K - at : kFillBatches;
        for (int i = 0; i < kFillBatches; ++i) c.b[i] = host[at + (i < n ? i : 0)];
        launch_(k_table_fill, dim3(1), dim3(kFillBatches * 8), 0, s, table + at, c, n);
        if (launched_() != SCCBF_OK) return SCCBF_ERR_LAUNCH;
    }
    return batches_launch<T>(sp, K, M, a_max, table, stream);
}
#endif

}  // namespace

extern "C" {

#if SCCBF_HAS_BATCHES
int sccbf_solve_batches_f32(const sccbf_params* p, int32_t K, int32_t M, int64_t A_max, const sccbf_batch* batches,
                            void* stream)
{
    return solve_batches_impl<float>(p, K, M, A_max, batches, stream);
}

int sccbf_solve_batches_f64(const sccbf_params* p, int32_t K, int32_t M, int64_t A_max, const sccbf_batch* batches,
                            void* stream)
{
    return solve_batches_impl<double>(p, K, M, A_max, batches, stream);
}

int sccbf_solve_batches_static(const sccbf_params* p, int32_t scalar_bytes, int32_t K, int32_t M,
                               const sccbf_batch* batches, void* stream)
{
    if (scalar_bytes == 4) return solve_batches_static_impl<float>(p, K, M, batches, stream);
    if (scalar_bytes == 8) return solve_batches_static_impl<double>(p, K, M, batches, stream);
    return SCCBF_ERR_ARG;
}
#endif

#if SCCBF_HAS_F32
int sccbf_ingest_bbox_f32(int64_t A, int32_t M, const float* bbox, const uint8_t* kind, double buffer,
                          uint32_t flags, float* obs, void* stream)
{
    return ingest_impl<float>(A, M, bbox, kind, buffer, flags, obs, stream);
}
#endif

#if SCCBF_HAS_F64
int sccbf_ingest_bbox_f64(int64_t A, int32_t M, const double* bbox, const uint8_t* kind, double buffer,
                          uint32_t flags, double* obs, void* stream)
{
    return ingest_impl<double>(A, M, bbox, kind, buffer, flags, obs, stream);
}
#endif

#if SCCBF_HAS_F32
int sccbf_prepare_obs_f32(int64_t A, int32_t M, float* obs, uint8_t* obs_type, uint32_t flags, void* stream)
{
    return prepare_impl<float>(A, M, obs, obs_type, flags, stream);
}
#endif

#if SCCBF_HAS_F64
int sccbf_prepare_obs_f64(int64_t A, int32_t M, double* obs, uint8_t* obs_type, uint32_t flags, void* stream)
{
    return prepare_impl<double>(A, M, obs, obs_type, flags, stream);
}
#endif

#if SCCBF_HAS_F32
int sccbf_lane_fit_f32(int64_t A, int32_t K, int32_t degree, const float* px, const float* py,
                       const float* sigma, double buffer, int32_t m, int32_t M, uint32_t flags, float* obs,
                       void* stream)
{
    return lane_fit_impl<float>(A, K, degree, px, py, sigma, buffer, m, M, flags, obs, stream);
}
#endif

#if SCCBF_HAS_F64
int sccbf_lane_fit_f64(int64_t A, int32_t K, int32_t degree, const double* px, const double* py,
                       const double* sigma, double buffer, int32_t m, int32_t M, uint32_t flags, double* obs,
                       void* stream)
{
    return lane_fit_impl<double>(A, K, degree, px, py, sigma, buffer, m, M, flags, obs, stream);
}
#endif

#if SCCBF_HAS_F32
int sccbf_solve_step_f32(const sccbf_params* p, double max_steer, double dt, int64_t A, int32_t M, float* state,
                         const float* u_ref, const float* obs, const uint8_t* obs_type, float* u_out,
                         int32_t* status, float* beta_out, void* stream)
{
    return solve_step_impl<float>(p, max_steer, dt, A, M, state, u_ref, obs, obs_type, u_out, status, beta_out, stream);
}
#endif

#if SCCBF_HAS_F64
int sccbf_solve_step_f64(const sccbf_params* p, double max_steer, double dt, int64_t A, int32_t M, double* state,
                         const double* u_ref, const double* obs, const uint8_t* obs_type, double* u_out,
                         int32_t* status, double* beta_out, void* stream)
{
    return solve_step_impl<double>(p, max_steer, dt, A, M, state, u_ref, obs, obs_type, u_out, status, beta_out, stream);
}
#endif

#if SCCBF_HAS_F32
int sccbf_version(void) { return SCCBF_VERSION; }
#endif

// one piece of sccbf_prewarm per part of the library (the parts are separate objects of this one source)
extern "C" __attribute__((visibility("hidden"))) int sccbf_prewarm_f32_(void);
extern "C" __attribute__((visibility("hidden"))) int sccbf_prewarm_f64_(void);
extern "C" __attribute__((visibility("hidden"))) int sccbf_prewarm_batches_(void);
#if SCCBF_HAS_F32
int sccbf_prewarm_f32_(void) { return prewarm_part_<float>(); }
int sccbf_prewarm(void)
{
    const int a = sccbf_prewarm_f32_(), b = sccbf_prewarm_f64_(), c = sccbf_prewarm_batches_();
    return a != SCCBF_OK ? a : (b != SCCBF_OK ? b : c);
}
#endif
#if SCCBF_HAS_F64
int sccbf_prewarm_f64_(void) { return prewarm_part_<double>(); }
#endif
#if SCCBF_HAS_BATCHES
int sccbf_prewarm_batches_(void)
{
    const bool ok = prewarm_wave_batches_<float, 16>() & prewarm_wave_batches_<float, 32>() & prewarm_wave_batches_<float, 64>() &
                    prewarm_wave_batches_<double, 16>() & prewarm_wave_batches_<double, 32>() &
                    prewarm_wave_batches_<double, 64>();
    return ok && static_table_() ? SCCBF_OK : SCCBF_ERR_NO_DEVICE;
}
#endif

#if SCCBF_HAS_F32
const char* sccbf_strerror(int code)
{
    switch (code) {
    case SCCBF_OK: return "ok";
    case SCCBF_ERR_ARG: return "invalid argument (null pointer, negative size or bad enum)";
    case SCCBF_ERR_ROWS: return "too many rows or way-points for this entry point (see its limits in sccbf.h)";
    case SCCBF_ERR_WEIGHT: return "QP weight R is not symmetric positive definite";
    case SCCBF_ERR_LAUNCH: return "kernel launch failed";
    case SCCBF_ERR_VARIANT: return "lanes_per_qp cannot hold M rows or is not offered for this model (use 0)";
    case SCCBF_ERR_NO_DEVICE: return "no usable HIP device";
    case SCCBF_ERR_BATCHES: return "more batches than the library's own table holds (SCCBF_MAX_STATIC_BATCHES)";
    default: return "unknown error";
    }
}
#endif

#if SCCBF_HAS_F32
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
#endif

#if SCCBF_HAS_F32
int sccbf_solve_f32(const sccbf_params* p, int64_t A, int32_t M, const float* state,
                    const float* u_ref, const float* obs, const uint8_t* obs_type, float* u_out,
                    int32_t* status, float* rows_out, void* stream)
{
    return solve_impl<float>(p, A, M, state, u_ref, obs, obs_type, u_out, status, rows_out, stream);
}
#endif

#if SCCBF_HAS_F64
int sccbf_solve_f64(const sccbf_params* p, int64_t A, int32_t M, const double* state,
                    const double* u_ref, const double* obs, const uint8_t* obs_type, double* u_out,
                    int32_t* status, double* rows_out, void* stream)
{
    return solve_impl<double>(p, A, M, state, u_ref, obs, obs_type, u_out, status, rows_out, stream);
}
#endif

#if SCCBF_HAS_F32
int sccbf_solve_sa_f32(const sccbf_params* p, double dt, int64_t A, int32_t M, const float* state,
                       float* slip, const float* u_ref, const float* obs, const uint8_t* obs_type,
                       float* u_out, int32_t* status, float* rows_out, void* stream)
{
    return solve_sa_impl<float>(p, dt, A, M, state, slip, u_ref, obs, obs_type, u_out, status, rows_out, stream);
}
#endif

#if SCCBF_HAS_F64
int sccbf_solve_sa_f64(const sccbf_params* p, double dt, int64_t A, int32_t M, const double* state,
                       double* slip, const double* u_ref, const double* obs, const uint8_t* obs_type,
                       double* u_out, int32_t* status, double* rows_out, void* stream)
{
    return solve_sa_impl<double>(p, dt, A, M, state, slip, u_ref, obs, obs_type, u_out, status, rows_out, stream);
}
#endif

#if SCCBF_HAS_F32
int sccbf_qp_f32(const sccbf_params* p, int64_t A, int32_t M, const float* rows, const float* u_nom,
                 float* u_out, int32_t* status, void* stream)
{
    return qp_impl<float>(p, A, M, rows, u_nom, u_out, status, stream);
}
#endif

#if SCCBF_HAS_F64
int sccbf_qp_f64(const sccbf_params* p, int64_t A, int32_t M, const double* rows,
                 const double* u_nom, double* u_out, int32_t* status, void* stream)
{
    return qp_impl<double>(p, A, M, rows, u_nom, u_out, status, stream);
}
#endif

#if SCCBF_HAS_F32
int sccbf_partials_f32(const sccbf_params* p, int64_t A, int32_t M, const float* state,
                       const float* obs, const uint8_t* obs_type, float* out, void* stream)
{
    return partials_impl<float>(p, A, M, state, obs, obs_type, out, stream);
}
#endif

#if SCCBF_HAS_F64
int sccbf_partials_f64(const sccbf_params* p, int64_t A, int32_t M, const double* state,
                       const double* obs, const uint8_t* obs_type, double* out, void* stream)
{
    return partials_impl<double>(p, A, M, state, obs, obs_type, out, stream);
}
#endif

#if SCCBF_HAS_F32
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
#endif

#if SCCBF_HAS_F32
int sccbf_nominal_f32(const sccbf_nominal_params* p, int64_t A, int32_t N, const float* state,
                      const float* traj, int32_t* target_idx, float* pid_state, float* u_ref, void* stream)
{
    return nominal_impl<float>(p, A, N, state, traj, target_idx, pid_state, u_ref, stream);
}
#endif

#if SCCBF_HAS_F64
int sccbf_nominal_f64(const sccbf_nominal_params* p, int64_t A, int32_t N, const double* state,
                      const double* traj, int32_t* target_idx, double* pid_state, double* u_ref, void* stream)
{
    return nominal_impl<double>(p, A, N, state, traj, target_idx, pid_state, u_ref, stream);
}
#endif

#if SCCBF_HAS_F32
int sccbf_vehicle_step_f32(int64_t A, float* state, const float* u, double lr, double lf, double max_steer,
                           double dt, float* beta_out, void* stream)
{
    return vehicle_step_impl<float>(A, state, u, lr, lf, max_steer, dt, beta_out, stream);
}
#endif

#if SCCBF_HAS_F64
int sccbf_vehicle_step_f64(int64_t A, double* state, const double* u, double lr, double lf, double max_steer,
                           double dt, double* beta_out, void* stream)
{
    return vehicle_step_impl<double>(A, state, u, lr, lf, max_steer, dt, beta_out, stream);
}
#endif

#if SCCBF_HAS_F32
void sccbf_default_rollout_params(sccbf_rollout_params* p)
{
    if (!p) return;
    p->max_steer = 30.0 * 3.14159265358979323846 / 180.0;
    p->dt = 0.1;
    p->ticks = 0;
    p->motion = SCCBF_MOTION_STATIC;
    p->seek_gain = 0.2;
    p->seek_v_min = 3.0;
}
#endif

#if SCCBF_HAS_F32
int sccbf_rollout_f32(const sccbf_params* p, const sccbf_nominal_params* np, const sccbf_rollout_params* rp,
                      int64_t A, int32_t M, int32_t N, float* state, const float* traj, int32_t* target_idx,
                      float* pid_state, float* slip, const float* obs, const uint8_t* obs_type, float* h_min,
                      int32_t* n_active, int32_t* n_infeasible, float* trace, void* stream)
{
    return rollout_impl<float>(p, np, rp, A, M, N, state, traj, target_idx, pid_state, slip, obs, obs_type, h_min,
                               n_active, n_infeasible, trace, stream);
}
#endif

#if SCCBF_HAS_F32
int sccbf_vehicle_step_kin_f32(int64_t A, float* state, const float* u, int32_t mode, double L, double max_steer,
                               double dt, void* stream)
{
    return vehicle_step_kin_impl<float>(A, state, u, mode, L, max_steer, dt, stream);
}
#endif

#if SCCBF_HAS_F64
int sccbf_vehicle_step_kin_f64(int64_t A, double* state, const double* u, int32_t mode, double L, double max_steer,
                               double dt, void* stream)
{
    return vehicle_step_kin_impl<double>(A, state, u, mode, L, max_steer, dt, stream);
}
#endif

#if SCCBF_HAS_ROLLOUT_F64
int sccbf_rollout_f64(const sccbf_params* p, const sccbf_nominal_params* np, const sccbf_rollout_params* rp,
                      int64_t A, int32_t M, int32_t N, double* state, const double* traj, int32_t* target_idx,
                      double* pid_state, double* slip, const double* obs, const uint8_t* obs_type, double* h_min,
                      int32_t* n_active, int32_t* n_infeasible, double* trace, void* stream)
{
    return rollout_impl<double>(p, np, rp, A, M, N, state, traj, target_idx, pid_state, slip, obs, obs_type, h_min,
                                n_active, n_infeasible, trace, stream);
}
#endif

}  // extern "C"
