// kexp.hip - stand-alone timing experiments for the filter kernels (not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -o tools/kexp tools/kexp.hip && tools/kexp
// Includes the library source to reuse its device code; kernels here ablate stages.
#include "../sccav_cbf_amd/csrc/sccbf.hip"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

namespace {

// K0: ideal streaming read of the same byte count; U loads of one float4 in flight per lane
template <int U>
__global__ __launch_bounds__(256) void k_stream(const float4* __restrict__ in, long long n4, float* out)
{
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long stride = (long long)gridDim.x * 256;
    float acc = 0.f;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = in[i + k * stride];
#pragma unroll
        for (int k = 0; k < U; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
    }
    for (; i < n4; i += stride) {
        float4 v = in[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) out[0] = acc;
}

// K1: the filter's load pattern, no arithmetic beyond a sum.
// VAR 0: as the library kernel; 1: lane g owns slots g*RPL..g*RPL+RPL-1; 2: no ego loads;
// 3: result stored by one lane per wave only; 4: 2 + 3
template <int G, int RPL, int VAR>
__global__ __launch_bounds__(256) void k_loadonly(const KArgs<float> p)
{
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    long long qp = gid / G;
    int g = (int)(gid % G);
    if (VAR == 7) {  // G = 4 only: lane = QP of the tile, wave w owns slots 4 w .. 4 w + 3
        qp = (long long)blockIdx.x * 64 + (threadIdx.x & 63);
        g = threadIdx.x >> 6;
    }
    if (qp >= p.A) return;
    const float4* obs4 = reinterpret_cast<const float4*>(p.obs);
    float4 oa[RPL], ob[RPL];
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
        const int m = VAR == 1 ? g * RPL + r : r * G + g;
        long long i0 = (long long)m * 2 * p.A + qp, i1 = i0 + p.A;
        if (VAR == 5) {  // slot groups interleaved: one contiguous KiB per wave instruction
            i0 = ((long long)(r * 2) * p.A + qp) * G + g;
            i1 = ((long long)(r * 2 + 1) * p.A + qp) * G + g;
        }
        if (VAR == 8) {  // [A/16][M][16][2] float4: slot halves adjacent, one wave instruction pair = 2 KiB in one piece
            i0 = (qp >> 4) * (long long)(p.M * 32) + (long long)m * 32 + (qp & 15) * 2;
            i1 = i0 + 1;
        }
        if (VAR == 7) {  // a wave reads 8 KiB of its tile in one piece
            const long long tb = (qp >> 6) * (long long)(p.M * 2 * 64) + (qp & 63);
            i0 = tb + (long long)((g * RPL + r) * 2) * 64;
            i1 = i0 + 64;
        }
        if (VAR == 6) {  // tiles of 64 agents: [A/64][M][2][64] float4
            const long long tb = (qp >> 6) * (long long)(p.M * 2 * 64) + (qp & 63);
            i0 = tb + (long long)(m * 2) * 64;
            i1 = i0 + 64;
        }
        oa[r] = obs4[i0];
        ob[r] = obs4[i1];
    }
    float acc = 0.f;
    if (VAR != 2 && VAR != 4) {
        const float4 s = reinterpret_cast<const float4*>(p.state)[qp];
        const float2 ur = reinterpret_cast<const float2*>(p.u_ref)[qp];
        acc = s.x + s.y + s.z + s.w + ur.x + ur.y;
    }
#pragma unroll
    for (int r = 0; r < RPL; ++r) acc += oa[r].x + oa[r].y + oa[r].z + oa[r].w + ob[r].x + ob[r].y + ob[r].z + ob[r].w;
    if (VAR != 7) acc = __uint_as_float(group_umax<G>(__float_as_uint(acc)));
    if (VAR == 7) {
        if (g == 0 || acc == 123.456f) { reinterpret_cast<float2*>(p.u_out)[qp] = float2{acc, acc}; p.status[qp] = 0; }
    } else if (VAR >= 3) {
        if ((threadIdx.x & 63) == 0) { reinterpret_cast<float2*>(p.u_out)[qp] = float2{acc, acc}; p.status[qp] = 0; }
    } else if (g == 0) {
        reinterpret_cast<float2*>(p.u_out)[qp] = float2{acc, acc};
        p.status[qp] = 0;
    }
}

// K2: loads + rows (barriers, Lie derivatives, w-space), no QP loop
template <int G, int RPL>
__global__ __launch_bounds__(256) void k_rowsonly(const KArgs<float> p)
{
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long qp = gid / G;
    const int g = (int)(gid % G);
    if (qp >= p.A) return;
    const float4* obs4 = reinterpret_cast<const float4*>(p.obs);
    float4 oa[RPL], ob[RPL];
    int ty[RPL];
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
        const int m = r * G + g;
        const long long i0 = (long long)m * 2 * p.A + qp;
        oa[r] = obs4[i0];
        ob[r] = obs4[i0 + p.A];
        ty[r] = p.obs_type[m];
    }
    Ego<float> e;
    float ur0, ur1;
    int bad;
    load_ego_(p, qp, false, e, ur0, ur1, bad);
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
        float a0, a1, b, h;
        const float o[8] = {oa[r].x, oa[r].y, oa[r].z, oa[r].w, ob[r].x, ob[r].y, ob[r].z, ob[r].w};
        slot_row_(ty[r], o, e, p, a0, a1, b, h);
        float t0, t1, tb;
        to_wspace_(a0, a1, b, e, p, t0, t1, tb);
        acc += t0 + t1 + tb;
    }
    acc = __uint_as_float(group_umax<G>(__float_as_uint(acc)));
    if (g == 0) store_result_(p, qp, false, 0, acc, acc, e, ur0, ur1);
}

// K3: the library's k_group<float, G, RPL> with some workgroups holding their loads back: do a launch's load
// and arithmetic phases overlap better when half of a CU's wavefronts start later?
// MODE picks the late workgroups by index: 1: (i >> 8) & 1, 2: (i >> 9) & 1, 3: i & 1, 4: (i >> 3) & 1, 5: (i >> 7) & 1
template <int G, int RPL, int MODE>
__global__ __launch_bounds__(kBlock) void k_group_late(long long A, const float* obs, const float* state, const float* u_ref,
                                                       const uint8_t* obs_type, int M, unsigned flags,
                                                       const KArgs<float> rest, int naps)
{
    KArgs<float> p = rest;
    p.A = A; p.obs = obs; p.state = state; p.u_ref = u_ref; p.obs_type = obs_type; p.M = M; p.flags = flags;
    const int g = (int)(threadIdx.x % G);
    const long long qp = (long long)blockIdx.x * (kBlock / G) + threadIdx.x / G;
    if (qp >= p.A) return;
    const unsigned i = blockIdx.x;
    const bool late = MODE == 1 ? (i >> 8) & 1 : MODE == 2 ? (i >> 9) & 1 : MODE == 3 ? i & 1 : MODE == 4 ? (i >> 3) & 1 : (i >> 7) & 1;
    if (late)
        for (int n = 0; n < naps; ++n) __builtin_amdgcn_s_sleep(8);   // 8 x 64 cycles
    LaneLoads<float, RPL> L;
    group_load_<float, RPL, false, true>(p, qp, g, G, L);
    group_solve_<float, G, RPL, false>(p, qp, g, L);
}

struct Batch {
    float *state, *u_ref, *obs, *u_out;
    uint8_t* type;
    int* status;
};

template <typename F> float time_it(F launch, int iters, int ring)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 2 * ring; ++i) launch(i % ring);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) launch(i % ring);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters * 1e3f;  // us
}

}  // namespace

int main(int argc, char** argv)
{
    const long long A = argc > 1 ? atoll(argv[1]) : 65536;
    const int M = 16, RING = argc > 2 ? atoi(argv[2]) : 10, ITERS = 400;
    const bool tiled = argc > 3 && atoi(argv[3]) != 0;  // upload obs as [A/64][M][2][64][4]
    const int tmode = argc > 4 ? atoi(argv[4]) : 0;     // slot kinds: 0 = 8 ellipse + 8 cone, 1 all ellipse, 2 all cone, 3 all empty
    const bool quick = argc > 5 && atoi(argv[5]) != 0;  // G = 4 only
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    std::vector<Batch> ring(RING);
    std::vector<float> hs(A * 4), hu(A * 2), ho((size_t)M * 2 * A * 4);
    std::vector<uint8_t> ht(M);
    for (int m = 0; m < M; ++m) ht[m] = tmode == 0 ? (m < 8 ? 0 : 1) : tmode == 1 ? 0 : tmode == 2 ? 1 : 255;
    auto is_ell = [&](int m) { return ht[m] == 0 || (ht[m] == 255 && m < 8); };
    for (int b = 0; b < RING; ++b) {
        for (long long a = 0; a < A; ++a) {
            hs[4 * a] = 50 * U(rng); hs[4 * a + 1] = 50 * U(rng); hs[4 * a + 2] = 3.14f * U(rng); hs[4 * a + 3] = 7.5f + 7.5f * U(rng);
            hu[2 * a] = 8.3f - hs[4 * a + 3]; hu[2 * a + 1] = 0.5f * U(rng);
        }
        for (int m = 0; m < M; ++m)
            for (long long a = 0; a < A; ++a) {
                float rg = 37.5f + 22.5f * U(rng), br = 3.14f * U(rng);
                float* o0 = &ho[(((size_t)m * 2 + 0) * A + a) * 4];
                float* o1 = &ho[(((size_t)m * 2 + 1) * A + a) * 4];
                o0[0] = hs[4 * a] + rg * cosf(br); o0[1] = hs[4 * a + 1] + rg * sinf(br);
                if (is_ell(m)) { o0[2] = 11 + 9 * U(rng); o0[3] = 5.5f + 4.5f * U(rng); o1[0] = 3.14f * U(rng); o1[1] = 2 * U(rng); o1[2] = 2 * U(rng); o1[3] = 0; }
                else { o0[2] = 3.14f * U(rng); o0[3] = 2.5f + 2.5f * U(rng); o1[0] = 4.5f + 2 * U(rng); o1[1] = 0; o1[2] = 0; o1[3] = 0; }
            }
        if (tiled) {
            std::vector<float> hz(ho.size());
            for (int m = 0; m < M; ++m)
                for (int h = 0; h < 2; ++h)
                    for (long long a = 0; a < A; ++a)
                        for (int c = 0; c < 4; ++c)
                            hz[((((size_t)(a / SCCBF_TILE) * M + m) * 2 + h) * SCCBF_TILE + (a % SCCBF_TILE)) * 4 + c] =
                                ho[(((size_t)m * 2 + h) * A + a) * 4 + c];
            ho.swap(hz);
        }
        Batch& B = ring[b];
        CK(hipMalloc(&B.state, A * 16)); CK(hipMalloc(&B.u_ref, A * 8)); CK(hipMalloc(&B.obs, (size_t)M * 2 * A * 16));
        CK(hipMalloc(&B.u_out, A * 8)); CK(hipMalloc(&B.status, A * 4)); CK(hipMalloc(&B.type, M));
        CK(hipMemcpy(B.state, hs.data(), A * 16, hipMemcpyHostToDevice));
        CK(hipMemcpy(B.u_ref, hu.data(), A * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(B.obs, ho.data(), (size_t)M * 2 * A * 16, hipMemcpyHostToDevice));
        CK(hipMemcpy(B.type, ht.data(), M, hipMemcpyHostToDevice));
    }
    sccbf_params sp;
    sccbf_default_params(&sp);
    sp.R[0] = sp.R[3] = 0.5;
    KArgs<float> k;
    if (tiled) sp.flags |= SCCBF_FLAG_OBS_TILED;
    fill_args<float>(&sp, A, M, k);
    const double bytes = (double)A * (4.0 * (4 + 2 + 8 * M + 2) + 4);
    auto report = [&](const char* name, float us) {
        printf("%-28s %8.2f us  %7.1f GB/s  (%.1f%% of 8 TB/s)\n", name, us, bytes / us * 1e-3, bytes / us * 1e-3 / 80.0);
        fflush(stdout);
    };
    auto setk = [&](int b) {
        k.state = ring[b].state; k.u_ref = ring[b].u_ref; k.obs = ring[b].obs; k.obs_type = ring[b].type;
        k.u_out = ring[b].u_out; k.status = ring[b].status; k.rows_out = nullptr;
    };
    printf("layout %s kinds %d  ", tiled ? "tiled" : "planes", tmode);
    printf("A=%lld M=%d ring=%d  algorithmic bytes/launch %.0f\n", A, M, RING, bytes);
    for (int blocks : {256, 512, 1024, 2048, 4096}) {
        if (quick && blocks != 2048) continue;
        char nm[64];
        snprintf(nm, sizeof nm, "stream U=1 grid %d", blocks);
        report(nm, time_it([&](int b) { hipLaunchKernelGGL(k_stream<1>, dim3(blocks), dim3(256), 0, 0, (const float4*)ring[b].obs, (long long)M * 2 * A, ring[b].u_out); }, ITERS, RING));
        snprintf(nm, sizeof nm, "stream U=4 grid %d", blocks);
        report(nm, time_it([&](int b) { hipLaunchKernelGGL(k_stream<4>, dim3(blocks), dim3(256), 0, 0, (const float4*)ring[b].obs, (long long)M * 2 * A, ring[b].u_out); }, ITERS, RING));
        snprintf(nm, sizeof nm, "stream U=8 grid %d", blocks);
        report(nm, time_it([&](int b) { hipLaunchKernelGGL(k_stream<8>, dim3(blocks), dim3(256), 0, 0, (const float4*)ring[b].obs, (long long)M * 2 * A, ring[b].u_out); }, ITERS, RING));
    }
#define RUN_G(G, RPL) \
    report("loadonly G=" #G, time_it([&](int b) { setk(b); hipLaunchKernelGGL((k_loadonly<G, RPL, 0>), dim3((unsigned)((A * G + 255) / 256)), dim3(256), 0, 0, k); }, ITERS, RING)); \
    report("loadonly G=" #G " interleaved", time_it([&](int b) { setk(b); hipLaunchKernelGGL((k_loadonly<G, RPL, 5>), dim3((unsigned)((A * G + 255) / 256)), dim3(256), 0, 0, k); }, ITERS, RING)); \
    report("loadonly G=" #G " tiles64", time_it([&](int b) { setk(b); hipLaunchKernelGGL((k_loadonly<G, RPL, 6>), dim3((unsigned)((A * G + 255) / 256)), dim3(256), 0, 0, k); }, ITERS, RING)); \
    report("loadonly G=" #G " wave-contiguous", time_it([&](int b) { setk(b); hipLaunchKernelGGL((k_loadonly<G, RPL, 7>), dim3((unsigned)((A * G + 255) / 256)), dim3(256), 0, 0, k); }, ITERS, RING)); \
    report("loadonly G=" #G " tile16 slot32", time_it([&](int b) { setk(b); hipLaunchKernelGGL((k_loadonly<G, RPL, 8>), dim3((unsigned)((A * G + 255) / 256)), dim3(256), 0, 0, k); }, ITERS, RING)); \
    report("loadonly G=" #G " slots/lane", time_it([&](int b) { setk(b); hipLaunchKernelGGL((k_loadonly<G, RPL, 1>), dim3((unsigned)((A * G + 255) / 256)), dim3(256), 0, 0, k); }, ITERS, RING)); \
    report("loadonly G=" #G " no ego", time_it([&](int b) { setk(b); hipLaunchKernelGGL((k_loadonly<G, RPL, 2>), dim3((unsigned)((A * G + 255) / 256)), dim3(256), 0, 0, k); }, ITERS, RING)); \
    report("loadonly G=" #G " 1 store/wave", time_it([&](int b) { setk(b); hipLaunchKernelGGL((k_loadonly<G, RPL, 3>), dim3((unsigned)((A * G + 255) / 256)), dim3(256), 0, 0, k); }, ITERS, RING)); \
    report("loadonly G=" #G " no ego, 1 store", time_it([&](int b) { setk(b); hipLaunchKernelGGL((k_loadonly<G, RPL, 4>), dim3((unsigned)((A * G + 255) / 256)), dim3(256), 0, 0, k); }, ITERS, RING)); \
    report("rowsonly G=" #G, time_it([&](int b) { setk(b); hipLaunchKernelGGL((k_rowsonly<G, RPL>), dim3((unsigned)((A * G + 255) / 256)), dim3(256), 0, 0, k); }, ITERS, RING)); \
    report("full     G=" #G, time_it([&](int b) { setk(b); launch_group<float, false, G, RPL>(k, 0); }, ITERS, RING));
    if (!quick) { RUN_G(1, 16) RUN_G(2, 8) }
    RUN_G(4, 4)
    if (!quick) { RUN_G(8, 2) RUN_G(16, 1) }
    if (argc > 6) {  // late workgroups: naps of 512 cycles
#define RUN_LATE(MODE) \
        for (int naps : {0, 2, 4, 6, 8, 12}) { \
            char nm[64]; \
            snprintf(nm, sizeof nm, "late mode " #MODE " naps %d", naps); \
            report(nm, time_it([&](int b) { setk(b); hipLaunchKernelGGL((k_group_late<4, 4, MODE>), dim3((unsigned)((A * 4 + 255) / 256)), dim3(256), 0, 0, \
                k.A, k.obs, k.state, k.u_ref, k.obs_type, k.M, k.flags, k, naps); }, ITERS, RING)); \
        }
        RUN_LATE(1) RUN_LATE(2) RUN_LATE(3) RUN_LATE(4) RUN_LATE(5)
    }
    if (argc > 6) {  // a burst of 20 launches from an idle GPU, dealt eagerly to n streams forked from / joined to stream 0
        for (int ns : {1, 2, 3, 4}) {
            hipStream_t st[4];
            hipEvent_t ev[4];
            for (int j = 0; j < ns; ++j) { CK(hipStreamCreate(&st[j])); CK(hipEventCreateWithFlags(&ev[j], hipEventDisableTiming)); }
            std::vector<double> wall, host;
            for (int rep = 0; rep < 60; ++rep) {
                for (int i = 0; i < 30; ++i) { setk(i % RING); launch_group<float, false, 4, 4>(k, st[0]); }  // clocks up
                CK(hipDeviceSynchronize());
                const auto t0 = std::chrono::steady_clock::now();
                if (ns > 1) {
                    CK(hipEventRecord(ev[0], st[0]));
                    for (int j = 1; j < ns; ++j) CK(hipStreamWaitEvent(st[j], ev[0], 0));
                }
                for (int i = 0; i < 20; ++i) { setk(i % RING); launch_group<float, false, 4, 4>(k, st[i % ns]); }
                for (int j = 1; j < ns; ++j) { CK(hipEventRecord(ev[j], st[j])); CK(hipStreamWaitEvent(st[0], ev[j], 0)); }
                const auto t1 = std::chrono::steady_clock::now();
                CK(hipStreamSynchronize(st[0]));
                const auto t2 = std::chrono::steady_clock::now();
                if (rep >= 10) {
                    host.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
                    wall.push_back(std::chrono::duration<double, std::micro>(t2 - t0).count());
                }
            }
            std::sort(wall.begin(), wall.end());
            std::sort(host.begin(), host.end());
            printf("burst of 20 eager launches on %d stream(s): %.1f us to the end of the synchronize (median; min %.1f), host calls %.1f us\n",
                   ns, wall[wall.size() / 2], wall[0], host[host.size() / 2]);
            fflush(stdout);
        }
    }
    CK(hipDeviceSynchronize());
    int st[8];
    CK(hipMemcpy(st, ring[0].status, sizeof st, hipMemcpyDeviceToHost));
    printf("status[0..7] = %d %d %d %d %d %d %d %d\n", st[0], st[1], st[2], st[3], st[4], st[5], st[6], st[7]);
    return 0;
}
