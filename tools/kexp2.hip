// kexp2.hip - stand-alone load-pattern experiments (not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/kexp2 tools/kexp2.hip && tools/kexp2 [ring [streams]]
// Every kernel reads the same 32 MiB (65536 agents x 16 slots x 32 B) and writes 12 B per agent; the
// variants differ in how the bytes are spread over wavefronts and load instructions.  Each wavefront
// stamps its start and end with the 100 MHz wall clock, so the in-kernel time (first start -> last end)
// is separated from what a launch costs around it.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float raw4 __attribute__((ext_vector_type(4)));

__device__ inline float sum4(float4 v) { return v.x + v.y + v.z + v.w; }

// MODE 0: a wave reads NL consecutive KiB (lane i: 16 B at i * 16 of each KiB)
// MODE 1: the 4-lanes-per-QP pattern on 64-agent tiles: per instruction 4 pieces of 256 B, 2 KiB apart
// MODE 2: as 0 with non-temporal loads
// MODE 3: as 0, the loads in two halves with a wait between
template <int NL, int MODE, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_pat(const float4* __restrict__ in, float* __restrict__ out,
                                               unsigned long long* __restrict__ stamps)
{
    const unsigned long long t0 = wall_clock64();
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * BLOCK + threadIdx.x) >> 6;
    float4 v[NL];
    if (MODE == 1) {
        // tiles of 64 agents x 16 slots x 2 halves: [tile][m][h][64]; wave w of a tile = agents 16 w .. 16 w + 15
        static_assert(MODE != 1 || NL == 8, "pattern 1 is the 4 x 4 kernel");
        const long long tile = wave >> 2;
        const int q = (int)(wave & 3) * 16 + (lane >> 2), g = lane & 3;
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int r = k >> 1, h = k & 1;
            v[k] = in[tile * 2048 + (long long)((r * 4 + g) * 2 + h) * 64 + q];
        }
    } else {
        const float4* base = in + wave * (NL * 64) + lane;
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            if (MODE == 3 && k == NL / 2) __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0)
            if (MODE == 2) {
                const raw4 r = __builtin_nontemporal_load(reinterpret_cast<const raw4*>(base + k * 64));
                v[k] = float4{r.x, r.y, r.z, r.w};
            } else {
                v[k] = base[k * 64];
            }
        }
    }
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < NL; ++k) acc += sum4(v[k]);
    // 12 B per agent: the agents of this wave's share
    const long long agents_per_wave = NL * 2;  // NL KiB = NL * 32 slots = 2 NL agents
    if (lane < agents_per_wave) {
        reinterpret_cast<float2*>(out)[wave * agents_per_wave + lane] = float2{acc, acc};
        out[131072 + wave * agents_per_wave + lane] = acc;
    }
    // the end stamp must not be read before the loads have landed: it takes the sum as an operand
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(acc) : "memory");
    const unsigned long long t1 = wall_clock64();
    asm volatile("" ::"s"(t1), "v"(acc) : "memory");
    if (lane == 0) {
        stamps[2 * wave] = t0;
        stamps[2 * wave + 1] = t1;
    }
}

// grid-stride stream, U loads in flight per lane
template <int U> __global__ __launch_bounds__(256) void k_stream(const float4* __restrict__ in, long long n4, float* out)
{
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long stride = (long long)gridDim.x * 256;
    float acc = 0.f;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = in[i + k * stride];
#pragma unroll
        for (int k = 0; k < U; ++k) acc += sum4(v[k]);
    }
    for (; i < n4; i += stride) acc += sum4(in[i]);
    if (acc == 123.456f) out[0] = acc;
}

int main(int argc, char** argv)
{
    const int RING = argc > 1 ? atoi(argv[1]) : 10;
    const int NS = argc > 2 ? atoi(argv[2]) : 1;  // streams the launches are dealt to (independent launches overlap)
    std::vector<hipStream_t> streams(NS);
    for (int i = 0; i < NS; ++i) CK(hipStreamCreate(&streams[i]));
    const long long N4 = 65536LL * 16 * 2;  // float4 elements: 32 MiB
    std::vector<float4*> in(RING);
    std::vector<float> h(N4 * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 977) * 1e-3f;
    for (int b = 0; b < RING; ++b) {
        CK(hipMalloc(&in[b], N4 * 16));
        CK(hipMemcpy(in[b], h.data(), N4 * 16, hipMemcpyHostToDevice));
    }
    float* out;
    CK(hipMalloc(&out, 65536 * 12 + 4096));
    unsigned long long* stamps;
    const int MAXW = 65536;
    CK(hipMalloc(&stamps, MAXW * 16));
    std::vector<unsigned long long> hs(MAXW * 2);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("ring %d (%s)\n", RING, RING * 32 > 256 ? "HBM" : "Infinity Cache");

    hipStream_t cur = 0;
    auto run = [&](const char* name, int waves, auto launch) {
        for (int i = 0; i < 3 * RING + 20; ++i) { cur = NS > 1 ? streams[i % NS] : 0; launch(i % RING); }
        CK(hipDeviceSynchronize());
        const int iters = NS > 1 ? 3000 : 300;
        float ms;
        if (NS > 1) {  // several streams: wall clock around the launches and the final synchronize
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < iters; ++i) { cur = streams[i % NS]; launch(i % RING); }
            CK(hipDeviceSynchronize());
            ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        } else {
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < iters; ++i) launch(i % RING);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        double in_kernel = 0, p50end = 0, p10end = 0;
        if (waves > 0) {
            // stamps of the last launch (every launch overwrites them)
            CK(hipMemcpy(hs.data(), stamps, (size_t)waves * 16, hipMemcpyDeviceToHost));
            unsigned long long s0 = ~0ull, e_max = 0;
            std::vector<unsigned long long> ends(waves);
            for (int w = 0; w < waves; ++w) {
                s0 = std::min(s0, hs[2 * w]);
                e_max = std::max(e_max, hs[2 * w + 1]);
                ends[w] = hs[2 * w + 1];
            }
            std::sort(ends.begin(), ends.end());
            in_kernel = (e_max - s0) * 0.01;
            p50end = (ends[waves / 2] - s0) * 0.01;
            p10end = (ends[waves / 10] - s0) * 0.01;
        }
        printf("%-44s %6.2f us/launch   in-kernel %5.2f us (ends p10 %5.2f p50 %5.2f)\n", name, ms / iters * 1e3, in_kernel,
               p10end, p50end);
        fflush(stdout);
    };
#define PAT(NL, MODE, BLOCK, label)                                                                         \
    run(label, (int)(N4 / (NL * 64)), [&](int b) {                                                          \
        hipLaunchKernelGGL((k_pat<NL, MODE, BLOCK>), dim3((unsigned)(N4 / (NL * BLOCK))), dim3(BLOCK), 0, cur, \
                           in[b], out, stamps);                                                             \
    });
    run("stream U=1 grid 2048", 0, [&](int b) { hipLaunchKernelGGL(k_stream<1>, dim3(2048), dim3(256), 0, cur, in[b], N4, out); });
    run("stream U=8 grid 2048", 0, [&](int b) { hipLaunchKernelGGL(k_stream<8>, dim3(2048), dim3(256), 0, cur, in[b], N4, out); });
    run("stream U=4 grid 512", 0, [&](int b) { hipLaunchKernelGGL(k_stream<4>, dim3(512), dim3(256), 0, cur, in[b], N4, out); });
    PAT(8, 0, 256, "contiguous 8 KiB/wave, 4096 waves, block 256")
    PAT(8, 1, 256, "4-lane pattern on tiles64, 4096 waves")
    PAT(8, 2, 256, "contiguous 8 KiB/wave, nt")
    PAT(8, 3, 256, "contiguous 8 KiB/wave, two halves")
    PAT(8, 0, 64, "contiguous 8 KiB/wave, block 64")
    PAT(8, 0, 1024, "contiguous 8 KiB/wave, block 1024")
    PAT(4, 0, 256, "contiguous 4 KiB/wave, 8192 waves")
    PAT(2, 0, 256, "contiguous 2 KiB/wave, 16384 waves")
    PAT(1, 0, 256, "contiguous 1 KiB/wave, 32768 waves")
    PAT(16, 0, 256, "contiguous 16 KiB/wave, 2048 waves")
    PAT(4, 2, 256, "contiguous 4 KiB/wave, nt")
    {   // one launch over 1 GiB (32 buffers' worth in one allocation): the HBM rate without launch effects
        float4* big;
        const long long BN4 = 32 * N4;
        CK(hipMalloc(&big, BN4 * 16));
        CK(hipMemset(big, 1, BN4 * 16));
        float* bout;
        CK(hipMalloc(&bout, 32 * (65536 * 12 + 4096)));
        unsigned long long* bst;
        CK(hipMalloc(&bst, 32LL * 4096 * 16));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL((k_pat<8, 0, 256>), dim3((unsigned)(BN4 / (8 * 256))), dim3(256), 0, 0, big, bout, bst);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("one launch over 1 GiB, contiguous 8 KiB/wave: %.1f us = %.0f GB/s\n", ms * 1e3, BN4 * 16.0 / ms * 1e-6);
        }
    }
    return 0;
}
