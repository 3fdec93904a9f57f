// Accuracy of the fp32 sin/cos used by the filter kernels (hardware unit on a reduced argument)
// against fp64, over [-pi, pi] and a few multiples: prints the largest absolute errors.
//   hipcc --offload-arch=gfx950 -O3 -o tools/sincos_err tools/sincos_err.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void k(const float* x, float* s, float* c, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float r = __builtin_amdgcn_fractf(x[i] * 0.15915494309189535f);
    s[i] = __builtin_amdgcn_sinf(r);
    c[i] = __builtin_amdgcn_cosf(r);
}

int main()
{
    const int n = 1 << 22;
    std::vector<float> hx(n), hs(n), hc(n);
    for (double span : {3.14159265358979, 20.0, 200.0}) {
        for (int i = 0; i < n; ++i) hx[i] = (float)(-span + 2.0 * span * (i + 0.5) / n);
        float *dx, *ds, *dc;
        hipMalloc(&dx, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dc, n * 4);
        hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, ds, dc, n);
        hipMemcpy(hs.data(), ds, n * 4, hipMemcpyDeviceToHost);
        hipMemcpy(hc.data(), dc, n * 4, hipMemcpyDeviceToHost);
        double es = 0, ec = 0, rs = 0;
        for (int i = 0; i < n; ++i) {
            es = fmax(es, fabs((double)hs[i] - sin((double)hx[i])));
            ec = fmax(ec, fabs((double)hc[i] - cos((double)hx[i])));
            rs += ((double)hs[i] - sin((double)hx[i])) * ((double)hs[i] - sin((double)hx[i]));
        }
        printf("|x| <= %-8g max abs error sin %.3g cos %.3g, rms sin %.3g\n", span, es, ec, sqrt(rs / n));
        hipFree(dx); hipFree(ds); hipFree(dc);
    }
    return 0;
}
