// Accuracy of the fp32 sin/cos used by the filter kernels (hardware unit on a reduced argument)
// against fp64: largest absolute errors over [-pi, pi] and a few multiples, and the largest RELATIVE error of
// the sine for small angles of either sign (a heading just below zero is the common case on a road along +x),
// for two reductions of the argument in revolutions: into [0, 1) (v_fract) and into [-1/2, 1/2] (r - rint(r)).
//   hipcc --offload-arch=gfx950 -O3 -o tools/sincos_err tools/sincos_err.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

template <int MODE> __global__ void k(const float* x, float* s, float* c, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float t = x[i] * 0.15915494309189535f;
    const float r = MODE == 0 ? __builtin_amdgcn_fractf(t) : t - __builtin_rintf(t);
    s[i] = __builtin_amdgcn_sinf(r);
    c[i] = __builtin_amdgcn_cosf(r);
}

int main()
{
    const int n = 1 << 22;
    std::vector<float> hx(n), hs(n), hc(n);
    float *dx, *ds, *dc;
    hipMalloc(&dx, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dc, n * 4);
    auto run = [&](int mode) {
        hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(n / 256), dim3(256), 0, 0, dx, ds, dc, n);
        else hipLaunchKernelGGL(k<1>, dim3(n / 256), dim3(256), 0, 0, dx, ds, dc, n);
        hipMemcpy(hs.data(), ds, n * 4, hipMemcpyDeviceToHost);
        hipMemcpy(hc.data(), dc, n * 4, hipMemcpyDeviceToHost);
    };
    for (int mode = 0; mode < 2; ++mode) {
        printf("reduction %s\n", mode == 0 ? "into [0, 1): fract(x / 2 pi)" : "into [-1/2, 1/2]: r - rint(r)");
        for (double span : {3.14159265358979, 20.0, 200.0}) {
            for (int i = 0; i < n; ++i) hx[i] = (float)(-span + 2.0 * span * (i + 0.5) / n);
            run(mode);
            double es = 0, ec = 0, rs = 0;
            for (int i = 0; i < n; ++i) {
                es = fmax(es, fabs((double)hs[i] - sin((double)hx[i])));
                ec = fmax(ec, fabs((double)hc[i] - cos((double)hx[i])));
                rs += ((double)hs[i] - sin((double)hx[i])) * ((double)hs[i] - sin((double)hx[i]));
            }
            printf("  |x| <= %-8g max abs error sin %.3g cos %.3g, rms sin %.3g\n", span, es, ec, sqrt(rs / n));
        }
        for (int sign : {1, -1}) {
            // |x| from 1e-6 to 0.1, logarithmically
            for (int i = 0; i < n; ++i) hx[i] = (float)(sign * pow(10.0, -6.0 + 5.0 * (i + 0.5) / n));
            run(mode);
            double er = 0, ea = 0, ecos = 0;
            for (int i = 0; i < n; ++i) {
                const double t = sin((double)hx[i]);
                er = fmax(er, fabs(((double)hs[i] - t) / t));
                ea = fmax(ea, fabs((double)hs[i] - t));
                ecos = fmax(ecos, fabs((double)hc[i] - cos((double)hx[i])));
            }
            printf("  x in %s[1e-6, 0.1]: max relative error of sin %.3g (absolute %.3g), max abs error of cos %.3g\n",
                   sign > 0 ? "+" : "-", er, ea, ecos);
        }
    }
    hipFree(dx); hipFree(ds); hipFree(dc);
    return 0;
}
