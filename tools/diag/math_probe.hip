// Diagnostic (never part of the product): latency of the fp64 library calls on the critical path of phase A (sincos in the joint
// transforms, tanh in the planar complementarity rows) against bounded-range closed forms, and the error of both against the host libm.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#include "fast_math.h"

template <int WHICH> __global__ __launch_bounds__(512) void k_time(const double* in, double* out, unsigned long long* cyc) {
    double x = in[threadIdx.x];
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "+v"(x) : : "memory");
    double a = 0, b = 0;
    if (WHICH == 0) sincos(x, &a, &b);
    if (WHICH == 1) hipnlp::fast_sincos(x, &a, &b);
    if (WHICH == 2) a = tanh(x);
    if (WHICH == 3) a = hipnlp::fast_tanh(x);
    double r = a + b;
    asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "+v"(r) : : "memory");
    out[2 * threadIdx.x] = a; out[2 * threadIdx.x + 1] = b;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int WHICH> __global__ void k_eval(const double* in, double* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a = 0, b = 0;
    if (WHICH == 0) sincos(in[i], &a, &b);
    if (WHICH == 1) hipnlp::fast_sincos(in[i], &a, &b);
    if (WHICH == 2) a = tanh(in[i]);
    if (WHICH == 3) a = hipnlp::fast_tanh(in[i]);
    out[2 * i] = a; out[2 * i + 1] = b;
}

static double ulp_err(double got, long double want) {
    if (want == 0) return got == 0 ? 0 : 1e300;
    int e; frexp((double)want, &e);
    return (double)(fabsl((long double)got - want) / ldexpl(1.0L, e - 53));
}

int main() {
    const char* names[4] = {"ocml sincos", "fast_sincos", "ocml tanh", "fast_tanh"};
    std::vector<double> in(512); for (int i = 0; i < 512; ++i) in[i] = -1.5 + 0.006 * i;
    double *d_in, *d_out; unsigned long long* d_c;
    CK(hipMalloc(&d_in, 512 * 8)); CK(hipMalloc(&d_out, 1024 * 8)); CK(hipMalloc(&d_c, 64));
    CK(hipMemcpy(d_in, in.data(), 512 * 8, hipMemcpyHostToDevice));
    for (int w = 0; w < 4; ++w) {
        unsigned long long c[8]; double best = 1e30;
        for (int rep = 0; rep < 5; ++rep) {
            if (w == 0) hipLaunchKernelGGL(k_time<0>, dim3(1), dim3(512), 0, 0, d_in, d_out, d_c);
            if (w == 1) hipLaunchKernelGGL(k_time<1>, dim3(1), dim3(512), 0, 0, d_in, d_out, d_c);
            if (w == 2) hipLaunchKernelGGL(k_time<2>, dim3(1), dim3(512), 0, 0, d_in, d_out, d_c);
            if (w == 3) hipLaunchKernelGGL(k_time<3>, dim3(1), dim3(512), 0, 0, d_in, d_out, d_c);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(c, d_c, 64, hipMemcpyDeviceToHost));
            double s = 0; for (int i = 0; i < 8; ++i) s += double(c[i]);
            if (s / 8 < best) best = s / 8;
        }
        printf("%-12s %6.0f cycles per call (8 waves on one CU, two per SIMD)\n", names[w], best);
    }
    // accuracy against the host's long double libm
    const int n = 1 << 20;
    std::vector<double> xs(n), out(2 * n);
    double *dx, *dout; CK(hipMalloc(&dx, n * 8)); CK(hipMalloc(&dout, 2 * n * 8));
    struct { const char* what; double lo, hi; } ranges[] = {{"[-pi, pi]", -3.1415926, 3.1415926}, {"[-20, 20]", -20, 20}, {"[-1e4, 1e4]", -1e4, 1e4}, {"[-1e9, 1e9]", -1e9, 1e9}};
    for (auto& rg : ranges) {
        unsigned long long st = 88172645463325252ull;
        for (int i = 0; i < n; ++i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; xs[i] = rg.lo + (rg.hi - rg.lo) * double(st >> 11) * (1.0 / 9007199254740992.0); }
        CK(hipMemcpy(dx, xs.data(), n * 8, hipMemcpyHostToDevice));
        for (int w = 0; w < 4; ++w) {
            if (w == 0) hipLaunchKernelGGL(k_eval<0>, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
            if (w == 1) hipLaunchKernelGGL(k_eval<1>, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
            if (w == 2) hipLaunchKernelGGL(k_eval<2>, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
            if (w == 3) hipLaunchKernelGGL(k_eval<3>, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(out.data(), dout, 2 * n * 8, hipMemcpyDeviceToHost));
            double worst = 0;
            for (int i = 0; i < n; ++i) {
                if (w < 2) { worst = fmax(worst, ulp_err(out[2 * i], sinl((long double)xs[i]))); worst = fmax(worst, ulp_err(out[2 * i + 1], cosl((long double)xs[i]))); }
                else worst = fmax(worst, ulp_err(out[2 * i], tanhl((long double)xs[i])));
            }
            printf("  %-12s on %-12s max error %.3f ulp\n", names[w], rg.what, worst);
        }
    }
    return 0;
}
