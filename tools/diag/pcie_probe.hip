// pcie_probe.hip — GPU box diagnostic (never in the product library): what the host-visible callback path can cost.
// Times, per call (median of REPS, wall clock): launch + synchronise, completion through a host flag, a kernel storing its
// outputs straight into pinned / registered host memory against a device write + hipMemcpyAsync, a kernel reading x from host
// memory against an H2D copy, and the host-side memcpy out of the pinned block.
//   hipcc --offload-arch=gfx950 -O3 -o tools/diag/_build/pcie_probe tools/diag/pcie_probe.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); std::exit(1); } } while (0)

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// spin `cycles` ticks of the 100 MHz real-time counter (emulates the knot program), then every workgroup streams `per_wg` doubles to out (coalesced), and
// optionally reads `rd_per_wg` doubles of x first; the last workgroup to finish raises *flag = seq (system scope)
__global__ __launch_bounds__(512) void work_kernel(const double* x, int rd_per_wg, double* out, long per_wg, long cycles, unsigned* counter,
                                                   volatile int* flag, int seq, int scatter_run) {
    __shared__ double sx[512];
    double acc = 0.0;
    if (x && rd_per_wg > 0) {
        for (int i = threadIdx.x; i < rd_per_wg; i += blockDim.x) acc += x[size_t(blockIdx.x) * rd_per_wg + i];
        sx[threadIdx.x] = acc;
        __syncthreads();
        acc = sx[(threadIdx.x + 1) & 511];
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // (100 MHz; s_memtime counts shader clocks)
    while (long(__builtin_amdgcn_s_memrealtime() - t0) < cycles) { }
    double* o = out + size_t(blockIdx.x) * per_wg;
    if (scatter_run <= 0) {
        for (long i = threadIdx.x; i < per_wg; i += blockDim.x) o[i] = acc + double(i);
    } else {   // runs of scatter_run doubles, strided by the number of workgroups (the type-major g layout)
        for (long i = threadIdx.x; i < per_wg; i += blockDim.x) {
            const long run = i / scatter_run, within = i % scatter_run;
            out[(run * gridDim.x + blockIdx.x) * scatter_run + within] = acc + double(i);
        }
    }
    if (flag) {
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned old = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1 == unsigned(seq) * gridDim.x) __hip_atomic_store((int*)flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

template <class F> static double median_us(int reps, F&& f) {
    std::vector<double> t;
    for (int i = 0; i < reps + 10; ++i) {
        const double a = now_us();
        f();
        const double b = now_us();
        if (i >= 10) t.push_back(b - a);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main() {
    const int REPS = 300, WGS = 100;
    CK(hipSetDevice(0));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const size_t NX = 18906, MAXOUT = 200000;   // doubles: x of 100 knots; f + grad + g + jac
    double *d_x, *d_out, *h_x, *h_out_c, *h_out_nc, *h_reg;
    unsigned* d_counter;
    int* h_flag;
    CK(hipMalloc(&d_x, NX * 8 * 2));
    CK(hipMalloc(&d_out, MAXOUT * 8));
    CK(hipMalloc(&d_counter, 64));
    CK(hipMemset(d_counter, 0, 64));
    CK(hipHostMalloc(&h_x, NX * 8 * 2, hipHostMallocDefault));
    CK(hipHostMalloc(&h_out_c, MAXOUT * 8, hipHostMallocDefault));
    CK(hipHostMalloc(&h_out_nc, MAXOUT * 8, hipHostMallocNonCoherent));
    CK(hipHostMalloc(&h_flag, 64, hipHostMallocDefault));
    h_reg = static_cast<double*>(std::aligned_alloc(4096, ((MAXOUT * 8 + 4095) / 4096) * 4096));
    std::memset(h_reg, 0, MAXOUT * 8);
    CK(hipHostRegister(h_reg, MAXOUT * 8, hipHostRegisterDefault));
    double* h_reg_dev = nullptr;
    CK(hipHostGetDevicePointer(reinterpret_cast<void**>(&h_reg_dev), h_reg, 0));
    double* user = static_cast<double*>(std::malloc(MAXOUT * 8));
    std::memset(user, 0, MAXOUT * 8);
    for (size_t i = 0; i < NX * 2; ++i) h_x[i] = double(i);
    *h_flag = 0;
    int seq = 0;
    const long CYC = 800;   // ticks of the 100 MHz real-time counter: 8 us of "knot program"

    std::printf("== launch + completion\n");
    std::printf("empty kernel + hipStreamSynchronize            %7.1f us\n", median_us(REPS, [&] {
        hipLaunchKernelGGL(work_kernel, dim3(WGS), dim3(512), 0, s, nullptr, 0, d_out, 0L, 0L, d_counter, nullptr, 0, 0);
        CK(hipStreamSynchronize(s)); }));
    CK(hipMemset(d_counter, 0, 64));
    seq = 0;
    std::printf("empty kernel + spin on a host flag              %7.1f us\n", median_us(REPS, [&] {
        ++seq;
        hipLaunchKernelGGL(work_kernel, dim3(WGS), dim3(512), 0, s, nullptr, 0, d_out, 0L, 0L, d_counter, h_flag, seq, 0);
        while (__atomic_load_n(h_flag, __ATOMIC_ACQUIRE) != seq) { } }));
    CK(hipStreamSynchronize(s));
    std::printf("8 us kernel + hipStreamSynchronize              %7.1f us\n", median_us(REPS, [&] {
        hipLaunchKernelGGL(work_kernel, dim3(WGS), dim3(512), 0, s, nullptr, 0, d_out, 0L, CYC, d_counter, nullptr, 0, 0);
        CK(hipStreamSynchronize(s)); }));

    std::printf("== x upload (151 KB)\n");
    std::printf("memcpy to pinned + H2D copy + 8 us kernel + sync %6.1f us\n", median_us(REPS, [&] {
        std::memcpy(h_x, user, NX * 8);
        CK(hipMemcpyAsync(d_x, h_x, NX * 8, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(work_kernel, dim3(WGS), dim3(512), 0, s, d_x, 378, d_out, 0L, CYC, d_counter, nullptr, 0, 0);
        CK(hipStreamSynchronize(s)); }));
    std::printf("memcpy to pinned + kernel reads host x (2x) + sync %5.1f us\n", median_us(REPS, [&] {
        std::memcpy(h_x, user, NX * 8);
        hipLaunchKernelGGL(work_kernel, dim3(WGS), dim3(512), 0, s, h_x, 378, d_out, 0L, CYC, d_counter, nullptr, 0, 0);
        CK(hipStreamSynchronize(s)); }));

    const long sizes[] = {1, 27500, 27500 + 18906 + 1, 137879, 137879 + 27500 + 18906 + 1};
    const char* names[] = {"f (8 B)", "g (220 KB)", "f+grad+g (371 KB)", "jac (1.1 MB)", "all (1.47 MB)"};
    for (int si = 0; si < 5; ++si) {
        const long per = (sizes[si] + WGS - 1) / WGS;
        const size_t bytes = size_t(per) * WGS * 8;
        std::printf("== outputs: %s\n", names[si]);
        std::printf("device write + D2H copy to pinned + sync         %7.1f us\n", median_us(REPS, [&] {
            hipLaunchKernelGGL(work_kernel, dim3(WGS), dim3(512), 0, s, nullptr, 0, d_out, per, CYC, d_counter, nullptr, 0, 0);
            CK(hipMemcpyAsync(h_out_c, d_out, bytes, hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s)); }));
        std::printf("device write + D2H copy to registered + sync     %7.1f us\n", median_us(REPS, [&] {
            hipLaunchKernelGGL(work_kernel, dim3(WGS), dim3(512), 0, s, nullptr, 0, d_out, per, CYC, d_counter, nullptr, 0, 0);
            CK(hipMemcpyAsync(h_reg, d_out, bytes, hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s)); }));
        std::printf("kernel stores to pinned (coherent) + sync        %7.1f us\n", median_us(REPS, [&] {
            hipLaunchKernelGGL(work_kernel, dim3(WGS), dim3(512), 0, s, nullptr, 0, h_out_c, per, CYC, d_counter, nullptr, 0, 0);
            CK(hipStreamSynchronize(s)); }));
        std::printf("kernel stores to pinned (non-coherent) + sync    %7.1f us\n", median_us(REPS, [&] {
            hipLaunchKernelGGL(work_kernel, dim3(WGS), dim3(512), 0, s, nullptr, 0, h_out_nc, per, CYC, d_counter, nullptr, 0, 0);
            CK(hipStreamSynchronize(s)); }));
        std::printf("kernel stores to registered malloc + sync        %7.1f us\n", median_us(REPS, [&] {
            hipLaunchKernelGGL(work_kernel, dim3(WGS), dim3(512), 0, s, nullptr, 0, h_reg_dev, per, CYC, d_counter, nullptr, 0, 0);
            CK(hipStreamSynchronize(s)); }));
        CK(hipMemset(d_counter, 0, 64));
        seq = 0;
        *h_flag = 0;
        std::printf("kernel stores to pinned + host-flag completion   %7.1f us\n", median_us(REPS, [&] {
            ++seq;
            hipLaunchKernelGGL(work_kernel, dim3(WGS), dim3(512), 0, s, nullptr, 0, h_out_c, per, CYC, d_counter, h_flag, seq, 0);
            while (__atomic_load_n(h_flag, __ATOMIC_ACQUIRE) != seq) { } }));
        CK(hipStreamSynchronize(s));
        if (si == 1) {
            for (int run : {3, 8, 23}) {
                std::printf("kernel stores to pinned, runs of %2d doubles + sync %6.1f us\n", run, median_us(REPS, [&] {
                    hipLaunchKernelGGL(work_kernel, dim3(WGS), dim3(512), 0, s, nullptr, 0, h_out_c, (per / run) * run, CYC, d_counter, nullptr, 0, run);
                    CK(hipStreamSynchronize(s)); }));
            }
        }
        std::printf("host memcpy pinned -> caller array               %7.1f us\n", median_us(REPS, [&] { std::memcpy(user, h_out_c, bytes); }));
    }
    // correctness of the direct stores (read back what the last kernel wrote to registered memory)
    {
        const long per = (sizes[4] + WGS - 1) / WGS;
        hipLaunchKernelGGL(work_kernel, dim3(WGS), dim3(512), 0, s, nullptr, 0, h_reg_dev, per, 0L, d_counter, nullptr, 0, 0);
        CK(hipStreamSynchronize(s));
        long bad = 0;
        for (long w = 0; w < WGS; ++w) for (long i = 0; i < per; ++i) bad += h_reg[w * per + i] != double(i);
        std::printf("direct stores to registered memory verified: %ld mismatches\n", bad);
    }
    return 0;
}
