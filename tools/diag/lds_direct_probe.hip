// Diagnostic (never part of the product): (1) where does global_load_lds_dwordx4 put each lane's 16 bytes; (2) what does a cold
// straight-line prologue cost per instruction at kernel start (s_memtime around N independent cheap instructions).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(512) void k_layout(const double* src, double* out) {
    __shared__ __attribute__((aligned(16))) double buf[2048];
    const int tid = threadIdx.x;
    for (int i = tid; i < 2048; i += 512) buf[i] = -1.0;
    __syncthreads();
    // wave w copies the 1 KB chunks w and w + 8: lane l -> 16 bytes at chunk base + 16 l (global side: src + 1 double, i.e. 8-byte aligned only)
    for (int c = (tid >> 6); c < 15; c += 8) {
        const double* g = src + 1 + c * 128 + (tid & 63) * 2;
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g, (void __attribute__((address_space(3)))*)(buf + c * 128), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < 2048; i += 512) out[i] = buf[i];
}

template <int N> __global__ __launch_bounds__(512) void k_cold(unsigned long long* stamps, int* sink, int a, int b) {
    int v = threadIdx.x + a;
    unsigned long long t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "+v"(v) : : "memory");   // (the chain starts behind it)
#pragma unroll
    for (int i = 0; i < N; ++i) v = (v ^ (b + i)) + (v >> 3);     // 3 dependent cheap VALU instructions per step
    unsigned long long t1;
    asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(v) : "memory");   // (ordered behind the chain through its input)
    if (v == 0x7fffffff) sink[0] = v;
    if ((threadIdx.x & 63) == 0) { stamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2] = t0; stamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2 + 1] = t1; }
}

template <int N> int cold(const char* name, unsigned long long* d_st, int* d_sink) {
    std::vector<unsigned long long> st(100 * 16);
    double first = 0, rest = 0;
    for (int rep = 0; rep < 6; ++rep) {
        hipLaunchKernelGGL(k_cold<N>, dim3(100), dim3(512), 0, 0, d_st, d_sink, rep, 7);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
        double s = 0;
        for (int i = 0; i < 800; ++i) s += double(st[2 * i + 1] - st[2 * i]);
        if (rep == 0) first = s / 800; else rest += s / 800 / 5;
    }
    printf("%s: %d steps (3 VALU each): cycles per wave, first launch %.0f, later launches %.0f  -> %.2f cycles / instruction\n", name, N, first, rest, rest / (3.0 * N));
    return 0;
}

int main() {
    double *d_src, *d_out;
    std::vector<double> src(4096), out(2048);
    for (int i = 0; i < 4096; ++i) src[i] = i;
    CK(hipMalloc(&d_src, 4096 * 8)); CK(hipMalloc(&d_out, 2048 * 8));
    CK(hipMemcpy(d_src, src.data(), 4096 * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(512), 0, 0, d_src, d_out);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out.data(), d_out, 2048 * 8, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 15 * 128; ++i) if (out[i] != double(i + 1)) { if (bad < 8) printf("  buf[%d] = %g (expected %d)\n", i, out[i], i + 1); ++bad; }
    for (int i = 15 * 128; i < 2048; ++i) if (out[i] != -1.0) { if (bad < 8) printf("  buf[%d] = %g (expected -1)\n", i, out[i]); ++bad; }
    printf("global_load_lds_dwordx4: lane l of a wave -> LDS base + 16 l, 8-byte aligned global source: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
    unsigned long long* d_st; int* d_sink;
    CK(hipMalloc(&d_st, 100 * 16 * 8)); CK(hipMalloc(&d_sink, 4));
    if (cold<16>("cold16", d_st, d_sink)) return 1;
    if (cold<64>("cold64", d_st, d_sink)) return 1;
    if (cold<256>("cold256", d_st, d_sink)) return 1;
    if (cold<1024>("cold1024", d_st, d_sink)) return 1;
    return 0;
}
