// Diagnostic (never part of the product): LDS-array cost of the read forms the knot kernels use, per wave-instruction, with four
// waves per SIMD on one CU (the batch regime): ds_read_b64, ds_read2_b64 of two ADJACENT doubles at 8-byte alignment, two separate
// ds_read_b64 of the same pair, ds_read_b128 at 16-byte alignment; lane strides of 3 and 9 doubles (3-vectors, 3 x 3 rows).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int REP = 64;
template <int KIND> __global__ __launch_bounds__(1024) void k(unsigned long long* cyc, double* sink, int stride_doubles) {
    __shared__ __attribute__((aligned(16))) double buf[8192];
    for (int i = threadIdx.x; i < 8192; i += 1024) buf[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned addr = unsigned(lane * stride_doubles * 8) % (8192 * 8 - 64);
    if (KIND == 3) addr &= ~15u;
    double acc = 0;
    unsigned long long t0, t1;
    asm volatile("s_barrier\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int r = 0; r < REP; ++r) {
        double a0, a1, a2, a3;
        if (KIND == 0) { asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:8\n\ts_waitcnt lgkmcnt(0)" : "=v"(a0), "=v"(a1) : "v"(addr) : "memory"); acc += a0 + a1; }
        if (KIND == 1) { double2 q; asm volatile("ds_read2_b64 %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(addr) : "memory"); acc += q.x + q.y; }
        if (KIND == 2) { asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(a0) : "v"(addr) : "memory"); acc += a0; }
        if (KIND == 3) { double2 q; asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(addr) : "memory"); acc += q.x + q.y; }
        if (KIND == 4) { asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:8\n\tds_read_b64 %2, %4 offset:16\n\tds_read_b64 %3, %4 offset:24\n\ts_waitcnt lgkmcnt(0)" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(addr) : "memory"); acc += a0 + a1 + a2 + a3; }
        if (KIND == 6) { asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %1 offset:8\n\ts_waitcnt lgkmcnt(0)" :: "v"(addr), "v"(acc) : "memory"); }
        if (KIND == 7) { asm volatile("ds_write2_b64 %0, %1, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)" :: "v"(addr), "v"(acc) : "memory"); }
        if (KIND == 8) { asm volatile("ds_write_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" :: "v"(addr), "v"(acc) : "memory"); }
        if (KIND == 5) { double2 q, p; asm volatile("ds_read2_b64 %0, %2 offset1:1\n\tds_read2_b64 %1, %2 offset0:2 offset1:3\n\ts_waitcnt lgkmcnt(0)" : "=v"(q), "=v"(p) : "v"(addr) : "memory"); acc += q.x + q.y + p.x + p.y; }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    sink[threadIdx.x] = acc;
    if (lane == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}
template <int KIND> int run(const char* name, int stride, unsigned long long* d_c, double* d_s) {
    hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(1024), 0, 0, d_c, d_s, stride);
    hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(1024), 0, 0, d_c, d_s, stride);
    CK(hipDeviceSynchronize());
    unsigned long long c[16];
    CK(hipMemcpy(c, d_c, sizeof c, hipMemcpyDeviceToHost));
    double m = 0; for (int i = 0; i < 16; ++i) m = c[i] > m ? c[i] : m;
    // 16 waves x REP groups on one CU's LDS: cycles per wave-group if the array serialises them
    std::printf("  %-44s lane stride %d doubles: %6.1f cycles per group per wave (16 waves on the CU)\n", name, stride, m / REP / 16.0);
    return 0;
}
int main() {
    unsigned long long* d_c; double* d_s;
    CK(hipMalloc(&d_c, 16 * 8)); CK(hipMalloc(&d_s, 1024 * 8));
    for (int stride : {1, 3, 9, 17}) {
        if (run<2>("one ds_read_b64", stride, d_c, d_s)) return 1;
        if (run<0>("two ds_read_b64 (adjacent doubles)", stride, d_c, d_s)) return 1;
        if (run<1>("one ds_read2_b64 (the same two doubles)", stride, d_c, d_s)) return 1;
        if (run<3>("one ds_read_b128 (16-byte aligned)", stride, d_c, d_s)) return 1;
        if (run<4>("four ds_read_b64 (four adjacent doubles)", stride, d_c, d_s)) return 1;
        if (run<5>("two ds_read2_b64 (the same four doubles)", stride, d_c, d_s)) return 1;
        if (run<8>("one ds_write_b64", stride, d_c, d_s)) return 1;
        if (run<6>("two ds_write_b64 (adjacent doubles)", stride, d_c, d_s)) return 1;
        if (run<7>("one ds_write2_b64 (the same two doubles)", stride, d_c, d_s)) return 1;
    }
    return 0;
}
