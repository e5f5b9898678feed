// Diagnostic (never part of the product): on which SIMD of its CU does wave w of a four-wave workgroup run, launch shape of the batch
// callback kernels (256 lanes, ~31 KB of LDS: five workgroups per CU, thousands of workgroups)?  If wave w of EVERY workgroup lands on
// SIMD w, the per-wave totals of the knot program are per-SIMD loads: a task table that is balanced phase by phase (for the latency of
// one workgroup) but not in its per-wave sums leaves one SIMD's issue slots as the bound of the batch launches.
// HW_ID (hwreg 4, gfx9 layout): wave slot [3:0], SIMD [5:4], pipe [7:6], CU [11:8], SH [12], SE [15:13].
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/simd_map_probe tools/diag/simd_map_probe.hip && /tmp/simd_map_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int LDS_DOUBLES = 30912 / 8;
__global__ __launch_bounds__(256) void k(unsigned* hwid, double* sink, int spin) {
    __shared__ double buf[LDS_DOUBLES];
    for (int i = threadIdx.x; i < LDS_DOUBLES; i += 256) buf[i] = i;
    __syncthreads();
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    double acc = buf[threadIdx.x];
    for (int r = 0; r < spin; ++r) acc = acc * 1.0000001 + buf[(threadIdx.x + r) % LDS_DOUBLES];   // (a workgroup lives ~10 us, as a knot does)
    if ((threadIdx.x & 63) == 0) hwid[blockIdx.x * 4 + (threadIdx.x >> 6)] = id;
    if (acc == 1.2345) sink[0] = acc;
}
int main() {
    const int wgs = 6464;
    unsigned* d_id; double* d_s;
    CK(hipMalloc(&d_id, wgs * 4 * sizeof(unsigned))); CK(hipMalloc(&d_s, 8));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, d_id, d_s, 600);
    CK(hipDeviceSynchronize());
    std::vector<unsigned> id(wgs * 4);
    CK(hipMemcpy(id.data(), d_id, id.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
    long hist[4][4] = {};
    long same_cu = 0;
    for (int g = 0; g < wgs; ++g) {
        for (int w = 0; w < 4; ++w) hist[w][(id[g * 4 + w] >> 4) & 3]++;
        bool one = true;
        for (int w = 1; w < 4; ++w) one = one && ((id[g * 4 + w] >> 8) & 0xFF) == ((id[g * 4] >> 8) & 0xFF);
        same_cu += one;
    }
    std::printf("%d workgroups of four waves (256 lanes, %d B of LDS); workgroups with all four waves on one CU/SH/SE: %ld\n", wgs, int(LDS_DOUBLES * 8), same_cu);
    std::printf("wave of the workgroup -> SIMD of the CU (counts)\n         SIMD0   SIMD1   SIMD2   SIMD3\n");
    for (int w = 0; w < 4; ++w) std::printf("wave %d  %6ld  %6ld  %6ld  %6ld\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    std::printf("first workgroups: ");
    for (int g = 0; g < 12; ++g) { std::printf("["); for (int w = 0; w < 4; ++w) std::printf("%u", (id[g * 4 + w] >> 4) & 3); std::printf("] "); }
    std::printf("\n");
    return 0;
}
