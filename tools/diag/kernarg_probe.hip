// Diagnostic (never part of the product): how long does a workgroup wait for kernel arguments that are NOT among the preloaded ones?
// The knot kernels take a ~400-byte argument block by value; the first sixteen dwords arrive in SGPRs (kernarg preload), the rest is read
// with scalar loads from the kernarg segment — which the command processor writes afresh for every launch — and the first
// `s_waitcnt lgkmcnt(0)` of a wave (the counter is shared by LDS and scalar memory, and scalar loads return out of order: any LDS wait is a
// wait for them too) sits out that read.  Per workgroup: cycles from a first time stamp to the arrival of (a) a dword of the by-value
// argument block far behind the preloaded part, (b) a dword of a device buffer read in every launch (hot in the L2).
//   hipcc --offload-arch=gfx950 -O2 -mllvm -amdgpu-kernarg-preload-count=16 -o /tmp/kernarg_probe tools/diag/kernarg_probe.hip && /tmp/kernarg_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
struct Big { long long pad[48]; long long tail; };   // 392 bytes by value: `tail` lies far behind the preloaded dwords
__global__ __launch_bounds__(256) void k(const long long* hot, unsigned long long* out, long long* sink, Big a) {
    unsigned long long t0, t1, t2, t3;
    long long va, vh;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    // (a) the by-value block, through the kernarg segment pointer behind a wall (the compiler hoists by-value loads to the entry)
    const __attribute__((address_space(4))) char* kp = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    asm volatile("s_load_dwordx2 %0, %1, 0x198\n\ts_waitcnt lgkmcnt(0)" : "=s"(va) : "s"(kp) : "memory");   // offset of a.tail: 16 + 48 * 8 + ... (checked on the host)
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    // (b) the hot device buffer, scalar load
    asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(vh) : "s"(hot) : "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
    // (c) the hot buffer, vector load
    long long vv = hot[threadIdx.x & 7];
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t3) : "v"(vv) : "memory");
    if (threadIdx.x == 0) { out[blockIdx.x * 4 + 0] = t1 - t0; out[blockIdx.x * 4 + 1] = t2 - t1; out[blockIdx.x * 4 + 2] = t3 - t2; out[blockIdx.x * 4 + 3] = (unsigned long long)(va == a.tail); }
    if (va + vh + vv == 0x7fffffffffffll) sink[0] = va;
}
int main() {
    const int wgs = 100;
    long long* d_hot; unsigned long long* d_out; long long* d_sink;
    CK(hipMalloc(&d_hot, 4096)); CK(hipMalloc(&d_out, wgs * 4 * 8)); CK(hipMalloc(&d_sink, 8));
    CK(hipMemset(d_hot, 1, 4096));
    Big a{};
    a.tail = 0x1234567;
    std::vector<unsigned long long> o(wgs * 4);
    for (int rep = 0; rep < 6; ++rep) {
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, d_hot, d_out, d_sink, a);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(o.data(), d_out, o.size() * 8, hipMemcpyDeviceToHost));
        std::vector<unsigned long long> ka, sh, vh;
        int okc = 0;
        for (int g = 0; g < wgs; ++g) { ka.push_back(o[g * 4]); sh.push_back(o[g * 4 + 1]); vh.push_back(o[g * 4 + 2]); okc += int(o[g * 4 + 3]); }
        std::sort(ka.begin(), ka.end()); std::sort(sh.begin(), sh.end()); std::sort(vh.begin(), vh.end());
        std::printf("cycles (s_memtime, 100 MHz x ...: shader clock) median / max over %d workgroups: by-value argument behind the preloaded part %llu / %llu   "
                    "hot buffer, scalar load %llu / %llu   hot buffer, vector load %llu / %llu   (argument read correctly in %d workgroups)\n",
                    wgs, ka[wgs / 2], ka[wgs - 1], sh[wgs / 2], sh[wgs - 1], vh[wgs / 2], vh[wgs - 1], okc);
    }
    return 0;
}
