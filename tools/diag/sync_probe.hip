// Diagnostic (never part of the product): what the completion wait costs with the device's scheduling flag (hipDeviceScheduleAuto /
// Spin / Yield / BlockingSync) and with event / stream queries polled by the host, around an 8 us kernel.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void busy(long ticks, int* sink) {
    const long t0 = __builtin_amdgcn_s_memrealtime();
    while (long(__builtin_amdgcn_s_memrealtime()) - t0 < ticks) {}
    if (ticks < 0) *sink = 1;
}
template <class F> double median_us(int reps, F f) {
    std::vector<double> v;
    for (int i = 0; i < 20; ++i) f();
    for (int i = 0; i < reps; ++i) { auto t0 = std::chrono::steady_clock::now(); f(); v.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count()); }
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}
int main(int argc, char** argv) {
    const unsigned flag = argc > 1 ? unsigned(std::atoi(argv[1])) : 0u;
    const char* names[] = {"auto", "spin", "yield", "", "blocking"};
    hipError_t fe = hipSetDeviceFlags(flag);
    std::printf("hipSetDeviceFlags(%s): %s\n", flag <= 4 ? names[flag] : "?", hipGetErrorString(fe));
    CK(hipSetDevice(0));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int* d_sink;
    CK(hipMalloc(&d_sink, 4));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const long T8 = 800;   // 100 MHz ticks: 8 us
    std::printf("  8 us kernel + hipStreamSynchronize          %7.1f us\n", median_us(300, [&] { hipLaunchKernelGGL(busy, dim3(100), dim3(512), 0, s, T8, d_sink); (void)hipStreamSynchronize(s); }));
    std::printf("  8 us kernel + spin on hipStreamQuery        %7.1f us\n", median_us(300, [&] { hipLaunchKernelGGL(busy, dim3(100), dim3(512), 0, s, T8, d_sink); while (hipStreamQuery(s) == hipErrorNotReady) {} }));
    std::printf("  8 us kernel + event + spin on hipEventQuery %7.1f us\n", median_us(300, [&] { hipLaunchKernelGGL(busy, dim3(100), dim3(512), 0, s, T8, d_sink); (void)hipEventRecord(ev, s); while (hipEventQuery(ev) == hipErrorNotReady) {} }));
    std::printf("  20 x 8 us kernels + hipStreamSynchronize    %7.1f us\n", median_us(100, [&] { for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(busy, dim3(100), dim3(512), 0, s, T8, d_sink); (void)hipStreamSynchronize(s); }));
    std::printf("  20 x 8 us kernels + hipDeviceSynchronize    %7.1f us\n", median_us(100, [&] { for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(busy, dim3(100), dim3(512), 0, s, T8, d_sink); (void)hipDeviceSynchronize(); }));
    return 0;
}
