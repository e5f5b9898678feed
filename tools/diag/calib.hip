// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE for THIS engine's access shape: 8 bytes per lane, coalesced
// (guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE is calibrated only for 16-B-per-lane streams; other widths must be
// calibrated on a known byte count).  Streams a buffer larger than the 256 MiB Infinity Cache.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void copy8(const double* __restrict__ in, double* __restrict__ out, size_t n) {
    size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
    const size_t stride = size_t(gridDim.x) * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i] + 1.0;
}
int main() {
    const size_t n = size_t(1) << 27;  // 128 Mi doubles = 1 GiB read + 1 GiB written per launch
    double *a, *b;
    hipMalloc(&a, n * 8); hipMalloc(&b, n * 8);
    hipMemset(a, 0, n * 8);
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(copy8, dim3(4096), dim3(256), 0, 0, a, b, n);
    hipDeviceSynchronize();
    printf("calib: %zu bytes read and %zu bytes written per launch\n", n * 8, n * 8);
    return 0;
}
