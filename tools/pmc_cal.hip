// FETCH_SIZE / WRITE_SIZE calibration kernels (known byte counts) for the access widths
// sig_kernel uses: 4 B/lane coalesced loads, 16 B/lane loads, 4 B/lane stores.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void cal_read_dword(const uint32_t* p, uint64_t n, uint32_t* out) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) acc ^= p[i];
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void cal_read_x4(const uint4* p, uint64_t n, uint32_t* out) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) { uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void cal_write_dword(uint32_t* p, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}
int main() {
    const uint64_t bytes = 2ull << 30;   // 2 GiB > 256 MiB Infinity Cache
    uint32_t *buf, *out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 4));
    CK(hipMemset(buf, 1, bytes));
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(cal_read_dword, dim3(4096), dim3(256), 0, 0, buf, bytes / 4, out);
    hipLaunchKernelGGL(cal_read_x4, dim3(4096), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, out);
    hipLaunchKernelGGL(cal_write_dword, dim3(4096), dim3(256), 0, 0, buf, bytes / 4);
    CK(hipDeviceSynchronize());
    printf("calibration kernels moved %llu bytes each\n", (unsigned long long)bytes);
    return 0;
}
