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
// scattered reads: every lane loads 16 (4) bytes from its own pseudo-random 64-byte line of the 2 GiB buffer -- the access
// shape of leaf_kernel's byte comparisons and of the hash-table kernels.  Nominal bytes = 64 per request (one line).
__global__ void cal_read_scatter16(const uint4* p, uint64_t n_lines, uint64_t n_req, uint32_t* out) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_req; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t x = (i + 1) * 0x9E3779B97F4A7C15ull;
        x ^= x >> 29;
        x *= 0xBF58476D1CE4E5B9ull;
        x ^= x >> 32;
        uint4 v = p[(x % n_lines) * 4];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void cal_read_scatter4(const uint32_t* p, uint64_t n_lines, uint64_t n_req, uint32_t* out) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_req; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t x = (i + 1) * 0x9E3779B97F4A7C15ull;
        x ^= x >> 29;
        x *= 0xBF58476D1CE4E5B9ull;
        x ^= x >> 32;
        acc ^= p[(x % n_lines) * 16];
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// scattered one-byte stores (leaf_kernel's marks): nominal bytes = 1 per request
__global__ void cal_write_scatter1(uint8_t* p, uint64_t n_bytes, uint64_t n_req) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_req; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t x = (i + 1) * 0x9E3779B97F4A7C15ull;
        x ^= x >> 29;
        x *= 0xBF58476D1CE4E5B9ull;
        x ^= x >> 32;
        p[x % n_bytes] = 1;
    }
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
    const uint64_t n_req = 1ull << 25;  // 32 Mi scattered requests (2 GiB nominal at 64 B per line)
    hipLaunchKernelGGL(cal_read_scatter16, dim3(4096), dim3(256), 0, 0, (const uint4*)buf, bytes / 64, n_req, out);
    hipLaunchKernelGGL(cal_read_scatter4, dim3(4096), dim3(256), 0, 0, (const uint32_t*)buf, bytes / 64, n_req, out);
    hipLaunchKernelGGL(cal_write_scatter1, dim3(4096), dim3(256), 0, 0, (uint8_t*)buf, bytes, n_req);
    CK(hipDeviceSynchronize());
    printf("calibration kernels moved %llu bytes each\n", (unsigned long long)bytes);
    return 0;
}
