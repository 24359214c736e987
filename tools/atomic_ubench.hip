// Random 8-byte atomicCAS / plain-load throughput against table size (development tool).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o build/atomic_ubench tools/atomic_ubench.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ uint64_t mix64(uint64_t x) { x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL; x ^= x >> 27; x *= 0x94d049bb133111ebULL; x ^= x >> 31; return x; }
__global__ void cas_kernel(unsigned long long* tab, uint64_t mask, uint64_t n, uint64_t salt) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t k = mix64(i + salt);
    atomicCAS(&tab[k & mask], ~0ULL, k | 1);
}
__global__ void load_kernel(const unsigned long long* tab, uint64_t mask, uint64_t n, uint64_t salt, unsigned long long* sink) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t k = mix64(i + salt);
    if (tab[k & mask] == 12345) *sink = i;
}
int main() {
    const uint64_t n = 1ull << 28;
    unsigned long long *tab, *sink;
    CK(hipMalloc(&tab, 8ull << 31));
    CK(hipMalloc(&sink, 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int lg = 20; lg <= 31; lg++) {  // table entries 2^lg -> 8 MB .. 16 GB
        uint64_t size = 1ull << lg;
        CK(hipMemset(tab, 0xFF, size * 8));
        float ms[2];
        for (int which = 0; which < 2; which++) {
            CK(hipEventRecord(e0));
            if (which == 0) hipLaunchKernelGGL(cas_kernel, dim3(n / 256), dim3(256), 0, 0, tab, size - 1, n, (uint64_t)lg << 40);
            else hipLaunchKernelGGL(load_kernel, dim3(n / 256), dim3(256), 0, 0, tab, size - 1, n, (uint64_t)lg << 40, sink);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms[which], e0, e1));
        }
        printf("table %8.0f MB  cas %.2f ms = %6.2f G/s   load %.2f ms = %6.2f G/s\n", size * 8 / 1048576.0, ms[0], n / ms[0] / 1e6, ms[1], n / ms[1] / 1e6);
    }
    return 0;
}
