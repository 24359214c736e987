// Checks the lane mapping of the two DPP wave shifts the boundary scan of mdd_shingle.h relies on (gfx950):
//   wave_shr:1 (0x138): lane l reads lane l-1, lane 0 keeps `old`;  wave_shl:1 (0x130): lane l reads lane l+1, lane 63 keeps `old`.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o build/dpp_probe tools/dpp_probe.hip && ./build/dpp_probe   (exit 0 = as assumed)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(int* out) {
    int lane = (int)threadIdx.x;
    out[lane] = __builtin_amdgcn_update_dpp(1000 + lane, lane, 0x138, 0xf, 0xf, false);
    out[64 + lane] = __builtin_amdgcn_update_dpp(1000 + lane, lane, 0x130, 0xf, 0xf, false);
}
int main() {
    int* d = nullptr;
    int h[128];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 2;
    probe<<<1, 64>>>(d);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    int bad = 0;
    for (int l = 0; l < 64; l++) {
        int e0 = l ? l - 1 : 1000, e1 = l < 63 ? l + 1 : 1063;
        if (h[l] != e0 || h[64 + l] != e1) bad++;
    }
    printf("wave_shr:1 lane0=%d lane1=%d lane63=%d | wave_shl:1 lane0=%d lane62=%d lane63=%d | mismatches=%d\n", h[0], h[1], h[63], h[64],
           h[64 + 62], h[64 + 63], bad);
    return bad ? 1 : 0;
}
