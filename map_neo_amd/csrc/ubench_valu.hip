// VALU integer-op throughput calibration for gfx950 (MI355X).
// AMD publishes no int32-op peak, so the signature kernel's roofline is
// priced against a measured rate per instruction class.  Each kernel runs NCHAIN
// independent dependency chains of ONE instruction per lane; rate = lane-ops / second.
// Build: hipcc --offload-arch=gfx950 -O3 -o build/ubench_valu map_neo_amd/csrc/ubench_valu.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int NCHAIN = 8;
constexpr int UNROLL = 32;     // asm statements per chain per loop iteration (256 instructions per iteration: the loop branch costs < 1 %)

#define KERNEL32(NAME, ASM)                                                              \
__global__ void __launch_bounds__(256) k_##NAME(uint32_t* out, int iters, uint32_t s0, uint32_t s1, unsigned long long* cyc) { \
    const unsigned long long c0 = __builtin_readcyclecounter();                          \
    uint32_t acc[NCHAIN];                                                                \
    uint32_t x = threadIdx.x * 2654435761u + s0, y = (threadIdx.x ^ s1) | 1u;            \
    for (int c = 0; c < NCHAIN; c++) acc[c] = x + c * 977u;                              \
    for (int it = 0; it < iters; it++) {                                                 \
        _Pragma("unroll") for (int u = 0; u < UNROLL; u++) {                             \
            _Pragma("unroll") for (int c = 0; c < NCHAIN; c++)                           \
                asm volatile(ASM : "+v"(acc[c]) : "v"(x), "v"(y) : "vcc", "s20", "s21");               \
        }                                                                                \
    }                                                                                    \
    uint32_t r = 0;                                                                      \
    for (int c = 0; c < NCHAIN; c++) r ^= acc[c];                                        \
    if (r == 0x12345678u) out[blockIdx.x * blockDim.x + threadIdx.x] = r;                \
    if (threadIdx.x == 0) atomicMax(cyc, __builtin_readcyclecounter() - c0);             \
}

#define KERNEL64(NAME, ASM)                                                              \
__global__ void __launch_bounds__(256) k_##NAME(uint32_t* out, int iters, uint32_t s0, uint32_t s1, unsigned long long* cyc) { \
    const unsigned long long c0 = __builtin_readcyclecounter();                          \
    uint64_t acc[NCHAIN];                                                                \
    uint32_t x = threadIdx.x * 2654435761u + s0, y = (threadIdx.x ^ s1) | 1u;            \
    for (int c = 0; c < NCHAIN; c++) acc[c] = ((uint64_t)x << 20) + c * 977u;            \
    for (int it = 0; it < iters; it++) {                                                 \
        _Pragma("unroll") for (int u = 0; u < UNROLL; u++) {                             \
            _Pragma("unroll") for (int c = 0; c < NCHAIN; c++)                           \
                asm volatile(ASM : "+v"(acc[c]) : "v"(x), "v"(y) : "vcc", "s20", "s21");               \
        }                                                                                \
    }                                                                                    \
    uint64_t r = 0;                                                                      \
    for (int c = 0; c < NCHAIN; c++) r ^= acc[c];                                        \
    if (r == 0x12345678u) out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)r;      \
    if (threadIdx.x == 0) atomicMax(cyc, __builtin_readcyclecounter() - c0);             \
}

KERNEL32(add_u32,      "v_add_u32 %0, %0, %1")
KERNEL32(xor_b32,      "v_xor_b32 %0, %0, %1")
KERNEL32(lshrrev_b32,  "v_lshrrev_b32 %0, 3, %0")
KERNEL32(alignbit_b32, "v_alignbit_b32 %0, %0, %0, 27")
KERNEL32(bfi_b32,      "v_bfi_b32 %0, %0, %1, %2")
KERNEL32(add3_u32,     "v_add3_u32 %0, %0, %1, %2")
KERNEL32(lshl_add_u32, "v_lshl_add_u32 %0, %0, 3, %1")
KERNEL32(xad_u32,      "v_xad_u32 %0, %0, %1, %2")
KERNEL32(and_or_b32,   "v_and_or_b32 %0, %0, %1, %2")
KERNEL32(perm_b32,     "v_perm_b32 %0, %0, %1, %2")
KERNEL32(min_u32,      "v_min_u32 %0, %0, %1")
KERNEL32(min3_u32,     "v_min3_u32 %0, %0, %1, %2")
KERNEL32(max3_u32,     "v_max3_u32 %0, %0, %1, %2")
KERNEL32(mul_lo_u32,   "v_mul_lo_u32 %0, %0, %1")
KERNEL32(mul_hi_u32,   "v_mul_hi_u32 %0, %0, %1")
KERNEL32(mul_u32_u24,  "v_mul_u32_u24 %0, %0, %1")
KERNEL32(mad_u32_u24,  "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL32(add_co_u32,   "v_add_co_u32 %0, vcc, %0, %1")
KERNEL32(cndmask_b32,  "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(fma_f32,      "v_fma_f32 %0, %0, %1, %2")
KERNEL32(and_b32,      "v_and_b32 %0, %0, %1")
KERNEL32(or_b32,       "v_or_b32 %0, %0, %1")
KERNEL32(sub_u32,      "v_sub_u32 %0, %0, %1")
KERNEL32(lshlrev_b32,  "v_lshlrev_b32 %0, 3, %0")
KERNEL32(max_u32,      "v_max_u32 %0, %0, %1")
KERNEL32(add_u32_e64,  "v_add_u32_e64 %0, %0, %1")
KERNEL32(xor_e64,      "v_xor_b32_e64 %0, %0, %1")
KERNEL32(min_u32_e64,  "v_min_u32_e64 %0, %0, %1")
KERNEL32(bfe_u32,      "v_bfe_u32 %0, %0, 3, 20")
KERNEL32(or3_b32,      "v_or3_b32 %0, %0, %1, %2")
KERNEL32(cmp_cnd,      "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
KERNEL32(pk_add_u16,   "v_pk_add_u16 %0, %0, %1")
KERNEL32(add_u16,      "v_add_u16 %0, %0, %1")
KERNEL32(sad_u32,      "v_sad_u32 %0, %0, %1, %2")
KERNEL32(mov_dpp,      "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL32(bitop3_b32,   "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
KERNEL32(add3_sgpr,    "v_add3_u32 %0, %0, %1, s20")
KERNEL32(xad_sgpr,     "v_xad_u32 %0, %0, %1, s20")
KERNEL32(add_lit,      "v_add_u32 %0, 0x5A827999, %0")
KERNEL32(add_sgpr,     "v_add_u32 %0, s20, %0")
KERNEL32(alignbit_2r,  "v_alignbit_b32 %0, %0, %1, 27")
KERNEL32(mov_b32,      "v_mov_b32 %0, %1")
KERNEL32(not_b32,      "v_not_b32 %0, %0")
KERNEL32(mix_xor_align, "v_xor_b32 %0, %0, %1\n v_alignbit_b32 %0, %0, %0, 27")
KERNEL32(mix_xor_xor_align, "v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %2\n v_alignbit_b32 %0, %0, %0, 27")
KERNEL32(mix_add_add3, "v_add_u32 %0, %0, %1\n v_add3_u32 %0, %0, %1, %2")
KERNEL64(mad_u64_u32,  "v_mad_u64_u32 %0, vcc, %1, %2, %0")
KERNEL64(fma_f64,      "v_fma_f64 %0, %0, %0, %0")
KERNEL64(add_f64,      "v_add_f64 %0, %0, %0")
KERNEL64(lshlrev_b64,  "v_lshlrev_b64 %0, 1, %0")

// round 2: the remaining opcodes / operand forms of sig_kernel's hot regions (tools/issue_model.py)
KERNEL32(cmp_le_e64,    "v_cmp_le_u32_e64 s[20:21], %0, %1")
KERNEL32(cmp_le_e32,    "v_cmp_le_u32_e32 vcc, %0, %1")
KERNEL32(readlane,      "v_readlane_b32 s20, %0, 3")
KERNEL32(readfirstlane, "v_readfirstlane_b32 s20, %0")
KERNEL32(writelane,     "v_writelane_b32 %0, s20, 3")
KERNEL32(add_e64_clamp, "v_add_u32_e64 %0, %0, 6 clamp")
KERNEL32(alignbyte_b32, "v_alignbyte_b32 %0, %0, %1, 1")
KERNEL32(cndmask_e64,   "v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
KERNEL32(bitop3_sgpr,   "v_bitop3_b32 %0, %0, %1, s20 bitop3:0x96")
KERNEL32(and_lit,       "v_and_b32 %0, 0x1FFFFFFF, %0")
KERNEL32(perm_sgpr,     "v_perm_b32 %0, %0, %1, s20")
KERNEL32(mbcnt_lo,      "v_mbcnt_lo_u32_b32 %0, -1, %0")
KERNEL32(min3_u32_b,    "v_min3_u32 %0, %0, %1, %2")
KERNEL32(xor_sgpr,      "v_xor_b32 %0, s20, %0")
KERNEL32(mul_lo_sgpr,   "v_mul_lo_u32 %0, %0, s20")
KERNEL64(mad_u64_sdst,  "v_mad_u64_u32 %0, s[20:21], %1, %2, %0")
KERNEL64(lshl_add_u64,  "v_lshl_add_u64 %0, %0, 0, %0")
KERNEL64(mov_b64,       "v_mov_b64 %0, %0")

typedef void (*kern_t)(uint32_t*, int, uint32_t, uint32_t, unsigned long long*);
struct Entry { const char* name; kern_t fn; };
#define E(NAME) { #NAME, k_##NAME }
static Entry entries[] = {
    E(add_u32), E(xor_b32), E(lshrrev_b32), E(alignbit_b32), E(bfi_b32), E(add3_u32), E(lshl_add_u32),
    E(xad_u32), E(and_or_b32), E(perm_b32), E(min_u32), E(min3_u32), E(max3_u32), E(mul_lo_u32),
    E(mul_hi_u32), E(mul_u32_u24), E(mad_u32_u24), E(add_co_u32), E(cndmask_b32), E(fma_f32),
    E(and_b32), E(or_b32), E(sub_u32), E(lshlrev_b32), E(max_u32), E(add_u32_e64), E(xor_e64), E(min_u32_e64), E(bfe_u32), E(or3_b32), E(cmp_cnd), E(pk_add_u16), E(add_u16), E(sad_u32), E(mov_dpp),
    E(bitop3_b32), E(add3_sgpr), E(xad_sgpr), E(add_lit), E(add_sgpr), E(alignbit_2r), E(mov_b32), E(not_b32),
    E(mix_xor_align), E(mix_xor_xor_align), E(mix_add_add3),
    E(mad_u64_u32), E(fma_f64), E(add_f64), E(lshlrev_b64),
    E(cmp_le_e64), E(cmp_le_e32), E(readlane), E(readfirstlane), E(writelane), E(add_e64_clamp), E(alignbyte_b32), E(cndmask_e64),
    E(bitop3_sgpr), E(and_lit), E(perm_sgpr), E(mbcnt_lo), E(min3_u32_b), E(xor_sgpr), E(mul_lo_sgpr), E(mad_u64_sdst),
    E(lshl_add_u64), E(mov_b64),
};

int main(int argc, char** argv) {
    int iters = 2000, blocks_per_cu = 8;
    const char* only = nullptr;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--iters") && i + 1 < argc) iters = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--bpc") && i + 1 < argc) blocks_per_cu = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--only") && i + 1 < argc) only = argv[++i];
    }
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    int blocks = cus * blocks_per_cu;
    uint32_t* out;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    // shader cycles (s_memtime) the longest-lived wave spent in the kernel: with blocks_per_cu = 8 every SIMD holds 8
    // co-resident waves from start to end, so cycles / (instructions per wave x 8) = SIMD cycles per instruction,
    // independent of the clock the chip happened to run at
    unsigned long long* cyc;
    CHECK(hipMalloc(&cyc, 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"blocks\": %d, \"iters\": %d, \"ops\": {\n",
           prop.gcnArchName, cus, prop.clockRate / 1000, blocks, iters);
    bool first = true;
    for (auto& e : entries) {
        if (only && strcmp(only, e.name)) continue;
        hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, 64, 1u, 2u, cyc);   // warm-up
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        unsigned long long best_cyc = ~0ull;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipMemset(cyc, 0, 8));
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, iters, 1u, 2u, cyc);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
            unsigned long long c; CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
            if (c < best_cyc) best_cyc = c;
        }
        double waves_per_simd = blocks_per_cu * 4 / 4.0;
        double cyc_per_inst = (double)best_cyc / ((double)iters * UNROLL * NCHAIN * waves_per_simd);
        double laneops = (double)blocks * 256 * (double)iters * UNROLL * NCHAIN;
        double tops = laneops / (best * 1e-3) / 1e12;
        // lane-ops per clock per CU at the nominal clock
        double per_clk_cu = laneops / (best * 1e-3) / ((double)prop.clockRate * 1e3) / cus;
        printf("%s  \"%s\": {\"ms\": %.3f, \"Tops\": %.2f, \"lanes_per_clk_per_cu\": %.1f, \"simd_cycles_per_op\": %.3f, \"mhz\": %.0f}",
               first ? "" : ",\n", e.name, best, tops, per_clk_cu, cyc_per_inst, (double)best_cyc / (best * 1e3));
        first = false;
        fflush(stdout);
    }
    printf("\n}}\n");
    return 0;
}
