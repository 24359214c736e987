// Standalone timing harness for sig_kernel (development tool; bench.py is the judged bench).
// Includes the library source directly so kernel variants can be toggled with -D macros.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o build/sig_bench tools/sig_bench.hip
//   ./build/sig_bench [n_docs] [mean_bytes] [sigma_x100] [K] [lang: 0 ascii, 1 cjk] [unit] [width] [seg_bytes] [filter_from]
// MDD_CORPUS_FILE=path: hash that corpus instead of the built-in random text (u64 n, u64 text_bytes, u64 offsets[n+1],
// text bytes -- written by tools/dump_corpus.py; n_docs / mean / sigma / lang arguments are then ignored).
// Built with -DMDD_PATHCOUNT (build/sig_bench_count) the kernel also counts how often each MDD_REGION is entered
// (one launch) and the tool prints them as JSON: the dynamic side of tools/issue_model.py.
#include "../map_neo_amd/csrc/mdd_hip.hip"
// mdd_ctx_destroy releases the context's communicator; this single-file tool has none and does not link mdd_comm.hip
extern "C" int mdd_comm_destroy(mdd_ctx*) { return 0; }

#include <cmath>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    uint64_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 200000;
    double mean = argc > 2 ? atof(argv[2]) : 2048;
    double sigma = argc > 3 ? atof(argv[3]) / 100.0 : 0.6;
    int K = argc > 4 ? atoi(argv[4]) : 128;
    int lang = argc > 5 ? atoi(argv[5]) : 0;
    int unit = argc > 6 ? atoi(argv[6]) : 0;
    int width = argc > 7 ? atoi(argv[7]) : 13;
    uint32_t seg_bytes = argc > 8 ? (uint32_t)strtoul(argv[8], 0, 10) : 0;
    uint32_t filter_from = argc > 9 ? (uint32_t)strtoul(argv[9], 0, 10) : 0;
    std::mt19937_64 rng(12345);
    std::lognormal_distribution<double> ln(std::log(mean) - 0.5 * sigma * sigma, sigma);
    std::vector<uint64_t> offs(n + 1, 0);
    for (uint64_t d = 0; d < n; d++) {
        uint64_t len = sigma > 0 ? (uint64_t)std::max(1.0, ln(rng)) : (uint64_t)mean;
        if (lang == 1) len = (len / 3) * 3;
        offs[d + 1] = offs[d] + len;
    }
    const char* cfile = getenv("MDD_CORPUS_FILE");
    FILE* cf = nullptr;
    if (cfile) {
        cf = fopen(cfile, "rb");
        uint64_t hdr[2];
        if (!cf || fread(hdr, 8, 2, cf) != 2) { fprintf(stderr, "cannot read %s\n", cfile); return 1; }
        n = hdr[0];
        offs.assign(n + 1, 0);
        if (fread(offs.data(), 8, n + 1, cf) != n + 1 || offs[n] != hdr[1]) { fprintf(stderr, "bad corpus file\n"); return 1; }
    }
    uint64_t total = offs[n];
    std::vector<uint8_t> text(((total + 3) / 4) * 4 + MDD_TEXT_PAD, 0);
    uint64_t x = 88172645463325252ull;
    if (cf) {
        if (fread(text.data(), 1, total, cf) != total) { fprintf(stderr, "short corpus file\n"); return 1; }
        fclose(cf);
        lang = 0;
        for (uint64_t i = 0; i < total; i++) if (text[i] & 0x80) { lang = 2; break; }
    } else if (lang == 0) {
        for (uint64_t i = 0; i < total; i++) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            uint32_t r = (uint32_t)(x >> 33) % 32;
            text[i] = r < 26 ? 'a' + r : ' ';
        }
    } else {
        for (uint64_t i = 0; i + 2 < total + 2 && i < total; i += 3) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            uint32_t cp = 0x4E00 + (uint32_t)(x >> 33) % 6000;
            text[i] = 0xE0 | (cp >> 12); if (i + 1 < total) text[i + 1] = 0x80 | ((cp >> 6) & 0x3F); if (i + 2 < total) text[i + 2] = 0x80 | (cp & 0x3F);
        }
    }
    uint64_t nsh = 0;
    for (uint64_t d = 0; d < n; d++) {
        uint64_t len = offs[d + 1] - offs[d];
        uint64_t cp = lang == 0 ? len : len / 3;
        if (lang == 2) { cp = 0; for (uint64_t i = offs[d]; i < offs[d + 1]; i++) cp += (text[i] & 0xC0) != 0x80; }
        if (unit == 1) {  // word unit: tokens = maximal runs of non-space bytes
            cp = 0;
            for (uint64_t i = offs[d]; i < offs[d + 1]; i++)
                if (text[i] != ' ' && (i == offs[d] || text[i - 1] == ' ')) cp++;
        }
        if (cp >= (uint64_t)width) nsh += cp - (uint64_t)width + 1;
    }
    std::vector<uint64_t> a(K), b(K);
    mdd_make_perms(1, K, a.data(), b.data());
    uint8_t* dt; uint64_t *doff, *da, *db; uint32_t* dsig;
    CK(hipMalloc(&dt, text.size())); CK(hipMalloc(&doff, (n + 1) * 8)); CK(hipMalloc(&da, K * 8)); CK(hipMalloc(&db, K * 8));
    CK(hipMalloc(&dsig, n * K * 4));
    CK(hipMemcpy(dt, text.data(), text.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(doff, offs.data(), (n + 1) * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(da, a.data(), K * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), K * 8, hipMemcpyHostToDevice));
    mdd_ctx* ctx;
    if (mdd_ctx_create(0, nullptr, &ctx)) { fprintf(stderr, "%s\n", mdd_last_error()); return 1; }
    if (mdd_ctx_set_segment_bytes(ctx, seg_bytes) || mdd_ctx_set_filter_from(ctx, filter_from)) { fprintf(stderr, "%s\n", mdd_last_error()); return 1; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
#ifdef MDD_PATHCOUNT
    {
        static const char* names[PR_COUNT] = {"item", "round", "scan", "sha_a13", "sha4", "sha10", "sha14", "sha_multi", "stage", "full",
                                              "fsetup", "fgroup", "fdetail", "ffull", "exact", "item_end", "asciichk", "ring"};
        unsigned long long zero[PR_COUNT] = {}, cnt[PR_COUNT];
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_pathcnt), zero, sizeof zero));
        if (mdd_signatures(ctx, dt, total, doff, n, unit, width, K, da, db, dsig)) { fprintf(stderr, "%s\n", mdd_last_error()); return 1; }
        CK(hipDeviceSynchronize());
        CK(hipMemcpyFromSymbol(cnt, HIP_SYMBOL(g_pathcnt), sizeof cnt));
        uint64_t waves = (uint64_t)ctx->cus * 32;
        printf("{\"n_docs\": %llu, \"text_bytes\": %llu, \"shingles\": %llu, \"K\": %d, \"unit\": %d, \"width\": %d, \"waves\": %llu, \"regions\": {",
               (unsigned long long)n, (unsigned long long)total, (unsigned long long)nsh, K, unit, width, (unsigned long long)waves);
        for (int i = 0; i < PR_COUNT; i++) printf("%s\"%s\": %llu", i ? ", " : "", names[i], cnt[i]);
        printf("}}\n");
        return 0;
    }
#endif
    for (int rep = 0; rep < 4; rep++) {
        CK(hipEventRecord(e0, 0));
        if (mdd_signatures(ctx, dt, total, doff, n, unit, width, K, da, db, dsig)) { fprintf(stderr, "%s\n", mdd_last_error()); return 1; }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    std::vector<uint32_t> sig(n * K);
    CK(hipMemcpy(sig.data(), dsig, n * K * 4, hipMemcpyDeviceToHost));
    uint64_t cks = 0;
    for (uint64_t i = 0; i < n * K; i++) cks = cks * 1099511628211ull + sig[i];
    printf("unit=%d width=%d ", unit, width);
    printf("n=%llu bytes=%llu shingles=%llu K=%d sigma=%.2f lang=%d  ms=%.3f  Gsh/s=%.2f  Mdocs/s=%.2f  cks=%016llx\n",
           (unsigned long long)n, (unsigned long long)total, (unsigned long long)nsh, K, sigma, lang, best, nsh / best / 1e6,
           n / best / 1e3, (unsigned long long)cks);
    return 0;
}
