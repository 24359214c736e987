// Development probe for mdd_signatures_from_host: resident kernel vs streamed single launch vs a resident kernel next
// to plain / callback-punctuated chunked copies.  Includes the library source like tools/sig_bench.hip.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o build/stream_bench tools/stream_bench.hip
//   ./build/stream_bench [n_docs] [chunk_mib]
#include "../map_neo_amd/csrc/mdd_hip.hip"
// mdd_ctx_destroy releases the context's communicator; this single-file tool has none and does not link mdd_comm.hip
extern "C" int mdd_comm_destroy(mdd_ctx*) { return 0; }

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    uint64_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 1000000;
    uint64_t chunk = (argc > 2 ? strtoull(argv[2], 0, 10) : 64) << 20;
    const int K = 128;
    std::mt19937_64 rng(12345);
    std::lognormal_distribution<double> ln(std::log(2048.0) - 0.18, 0.6);
    std::vector<uint64_t> offs(n + 1, 0);
    for (uint64_t d = 0; d < n; d++) offs[d + 1] = offs[d] + (uint64_t)std::max(1.0, ln(rng));
    uint64_t total = offs[n], cap = ((total + 3) / 4) * 4 + MDD_TEXT_PAD;
    uint8_t* host;
    CK(hipHostMalloc(reinterpret_cast<void**>(&host), cap, hipHostMallocDefault));
    uint64_t x = 88172645463325252ull;
    for (uint64_t i = 0; i < total; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; uint32_t r = (uint32_t)(x >> 33) % 32; host[i] = r < 26 ? 'a' + r : ' '; }
    std::vector<uint64_t> a(K), b(K);
    mdd_make_perms(1, K, a.data(), b.data());
    uint8_t *dt, *dt2; uint64_t *doff, *da, *db; uint32_t *dsig, *dsig2;
    CK(hipMalloc(&dt, cap)); CK(hipMalloc(&dt2, cap)); CK(hipMalloc(&doff, (n + 1) * 8)); CK(hipMalloc(&da, K * 8)); CK(hipMalloc(&db, K * 8));
    CK(hipMalloc(&dsig, n * K * 4)); CK(hipMalloc(&dsig2, n * K * 4));
    CK(hipMemcpy(dt, host, cap, hipMemcpyHostToDevice));
    CK(hipMemcpy(doff, offs.data(), (n + 1) * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(da, a.data(), K * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), K * 8, hipMemcpyHostToDevice));
    hipStream_t s0, cs;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    mdd_ctx* ctx;
    if (mdd_ctx_create(0, s0, &ctx)) { fprintf(stderr, "%s\n", mdd_last_error()); return 1; }
    auto resident = [&]() { if (mdd_signatures(ctx, dt, total, doff, n, 0, 13, K, da, db, dsig)) { fprintf(stderr, "%s\n", mdd_last_error()); exit(1); } };
    auto copies = [&](bool callbacks) {
        static volatile uint64_t sink;
        for (uint64_t lo = 0; lo < total; lo += chunk) {
            uint64_t nb = std::min(chunk, total - lo);
            CK(hipMemcpyAsync(dt2 + lo, host + lo, nb, hipMemcpyHostToDevice, cs));
            if (callbacks) CK(hipLaunchHostFunc(cs, [](void* p) { sink = (uint64_t)p; }, (void*)lo));
        }
    };
    resident(); CK(hipDeviceSynchronize());
    double t;
    for (int rep = 0; rep < 2; rep++) { t = now_ms(); resident(); CK(hipDeviceSynchronize()); printf("resident              %.2f ms\n", now_ms() - t); }
    for (int rep = 0; rep < 2; rep++) { t = now_ms(); copies(false); CK(hipDeviceSynchronize()); printf("copies alone          %.2f ms\n", now_ms() - t); }
    for (int rep = 0; rep < 2; rep++) { t = now_ms(); copies(true); CK(hipDeviceSynchronize()); printf("copies+callbacks alone %.2f ms\n", now_ms() - t); }
    for (int rep = 0; rep < 2; rep++) {
        t = now_ms(); resident(); copies(false); CK(hipStreamSynchronize(cs)); double tc = now_ms() - t; CK(hipDeviceSynchronize());
        printf("resident || copies    copies %.2f ms, all %.2f ms\n", tc, now_ms() - t);
    }
    for (int rep = 0; rep < 2; rep++) {
        t = now_ms(); resident(); copies(true); CK(hipStreamSynchronize(cs)); double tc = now_ms() - t; CK(hipDeviceSynchronize());
        printf("resident || copies+cb copies %.2f ms, all %.2f ms\n", tc, now_ms() - t);
    }
    for (int rep = 0; rep < 3; rep++) {
        t = now_ms();
        if (mdd_signatures_from_host(ctx, cs, host, total, dt2, doff, n, 0, 13, K, da, db, dsig2, chunk)) { fprintf(stderr, "%s\n", mdd_last_error()); return 1; }
        CK(hipStreamSynchronize(cs)); double tc = now_ms() - t;
        if (mdd_ctx_sync(ctx)) { fprintf(stderr, "%s\n", mdd_last_error()); return 1; }
        double ta = now_ms() - t;
        uint64_t ticks = 0; mdd_stream_wait_ticks(ctx, &ticks);
        printf("streamed              copies %.2f ms, all %.2f ms, waited %.1f wave-ms\n", tc, ta, ticks / 1e5);
    }
    // variants of the streaming protocol, driven from here with the library's internals:
    //   dst = dt2 (kernel reads what is being copied) or dt3 (kernel reads pre-filled dt2, copies land elsewhere)
    uint8_t* dt3; CK(hipMalloc(&dt3, cap));
    volatile uint64_t* flag; void* hp; CK(hipHostMalloc(&hp, 64, hipHostMallocDefault)); flag = (volatile uint64_t*)hp; *flag = 0;
    uint64_t* flag_dev; CK(hipHostGetDevicePointer((void**)&flag_dev, hp, 0));
    static ReadyNote notes[8192];
    uint64_t base = 0;
    for (int variant = 0; variant < 4; variant++) {
        uint8_t* dst = (variant & 1) ? dt3 : dt2;
        bool instant = variant & 2;      // flag already at the end before the kernel starts: the kernel never waits
        for (int rep = 0; rep < 2; rep++) {
            t = now_ms();
            StreamIn in{flag_dev, ctx->stream_words, base};
            CK(hipMemsetAsync(ctx->stream_words + 1, 0, 8, s0)); CK(hipMemsetAsync(ctx->stream_words + 3, 0, 8, s0));
            if (instant) *flag = base + total;
            if (signatures_impl("probe", ctx, dt2, total, doff, n, 0, 13, K, da, db, dsig2, ctx->seg_items, ctx->seg_n_items, SEG_ITEM_CAP, true, &in)) { fprintf(stderr, "%s\n", mdd_last_error()); return 1; }
            uint64_t done = 0, next = 2ull << 20; int k = 0;
            while (done < total) {
                uint64_t nb = std::min(next, total - done);
                CK(hipMemcpyAsync(dst + done, host + done, nb, hipMemcpyHostToDevice, cs));
                done += nb;
                notes[k] = ReadyNote{flag, base + done};
                if (!instant) CK(hipLaunchHostFunc(cs, [](void* p) { ReadyNote* q = (ReadyNote*)p; *q->flag = q->value; }, &notes[k]));
                k++;
                if (next < chunk) next = std::min(next * 2, chunk);
            }
            base += total;
            CK(hipStreamSynchronize(cs)); double tc = now_ms() - t;
            CK(hipStreamSynchronize(s0)); double ta = now_ms() - t;
            uint64_t ticks = 0; mdd_stream_wait_ticks(ctx, &ticks);
            printf("protocol dst=%s %s  copies %.2f ms, all %.2f ms, waited %.1f wave-ms\n", (variant & 1) ? "other" : "input", instant ? "flag-preset" : "flag-live  ", tc, ta, ticks / 1e5);
        }
    }
    std::vector<uint32_t> s1(n * K), s2(n * K);
    CK(hipMemcpy(s1.data(), dsig, n * K * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(s2.data(), dsig2, n * K * 4, hipMemcpyDeviceToHost));
    printf("equal %d\n", (int)(s1 == s2));
    return 0;
}
