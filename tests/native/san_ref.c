/* AddressSanitizer / UBSan driver for the plain-C oracle (oracle/mdd_ref.c, compiled into this binary with
 * -fsanitize=address,undefined by `make -C oracle asan`): signatures, shingle hashes and clustering on a corpus file of
 *   u64 n, u64 text_bytes, u64 offsets[n+1], text
 * (arbitrary bytes: valid and invalid utf-8, empty documents).  Exit 0 = no sanitizer report. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

void ref_make_perms(uint32_t seed, int K, uint64_t* a, uint64_t* b);
uint64_t ref_signatures(const uint8_t* text, const uint64_t* offsets, uint64_t n_docs, int unit, int width, int K, const uint64_t* a,
                        const uint64_t* b, uint32_t* sig);
uint64_t ref_shingle_hashes(const uint8_t* text, uint64_t len, int unit, int width, uint32_t* out, uint64_t cap);
uint64_t ref_cluster(const uint32_t* sig, uint64_t n, int K, int bands, int rows, uint32_t* labels, uint64_t* bucket_of);

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "rb");
    uint64_t hdr[2];
    if (!f || fread(hdr, 8, 2, f) != 2) return 2;
    uint64_t n = hdr[0], tb = hdr[1];
    uint64_t* offs = malloc((n + 1) * 8);
    uint8_t* text = malloc(tb ? tb : 1);      /* EXACT size: any read past the text is an ASan report */
    if (fread(offs, 8, n + 1, f) != n + 1 || fread(text, 1, tb, f) != tb) return 2;
    fclose(f);
    uint64_t sink = 0;
    const int cfg[5][3] = {{0, 13, 128}, {0, 1, 64}, {0, 40, 256}, {1, 5, 128}, {1, 1, 100}};
    for (int c = 0; c < 5; c++) {
        int unit = cfg[c][0], width = cfg[c][1], K = cfg[c][2];
        uint64_t *a = malloc(K * 8), *b = malloc(K * 8);
        ref_make_perms(1, K, a, b);
        uint32_t* sig = malloc((n ? n : 1) * (size_t)K * 4);
        sink += ref_signatures(text, offs, n, unit, width, K, a, b, sig);
        for (uint64_t d = 0; d < n && d < 50; d++) {
            uint64_t len = offs[d + 1] - offs[d];
            uint32_t* out = malloc((len ? len : 1) * 4);
            sink += ref_shingle_hashes(text + offs[d], len, unit, width, out, len ? len : 1);
            sink += ref_shingle_hashes(text + offs[d], len, unit, width, out, 1);      /* capacity smaller than the result */
            free(out);
        }
        if (K >= 117) {
            uint32_t* labels = malloc((n ? n : 1) * 4);
            uint64_t* bucket = malloc((n ? n : 1) * 9 * 8);
            sink += ref_cluster(sig, n, K, 9, 13, labels, bucket);
            free(labels);
            free(bucket);
        }
        free(sig);
        free(a);
        free(b);
    }
    free(offs);
    free(text);
    printf("san_ref ok (%llu)\n", (unsigned long long)(sink & 0xFF));
    return 0;
}
