// AddressSanitizer / UBSan driver for the native JSON-lines scanner and writer (map_neo_amd/csrc/mdd_io.cpp):
// the two pieces of plain C++ on the path that parse untrusted input ("sanitizers").  Built by
// `make -C map_neo_amd/csrc asan` together with mdd_io.cpp itself, both with -fsanitize=address,undefined.
//   san_io <shard> [<shard> ...]   reads every shard with both normalisation modes, touches every output array,
//                                  writes the documents back with mdd_io_write_jsonl (plain and gzip) into <shard>.out*
// Exit code 0 = no sanitizer report (ASan / UBSan abort the process otherwise: -fno-sanitize-recover).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mdd_io.h"

static uint64_t touch(const uint8_t* p, uint64_t n) {
    uint64_t s = 0;
    for (uint64_t i = 0; i < n; i++) s += p[i];
    return s;
}

int main(int argc, char** argv) {
    uint64_t sink = 0, n_ok = 0, n_fb = 0, n_files = 0;
    for (int a = 1; a < argc; a++) {
        for (int norm = 0; norm < 2; norm++) {
            for (int with_id = 0; with_id < 2; with_id++) {
                mdd_io_shard* s = nullptr;
                int rc = mdd_io_read_jsonl(argv[a], "text", with_id ? "doc_id" : nullptr, norm, &s);
                if (rc != 0) {  // a refused file (e.g. a truncated gzip member) is a clean error, not a crash
                    sink += strlen(mdd_io_last_error());
                    continue;
                }
                n_files++;
                uint64_t nl = mdd_io_n_lines(s), nd = mdd_io_n_docs(s), rb = mdd_io_raw_bytes(s);
                const uint8_t* st = mdd_io_status(s);
                const uint64_t *lb = mdd_io_line_begin(s), *le = mdd_io_line_end(s), *ib = mdd_io_id_begin(s), *ie = mdd_io_id_end(s);
                const uint8_t* ik = mdd_io_id_kind(s);
                const uint8_t* raw = mdd_io_raw(s);
                for (uint64_t i = 0; i < nl; i++) {
                    n_ok += st[i] == MDD_IO_OK;
                    n_fb += st[i] == MDD_IO_FALLBACK;
                    if (le[i] > rb || lb[i] > le[i]) return 3;
                    sink += touch(raw + lb[i], le[i] - lb[i]) + mdd_io_line_no(s)[i];
                    if (ik[i] != MDD_IO_ID_ABSENT) {
                        if (ie[i] > rb || ib[i] > ie[i]) return 4;
                        sink += touch(raw + ib[i], ie[i] - ib[i]);
                    }
                }
                const uint64_t* off = mdd_io_offsets(s);
                for (uint64_t k = 0; k < nd; k++) {
                    if (off[k] > off[k + 1]) return 5;
                    sink += touch(mdd_io_text(s) + off[k], off[k + 1] - off[k]);
                }
                // egress: the documents back out, ids = their ordinal
                std::string ids;
                for (uint64_t k = 0; k < nd; k++) ids += (k ? "\n" : "") + std::to_string(k);
                for (int gz = 0; gz < 2; gz++) {
                    std::string out = std::string(argv[a]) + (gz ? ".out.gz" : ".out");
                    uint64_t written = 0;
                    int wrc = mdd_io_write_jsonl(out.c_str(), "\"doc_id\"", "\"text\"", reinterpret_cast<const uint8_t*>(ids.data()), ids.size(),
                                                 mdd_io_text(s), off, nd, gz ? 6 : 0, &written);
                    sink += written + (wrc ? strlen(mdd_io_last_error()) : 0);
                }
                mdd_io_free(s);
            }
        }
    }
    printf("san_io ok: %llu reads, %llu documents, %llu fallback lines (%llu)\n", (unsigned long long)n_files, (unsigned long long)n_ok,
           (unsigned long long)n_fb, (unsigned long long)(sink & 0xFF));
    return 0;
}
