#!/usr/bin/env python3
"""Per-stage timings of the other BASELINE.json configurations on ONE GPU (development /
documentation tool; bench.py is the judged benchmark).  Shapes are one GPU's shard:
  C3  2.5M mixed EN/ZH docs ~2 KB, K=256, 20x12      (20M docs / 8 GPUs)
  C5  250k docs ~20 KB, (9,13,128) | (14,9,128) | (20,12,256)   (5M docs / 8 GPUs, scaled)
Prints one JSON object per configuration."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from map_neo_amd import corpus  # noqa: E402
from map_neo_amd.dedup import StageTimer, dedup_local  # noqa: E402
from map_neo_amd.minhash import Engine  # noqa: E402


def run(eng, name, spec, K, bands, rows, steps=2):
    t0 = time.time()
    text, offs = corpus.generate_chunked(spec, eng.device, docs_per_chunk=max(1024, int(32768 * 2048 / spec.mean_bytes)))
    torch.cuda.synchronize()
    tg = time.time() - t0
    nbytes = int(offs[-1].item())
    dedup_local(eng, text, offs, K, 13, "char", bands, rows, text_bytes=nbytes)
    torch.cuda.synchronize()
    timer = StageTimer()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = dedup_local(eng, text, offs, K, 13, "char", bands, rows, text_bytes=nbytes, timer=timer)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    st = {k: v / steps for k, v in timer.result_ms().items()}
    n_rec = spec.n_docs * bands
    res = {"config": name, "n_docs": spec.n_docs, "text_bytes": nbytes, "K": K, "bands": bands, "rows": rows,
           "ms_per_step": dt * 1e3, "docs_per_s": spec.n_docs / dt, "text_gb_per_s": nbytes / dt / 1e9,
           "stage_ms": st, "n_removed": int(out["n_removed"].item()), "n_edges": int(out["edge_u"].numel()),
           "n_records": n_rec, "gen_s": tg,
           # HBM-bound stages: algorithmic bytes / stage time
           "band_keys_gb_per_s": n_rec * (4 * rows + 16) / (st["band_keys"] * 1e-3) / 1e9,
           "bucket_edges_gb_per_s": n_rec * (8 + 16 + 8 * rows + 8 + 12 * 2) / (st["bucket_edges"] * 1e-3) / 1e9}
    print(json.dumps(res), flush=True)
    del text, offs, out
    torch.cuda.empty_cache()


def main():
    which = sys.argv[1:] or ["C3", "C4", "C5"]
    eng = Engine(0)
    if "C3" in which:
        run(eng, "C3-shard", corpus.CorpusSpec(2_500_000, 2048, seed=20240519 + 3, lang="mixed"), 256, 20, 12)
    if "C4" in which:
        # one GPU's share of C4 (12.5 M of 100 M documents), cut to 2.5 M here: clusters of 2-50 near-copies (Zipf)
        run(eng, "C4-shard (2.5M docs)", corpus.CorpusSpec(2_500_000, 2048, seed=20240519 + 4, lang="en", dup_frac=0.15,
                                                           edit_rates=(0.01, 0.05, 0.1, 0.2), indel_rate=0.005, clusters="zipf50"),
            128, 9, 13)
    if "C5" in which:
        # C5 as SURVEY 8(d) describes it: long documents, boilerplate-heavy ones for bucket skew, edit-rate sweep with
        # token insertions / deletions
        spec = corpus.CorpusSpec(250_000, 20480, seed=20240519 + 5, lang="en", dup_frac=0.15, edit_rates=(0.01, 0.05, 0.1, 0.2),
                                 indel_rate=0.002, boiler_frac=0.05, boiler_share=0.9, n_templates=200)
        for (b, r, K) in [(9, 13, 128), (14, 9, 128), (20, 12, 256)]:
            run(eng, f"C5-shard b{b}xr{r} K{K}", spec, K, b, r)


if __name__ == "__main__":
    main()
