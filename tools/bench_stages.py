#!/usr/bin/env python3
"""Throughput of the stages around MinHash on a C2-shaped corpus:
device EN normalisation, exact document grouping, exact 50-character-window grouping.
Prints one JSON line per stage; HBM-bound, so each line carries algorithmic GB/s vs 8 TB/s."""
import argparse
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from map_neo_amd import corpus
from map_neo_amd.minhash import Engine, padded_size

HBM_PEAK = 8000.0  # GB/s


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=1_000_000)
    ap.add_argument("--mean-bytes", type=int, default=2048)
    ap.add_argument("--window-bytes", type=int, default=400_000_000, help="text prefix cut into 50-char windows")
    a = ap.parse_args()
    eng = Engine(0)
    spec = corpus.CorpusSpec(n_docs=a.docs, mean_bytes=a.mean_bytes, lang="en", seed=20240521)
    d_text, d_offs = corpus.generate_chunked(spec, eng.device)
    total = int(d_offs[-1].item())
    # make the raw text look un-normalised: upper-case 1/8 of the letters, turn 1/16 of the spaces into commas
    g = torch.Generator(device=eng.device).manual_seed(1)
    d_raw = d_text.clone()
    body = d_raw[:total]
    r = torch.rand(total, device=eng.device, generator=g)
    letters = (body >= 97) & (body <= 122)
    body.copy_(torch.where(letters & (r < 0.125), body - 32, body))
    body.copy_(torch.where((body == 32) & (r > 0.9375), torch.full_like(body, ord(",")), body))
    del r, letters
    out = {}

    res = {}
    def norm():
        res["n"] = eng.normalize_en(d_raw, d_offs)
    ms = timed(norm)
    n_text, n_offs, flags = res["n"]
    out_bytes = int(n_offs[-1].item())
    alg = 2 * total + out_bytes
    print(json.dumps({"stage": "normalize_en (count + scan + write)", "docs": a.docs, "text_bytes": total, "out_bytes": out_bytes,
                      "ms": ms, "text_gb_per_s": total / ms / 1e6, "algorithmic_gb_per_s": alg / ms / 1e6,
                      "hbm_frac": alg / ms / 1e6 / HBM_PEAK, "non_ascii_docs": int(flags.sum().item())}))

    ub, ue = d_offs[:-1].contiguous(), d_offs[1:].contiguous()
    def docs():
        res["d"] = eng.exact_group(d_text, ub, ue)
    ms = timed(docs)
    first = res["d"].cpu().numpy().view(np.uint32)
    n_dup = int((first != np.arange(a.docs, dtype=np.uint32)).sum())
    alg = total + 16 * a.docs + 4 * a.docs
    print(json.dumps({"stage": "exact document grouping", "units": a.docs, "ms": ms, "units_per_s": a.docs / ms * 1e3,
                      "text_gb_per_s": total / ms / 1e6, "algorithmic_gb_per_s": alg / ms / 1e6,
                      "hbm_frac": alg / ms / 1e6 / HBM_PEAK, "duplicates": n_dup, "hash_collisions": eng.last_exact_collisions}))

    wb = min(a.window_bytes, total - 64)
    beg = torch.arange(0, wb - 50, dtype=torch.int64, device=eng.device)
    end = beg + 50
    def windows():
        res["w"] = eng.exact_group(d_text, beg, end)
    ms = timed(windows, reps=2)
    nu = beg.numel()
    alg = wb + 20 * nu
    first = res["w"]
    n_dup = int((first.view(torch.int32) != torch.arange(nu, dtype=torch.int32, device=eng.device)).sum().item())
    print(json.dumps({"stage": "exact 50-char window grouping (substring dedup core)", "units": nu, "ms": ms,
                      "units_per_s": nu / ms * 1e3, "text_gb_per_s": wb / ms / 1e6, "algorithmic_gb_per_s": alg / ms / 1e6,
                      "hbm_frac": alg / ms / 1e6 / HBM_PEAK, "duplicate_windows": n_dup,
                      "hash_collisions": eng.last_exact_collisions}))


if __name__ == "__main__":
    main()
