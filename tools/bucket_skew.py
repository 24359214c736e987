#!/usr/bin/env python3
"""Bucket stage under skew (BASELINE.json configs[4] "exercising bucket skew"): times band_keys + bucket_edges +
union-find on synthetic signatures with (a) no planted buckets, (b) ONE bucket of 1 M records, (c) 10 % of the
records in 100 buckets, (d) every document in one bucket per band (the all-0xFFFFFFFF short-document case).
    python tools/bucket_skew.py [n_docs=1000000]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from map_neo_amd.dedup import StageTimer, cluster_local  # noqa: E402
from map_neo_amd.minhash import Engine  # noqa: E402


def make_sig(dev, n, case, seed=7):
    g = torch.Generator(device=dev).manual_seed(seed)
    sig = torch.randint(-2 ** 31, 2 ** 31 - 1, (n, 128), dtype=torch.int32, device=dev, generator=g)
    if case == "one_bucket_1M":
        hot = torch.randperm(n, device=dev, generator=g)[: min(n, 1_000_000)]
        sig[hot, 39:52] = sig[hot[0], 39:52]                       # band 3 of 1 M documents: one bucket
    elif case == "100_buckets_10pct":
        per = n * 9 // 10 // 100                                    # 10 % of the n*9 records, in 100 buckets
        perm = torch.randperm(n, device=dev, generator=g)
        for b in range(100):
            docs = perm[b * per:(b + 1) * per]
            j = b % 9
            sig[docs, j * 13:(j + 1) * 13] = sig[docs[0], j * 13:(j + 1) * 13]
    elif case == "all_equal":
        sig[:] = -1                                                 # 0xFFFFFFFF everywhere: 9 buckets of n records
    return sig


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    eng = Engine(0)
    out = {}
    for case in ("uniform", "one_bucket_1M", "100_buckets_10pct", "all_equal"):
        sig = make_sig(eng.device, n, case)
        cluster_local(eng, sig)
        torch.cuda.synchronize()
        tm = StageTimer()
        for _ in range(3):
            tm.mark("start")
            r = cluster_local(eng, sig, timer=tm)
        ms = {k: v / 3 for k, v in tm.result_ms().items()}
        out[case] = {"n_docs": n, "records": n * 9, "ms": ms, "n_edges": int(r["edge_u"].numel()), "n_removed": int(r["n_removed"].item())}
        del sig, r
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
