#!/usr/bin/env python3
"""Exact substring dedup on the C2 corpus: table method vs sorted method (times, identical survivors).
    python tools/substr_probe.py [n_docs=1000000] [methods=sorted,table,sorted]   ("sorted-nofilter": without mdd_substr_candidates)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from map_neo_amd import corpus  # noqa: E402
from map_neo_amd.minhash import Engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
eng = Engine(0)
spec = corpus.CorpusSpec(n, 2048, seed=20240519 + 2, lang="en")
text, offs = corpus.generate_chunked(spec, eng.device)
res = {}
for method in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("sorted", "table", "sorted")):
    eng.substr_filter = not method.endswith("-nofilter")        # "sorted-nofilter": every window is sorted (round 2's method)
    key, method = method, method.split("-")[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = eng.substr_dedup(text, offs, 50, 20, method=method)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    print(f"{key}: {ms:.1f} ms, parts {eng.last_substr_parts}, kept bytes {int(r[1][-1].item())}, docs {r[2].numel()}, "
          f"candidates {eng.last_substr_candidates}", flush=True)
    res.setdefault(key, r)
keys = list(res)
for k in keys[1:]:
    a, b = res[keys[0]], res[k]
    nb = int(a[1][-1].item())
    print(f"{keys[0]} == {k}:", bool(torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[0][:nb], b[0][:nb])))
if os.environ.get("MDD_LEAF_PROF"):
    print("leaf phases (cycles summed over waves): wait/loads, clear, insert, drain, rotate:", eng.last_leaf_prof)
