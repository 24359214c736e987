#!/usr/bin/env python3
"""Signature kernel time on the C2 corpus as a function of mdd_ctx_set_filter_from (the shingle count from which the
permute-phase filter is used; results never change).  python tools/filter_sweep.py [n_docs]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from map_neo_amd import corpus
from map_neo_amd.minhash import Engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
eng = Engine(0)
text, offs = corpus.generate_chunked(corpus.CorpusSpec(n, 2048, seed=20240519 + 2, lang="en"), eng.device)
ref = None
for f in (0, 1, 32, 64, 128, 256, 512, 1024, 0xFFFFFFFF):
    eng.L.mdd_ctx_set_filter_from(eng.ctx, f)
    sig = eng.signatures(text, offs)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        sig = eng.signatures(text, offs)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    if ref is None:
        ref = sig.clone()
    print("filter_from %10d: %7.2f ms  same=%s" % (f, best * 1e3, bool(torch.equal(sig, ref))), flush=True)
eng.L.mdd_ctx_set_filter_from(eng.ctx, 0)
