#!/usr/bin/env python3
"""Extended run of tests/test_gpu_fuzz.py::test_random_exact_stages' sweep: exact substring dedup with BOTH methods (the sorted
method forced on small inputs, where `auto` would take the table method), paragraph and exact document dedup, against the
oracle restatement.  python tools/fuzz_exact_more.py [n_seeds] [first_seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from map_neo_amd import exact
from map_neo_amd.minhash import Engine
from oracle import ref_exact as RX
import test_gpu_fuzz as F

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
first_seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # a later range of seeds: [first_seed, first_seed + n_seeds)
eng = Engine(0)
bad = 0
for seed in range(first_seed, first_seed + n_seeds):
    rng = np.random.Generator(np.random.PCG64(70_000 + seed))
    pieces = [p.replace("\n", " ") for p in F._docs(rng, 40) if p] or ["x"]
    docs = []
    for _ in range(80):
        k = int(rng.integers(0, 6))
        sep = "\n" if rng.random() < 0.7 else ""
        docs.append(sep.join(pieces[int(i)] for i in rng.integers(0, len(pieces), k)) + ("\n" if rng.random() < 0.2 else ""))
    window = int(rng.choice([1, 2, 13, 50, 50, 50, 60, 64, 128]))
    min_length = int(rng.choice([0, 1, 20, 200]))
    seg = int(rng.choice([0, 256, 1024]))
    rows = [(d, i) for i, d in enumerate(docs)]
    exp = RX.exact_substr_dedup(rows, window, min_length)
    eng.set_segment_bytes(seg)
    try:
        for method in ("table", "sorted", "sorted-nofilter"):      # sorted: anchors + in-leaf resolution from 50 code points on (round 3)
            eng.substr_method = method.split("-")[0]
            eng.substr_filter = method != "sorted-nofilter"
            got = exact.exact_substr_dedup(eng, rows, window, min_length)
            if got != exp:
                bad += 1
                print(f"seed {seed}: MISMATCH method={method} window={window} min_length={min_length} seg={seg}", flush=True)
    finally:
        eng.substr_method = "auto"
        eng.substr_filter = True
        eng.set_segment_bytes(0)
    ids = [int(i) for i in rng.permutation(len(docs))]
    prow = list(zip(ids, docs))
    if exact.paragraph_dedup(eng, prow, min_length) != RX.paragraph_dedup(prow, min_length):
        bad += 1
        print(f"seed {seed}: paragraph MISMATCH", flush=True)
    if exact.exact_document_dedup(eng, docs).tolist() != RX.exact_document_dedup(docs):
        bad += 1
        print(f"seed {seed}: exact document MISMATCH", flush=True)
    if seed % 20 == 0:
        print(f"seed {seed}: ok (window {window}, min_length {min_length}, seg {seg})", flush=True)
print("mismatches:", bad, "in", n_seeds, "seeds")
sys.exit(1 if bad else 0)
