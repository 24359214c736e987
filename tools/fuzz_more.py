#!/usr/bin/env python3
"""Extended run of tests/test_gpu_fuzz.py's random sweep (signatures and labels against the C oracle) over many more seeds than
the test suite affords, with longer documents so that scan windows, segments and claims interact.  python tools/fuzz_more.py [n_seeds] [first_seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

from map_neo_amd.minhash import Engine, as_u32, pack_texts
from oracle import ref_c as RC
import test_gpu_fuzz as F

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 150
first_seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # a later range of seeds: [first_seed, first_seed + n_seeds)
eng = Engine(0)
bad = 0
for seed in range(first_seed, first_seed + n_seeds):
    rng = np.random.Generator(np.random.PCG64(50_000 + seed))
    docs = F._docs(rng, 120)
    # some long documents (several scan windows / segments), with leading / trailing / runs of spaces
    for _ in range(12):
        parts = [docs[int(rng.integers(0, len(docs)))] for _ in range(int(rng.integers(2, 9)))]
        docs.append((" " * int(rng.integers(0, 3))).join(parts) + " " * int(rng.integers(0, 3)))
    K = int(rng.choice([1, 64, 128, 129, 256]))
    unit = "char" if rng.random() < 0.5 else "word"
    width = int(rng.integers(1, 41)) if unit == "char" else int(rng.integers(1, 9))
    seg = int(rng.choice([0, 256, 512, 768]))
    buf, offs = pack_texts(docs)
    eng.set_segment_bytes(seg)
    try:
        got = as_u32(eng.signatures(eng.to_device(buf), eng.to_device(offs), K, width, unit))
    finally:
        eng.set_segment_bytes(0)
    exp = RC.signatures(buf, offs, K, width, unit)
    ok = bool((got == exp).all())
    if not ok:
        bad += 1
        rows = np.nonzero((got != exp).any(axis=1))[0]
        print(f"seed {seed}: MISMATCH K={K} {unit}-{width} seg={seg} docs {rows[:5].tolist()} lens {[len(docs[i]) for i in rows[:5]]}", flush=True)
    elif seed % 25 == 0:
        print(f"seed {seed}: ok (K={K} {unit}-{width} seg={seg}, {len(docs)} docs)", flush=True)
print("mismatching seeds:", bad, "of", n_seeds)
sys.exit(1 if bad else 0)
