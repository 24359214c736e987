#!/usr/bin/env python3
"""Write bench.py's synthetic corpus (same generator, seed and parameters) to a flat file that
tools/sig_bench.hip reads with MDD_CORPUS_FILE: u64 n, u64 text_bytes, u64 offsets[n+1], text.
    python tools/dump_corpus.py /tmp/c2.bin [n_docs=1000000] [mean_bytes=2048] [lang=en] [config_id=2]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from map_neo_amd import corpus  # noqa: E402

path = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
mean = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
lang = sys.argv[4] if len(sys.argv) > 4 else "en"
cfg = int(sys.argv[5]) if len(sys.argv) > 5 else 2
dev = "cuda" if torch.cuda.is_available() else "cpu"
spec = corpus.CorpusSpec(n, mean, seed=20240519 + cfg, lang=lang)
text, offs = corpus.generate_chunked(spec, dev)
tb = int(offs[-1].item())
with open(path, "wb") as f:
    f.write(np.array([n, tb], np.uint64).tobytes())
    f.write(offs.cpu().numpy().astype(np.uint64).tobytes())
    f.write(text[:tb].cpu().numpy().tobytes())
print(f"{path}: {n} docs, {tb} bytes")
