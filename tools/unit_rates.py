#!/usr/bin/env python3
"""Signature kernel rate in shingles/s on 1 M-document corpora of different shapes: EN char-13 (ASCII fast path), ZH char-13
(3-byte code points: 39-byte shingles), mixed, EN word-5.   python tools/unit_rates.py [n_docs]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from map_neo_amd import corpus
from map_neo_amd.minhash import Engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
eng = Engine(0)
for lang, unit, width in (("en", "char", 13), ("zh", "char", 13), ("mixed", "char", 13), ("en", "word", 5)):
    text, offs = corpus.generate_chunked(corpus.CorpusSpec(n, 2048, seed=20240519 + 2, lang=lang), eng.device)
    tb = int(offs[-1].item())
    def per_doc_count(mask):
        """number of set positions of `mask` (bool over the text bytes) inside every document"""
        c = torch.zeros(tb + 1, dtype=torch.int32, device=eng.device)
        torch.cumsum(mask.to(torch.int32), 0, out=c[1:])
        return (c[offs[1:]] - c[offs[:-1]]).to(torch.int64)

    lens = offs[1:] - offs[:-1]
    if unit == "char":
        # code points per document = bytes - continuation bytes; shingles = max(0, cps - width + 1) per document
        sh = int(torch.clamp(lens - per_doc_count((text[:tb] & 0xC0) == 0x80) - width + 1, min=0).sum().item())
    else:
        # words = spaces + 1 (the generator separates words by single spaces); empty documents have none
        words = torch.where(lens > 0, per_doc_count(text[:tb] == 32) + 1, torch.zeros_like(lens))
        sh = int(torch.clamp(words - width + 1, min=0).sum().item())
    eng.signatures(text, offs, 128, width, unit)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        eng.signatures(text, offs, 128, width, unit)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print("%-6s %s-%d: %7.2f ms  %6.2f G shingles/s  %5.1f GB/s  (%d docs, %.2f GB, %.3g shingles)" %
          (lang, unit, width, best * 1e3, sh / best / 1e9, tb / best / 1e9, n, tb / 1e9, sh), flush=True)
    del text, offs
