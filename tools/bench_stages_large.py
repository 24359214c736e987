#!/usr/bin/env python3
"""Exact stages on a C4-sized shard (12.5 M documents, ~27 GB of text) resident in HBM on one GPU:
exercises the > 2^31-byte paths and the key-partitioned passes of the substring stage."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from map_neo_amd import corpus, exact
from map_neo_amd.minhash import Engine


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=12_500_000)
    a = ap.parse_args()
    eng = Engine(0)
    spec = corpus.CorpusSpec(a.docs, 2048, seed=20240519 + 4, lang="en")
    t0 = time.time()
    text, offs = corpus.generate_chunked(spec, eng.device)
    total = int(offs[-1].item())
    # paragraphs: every 40th space becomes a newline (chunked: torch's rand on > 2^31 elements is avoided)
    g = torch.Generator(device=eng.device).manual_seed(1)
    for c in range(0, total, 1 << 30):
        b = text[c: min(total, c + (1 << 30))]
        r = torch.rand(b.numel(), device=eng.device, generator=g)
        b.copy_(torch.where((b == 32) & (r < 0.025), torch.full_like(b, 10), b))
    torch.cuda.synchronize()
    print(json.dumps({"docs": a.docs, "text_bytes": total, "gen_s": time.time() - t0}), flush=True)
    for name, fn in (("exact_document_dedup", lambda: exact.exact_document_dedup_packed(eng, text, offs)),
                     ("paragraph_dedup", lambda again=(lambda t, o: exact.paragraph_dedup_packed(eng, t, o, 20)): again(text, offs)),
                     ("exact_substr_dedup", lambda: exact.exact_substr_dedup_packed(eng, text, offs, 50, 20))):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res = {"stage": name, "s": dt, "docs_per_s": a.docs / dt, "text_gb_per_s": total / dt / 1e9}
        if name == "exact_document_dedup":
            res["n_removed"] = int((~out).sum().item())
        else:
            res["bytes_out"] = int(out[1][-1].item())
            res["docs_out"] = int(out[2].numel())
        if name == "exact_substr_dedup":
            res["key_partition_passes"] = eng.last_substr_parts
            res["method"] = eng.last_substr_method
            # the first call pays for mapping ~150 GB of workspace (hipMalloc through torch's caching allocator); a pipeline
            # that runs the stage batch after batch keeps the blocks: time it again
            del out
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = fn()
            torch.cuda.synchronize()
            res["s_warm"] = time.perf_counter() - t0
            res["docs_per_s_warm"] = a.docs / res["s_warm"]
            res["key_partition_passes_warm"] = eng.last_substr_parts
        if name == "paragraph_dedup":
            # size-independent property: after the stage no paragraph repeats, so a second pass changes nothing
            again = fn.__defaults__[0](out[0], out[1])
            n_b = int(out[1][-1].item())
            res["idempotent"] = bool(again[2].numel() == out[2].numel() and torch.equal(again[1], out[1])
                                     and torch.equal(again[0][:n_b], out[0][:n_b]))
            del again
        print(json.dumps(res), flush=True)
        del out
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
