#!/usr/bin/env python3
"""Development probe for minhash.HostPipeline: where the time of signatures-from-pinned-host goes.
Times (a) the H2D copy alone, per batch size, (b) the same copies while a resident sig_kernel runs on
another stream, (c) the pipeline for several batch sizes."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from map_neo_amd import corpus  # noqa: E402
from map_neo_amd.minhash import Engine, HostPipeline  # noqa: E402


def main():
    eng = Engine(0)
    spec = corpus.CorpusSpec(1_000_000, 2048)
    text, offs = corpus.generate(spec, device=eng.device)
    nbytes = int(offs[-1].item())
    host = torch.empty(text.numel(), dtype=torch.uint8).pin_memory()
    host.copy_(text.cpu())
    offs_h = offs.cpu().numpy()
    out = {}
    eng.signatures(text, offs, 128)
    torch.cuda.synchronize()
    t = time.perf_counter(); eng.signatures(text, offs, 128); torch.cuda.synchronize()
    out["resident_ms"] = (time.perf_counter() - t) * 1e3
    dst = torch.empty_like(text)
    cs = torch.cuda.Stream()
    for mb in (64, 256, 1024):
        step = mb << 20
        torch.cuda.synchronize()
        t = time.perf_counter()
        with torch.cuda.stream(cs):
            for lo in range(0, nbytes, step):
                dst[lo:lo + step].copy_(host[lo:lo + step], non_blocking=True)
        torch.cuda.synchronize()
        alone = (time.perf_counter() - t) * 1e3
        t = time.perf_counter()
        eng.signatures(text, offs, 128)                   # compute stream busy ...
        with torch.cuda.stream(cs):
            for lo in range(0, nbytes, step):
                dst[lo:lo + step].copy_(host[lo:lo + step], non_blocking=True)
        cs.synchronize()
        under = (time.perf_counter() - t) * 1e3          # ... copies finish after this long
        torch.cuda.synchronize()
        both = (time.perf_counter() - t) * 1e3
        out[f"h2d_{mb}MB_batches"] = {"alone_ms": alone, "copies_done_under_kernel_ms": under, "kernel_and_copies_ms": both}
    for mb in (32, 64, 128, 256, 512):
        pipe = HostPipeline(eng, batch_bytes=mb << 20)
        pipe.signatures(host, offs_h, 128)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t = time.perf_counter(); pipe.signatures(host, offs_h, 128); torch.cuda.synchronize()
            ts.append((time.perf_counter() - t) * 1e3)
        out[f"pipeline_{mb}MB_ms"] = [round(x, 1) for x in ts]
        del pipe
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
