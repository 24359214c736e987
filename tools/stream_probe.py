#!/usr/bin/env python3
"""Timing of the pinned-host -> signatures paths on the C2 corpus: resident, streamed single launch
(mdd_signatures_from_host), batched HostPipeline.   python tools/stream_probe.py [n_docs] [chunk_mib]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from map_neo_amd import corpus  # noqa: E402
from map_neo_amd.minhash import Engine, HostPipeline  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
chunk = (int(sys.argv[2]) if len(sys.argv) > 2 else 64) << 20
eng = Engine(0)
spec = corpus.CorpusSpec(n, 2048, seed=20240519 + 2, lang="en")
text, offs = corpus.generate_chunked(spec, eng.device)
tb = int(offs[-1].item())
host = torch.empty(text.numel(), dtype=torch.uint8, pin_memory=True)
host.copy_(text)
offs_h = offs.cpu().pin_memory()


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    return best, r


ms_res, ref = timed(lambda: eng.signatures(text, offs, text_bytes=tb))
stage = torch.empty_like(text)
ms_copy, _ = timed(lambda: stage.copy_(host, non_blocking=True))
print("plain H2D copy of the text: %.2f ms (%.1f GB/s)" % (ms_copy, tb / ms_copy / 1e6))
del stage
tdev = torch.empty_like(text)
out = torch.empty_like(ref)


def streamed():
    od = offs_h.to(eng.device, non_blocking=True)
    return eng.signatures_from_host(host, od, text_bytes=tb, text_dev=tdev, out=out, chunk_bytes=chunk)[0]


ms_str, got = timed(streamed)
# where does the time go: end of the copy stream's work vs end of the kernel, from a common start
e0, ec, ek = (torch.cuda.Event(enable_timing=True) for _ in range(3))
torch.cuda.synchronize()
e0.record()
streamed()
ec.record(eng._copy_stream)
ek.record()
torch.cuda.synchronize()
print("streamed: copies done at %.2f ms, kernel done at %.2f ms, waves waited %.1f wave-ms in total (4096 resident waves)"
      % (e0.elapsed_time(ec), e0.elapsed_time(ek), eng.stream_wait_ms()))
eng.sync()
print("resident %.2f ms, streamed %.2f ms (chunk %d MiB), equal %s" % (ms_res, ms_str, chunk >> 20, bool(torch.equal(got, ref))))
pipe = HostPipeline(eng)
ms_pipe, g2 = timed(lambda: pipe.signatures(host, offs_h.numpy()))
print("batched HostPipeline %.2f ms, equal %s" % (ms_pipe, bool(torch.equal(g2, ref))))
