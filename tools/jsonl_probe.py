#!/usr/bin/env python3
"""Where the time of the device-side JSON-lines ingest goes (development tool): N ZH documents in S plain shards;
file read into pinned memory, H2D, mdd_jsonl_* kernels, signatures -- per phase, and the kernels alone on one big shard."""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from map_neo_amd import corpus, hostio  # noqa: E402
from map_neo_amd.minhash import Engine, padded_size  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    shards = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    lang = sys.argv[3] if len(sys.argv) > 3 else "zh"
    spec = corpus.CorpusSpec(n, 2048, seed=9, lang=lang)
    text, offs = corpus.generate(spec)
    tb, o = text.numpy().tobytes(), offs.numpy()
    root = tempfile.mkdtemp(prefix="mdd_jsonl_probe_")
    per = (n + shards - 1) // shards
    files = []
    for s in range(shards):
        p = os.path.join(root, f"part-{s:04d}.jsonl")
        files.append(p)
        with open(p, "w", encoding="utf8") as f:
            for i in range(s * per, min(n, (s + 1) * per)):
                f.write(json.dumps({"text": tb[o[i]:o[i + 1]].decode(), "url": "x"}, ensure_ascii=False) + "\n")
    total = sum(os.path.getsize(p) for p in files)
    t0 = time.perf_counter()
    eng = Engine(0)
    eng.warm_up()
    t_engine = time.perf_counter() - t0
    res = {"docs": n, "shards": shards, "jsonl_bytes": total, "engine_s": t_engine}
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(16)
    for rep in range(2):
        ph = {}
        t_all = time.perf_counter()
        t = time.perf_counter()
        pre = hostio.read_files_pinned(files, pool)
        ph["read_pinned_16_threads"] = time.perf_counter() - t
        host, starts, _ = pre
        nb = int(starts[-1])
        t = time.perf_counter()
        raw = torch.empty(padded_size(nb), dtype=torch.uint8, device=eng.device)
        raw[:nb].copy_(host[:nb], non_blocking=True)
        raw[nb:].zero_()
        torch.cuda.synchronize()
        ph["h2d"] = time.perf_counter() - t
        t = time.perf_counter()
        x = eng.jsonl_extract(raw, nb, "text", None)
        torch.cuda.synchronize()
        ph["extract"] = time.perf_counter() - t
        t = time.perf_counter()
        eng.signatures(x["text"], x["offsets"], text_bytes=x["text_bytes"])
        torch.cuda.synchronize()
        ph["signatures"] = time.perf_counter() - t
        ph["wall"] = time.perf_counter() - t_all
        t = time.perf_counter()
        r = hostio.read_jsonl_batch_device(eng, files, "text", None)
        eng.signatures(r["text"], r["offsets"], text_bytes=r["text_bytes"])
        torch.cuda.synchronize()
        ph["batch_reader_plus_signatures_wall"] = time.perf_counter() - t
        ph["batch_reader_gb_per_s"] = total / ph["batch_reader_plus_signatures_wall"] / 1e9
        t = time.perf_counter()
        for p in files:
            r = hostio.read_jsonl_shard_device(eng, p, "text", None)
            eng.signatures(r["text"], r["offsets"], text_bytes=r["text_bytes"])
        torch.cuda.synchronize()
        ph["per_shard_reader_plus_signatures_wall"] = time.perf_counter() - t
        res["pass%d" % rep] = ph
    # the kernels alone on ONE big shard (all files concatenated)
    big = b"".join(open(p, "rb").read() for p in files)
    raw = torch.zeros(padded_size(len(big)), dtype=torch.uint8, device=eng.device)
    raw[:len(big)].copy_(torch.from_numpy(np.frombuffer(big, dtype=np.uint8).copy()))
    torch.cuda.synchronize()
    for rep in range(2):
        t = time.perf_counter()
        x = eng.jsonl_extract(raw, len(big), "text", None)
        torch.cuda.synchronize()
        res["one_big_shard_extract_s_%d" % rep] = time.perf_counter() - t
    res["one_big_shard_gb_per_s"] = len(big) / res["one_big_shard_extract_s_1"] / 1e9
    res["fallback_lines"] = int((x["status"] == 2).sum())
    print(json.dumps(res))
    import shutil
    shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
