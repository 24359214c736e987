#!/usr/bin/env python3
"""What the single-GPU path removes from the corpora of `bench.py --gpus N`: the global corpus of an N-GPU run (one spec
for the whole job, bench.bench_spec) hashed block by block on ONE MI355X (dedup.signatures_streamed) and clustered in one
call, with the size-independent properties and the oracle sample of tests/full_configs.py asserted on the way.  One
JSON line per job size -> profiles/r3_bench_expected.jsonl; bench.py reads it and reports whether a multi-GPU run lost
exactly the same number of documents (n_removed_matches_single_gpu).

    python tools/bench_expected.py [gpus ...]        default: 1 2 4 8 (12.5 M documents per GPU from 2 GPUs on)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import full_configs  # noqa: E402
from map_neo_amd.minhash import Engine  # noqa: E402


def main():
    gpus = [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8]
    scale = float(os.environ.get("MDD_FULL_SCALE", "1"))
    eng = Engine(0)
    out = os.environ.get("MDD_EXPECTED_SINK", os.path.join(ROOT, "gpurun_out", "bench_expected.jsonl"))
    for n_gpus in gpus:
        n_global = int((1_000_000 if n_gpus == 1 else 12_500_000 * n_gpus) * scale)
        cid = bench.bench_config_id(n_global)
        spec = bench.bench_spec(n_global, 2048, cid)
        tmp = out + ".tmp"
        if os.path.exists(tmp):
            os.remove(tmp)
        full_configs.run_config(eng, "bench --gpus %d" % n_gpus, spec, [(bench.BANDS, bench.ROWS, bench.K)], 262144, tmp, width=bench.WIDTH)
        r = json.loads(open(tmp).read().strip().splitlines()[-1])
        os.remove(tmp)
        r.update(gpus=n_gpus, mean_bytes=2048, corpus_config=cid)
        with open(out, "a") as f:
            f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
