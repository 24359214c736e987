"""CPU suite: the N>1 path (shard -> route -> all-to-all -> local buckets -> all-gather ->
replicated union-find) of map_neo_amd.dedup.dedup_distributed under gloo with world_size 2
and 3, using the oracle-backed test backend.  Labels must equal the single-process oracle."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
from cpu_backend import OracleBackend
from map_neo_amd.dedup import dedup_distributed
from map_neo_amd.minhash import pack_texts
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
docs = json.load(open(os.path.join({root!r}, "tests", "golden", "mini_corpus.json"), encoding="utf8"))
n = len(docs)
per = (n + world - 1) // world
lo, hi = rank * per, min(n, (rank + 1) * per)
buf, offs = pack_texts(docs[lo:hi])
be = OracleBackend()
stats = {{}}
out = dedup_distributed(be, torch.from_numpy(buf), torch.from_numpy(offs.view(np.int64)), lo, n,
                        {K}, 13, "char", {bands}, {rows}, timings=stats)
labels = out["labels"].numpy().view(np.uint32)
removed = out["removed"].numpy()
np.save(os.path.join({out!r}, f"labels_{{rank}}.npy"), labels)
np.save(os.path.join({out!r}, f"removed_{{rank}}.npy"), removed)
json.dump(stats, open(os.path.join({out!r}, f"stats_{{rank}}.json"), "w"))
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world,bands,rows,K", [(2, 9, 13, 128), (3, 20, 12, 256)])
def test_distributed_gloo_matches_oracle(tmp_path, mini_expected, world, bands, rows, K):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, out=str(tmp_path), K=K, bands=bands, rows=rows))
    port = 29500 + (os.getpid() % 2000) + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    exp = mini_expected[f"labels_b{bands}_r{rows}_K{K}"]
    n = len(exp)
    per = (n + world - 1) // world
    sent = recv = 0
    for rank in range(world):
        labels = np.load(tmp_path / f"labels_{rank}.npy")
        assert (labels == exp).all()                      # replicated, identical to the 1-process result
        removed = np.load(tmp_path / f"removed_{rank}.npy").astype(bool)
        lo, hi = rank * per, min(n, (rank + 1) * per)
        assert (removed == (exp[lo:hi] != np.arange(lo, hi))).all()
        st = json.load(open(tmp_path / f"stats_{rank}.json"))
        sent += st["records_sent"]
        recv += st["records_received"]
    assert sent == recv == n * bands                      # every band record reaches exactly one owner
