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


CLI_WORKER = r'''
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
from cpu_backend import OracleBackend
from map_neo_amd.cli import dedup as stage
args = stage.build_parser().parse_args(["--input_dir", {inp!r}, "--output_dir", {out!r}, "--normalize", "none",
                                        "--write_shards", "--workers", "2"])
stage.run(args, engine_factory=lambda device: OracleBackend())
'''


@pytest.mark.parametrize("world", [2, 5])
def test_stage_cli_two_ranks_gloo(tmp_path, mini_docs, world):
    """The stage entry point under torchrun with 2 ranks (oracle-backed CPU backend, gloo): every rank ingests
    its block of shards; ids / clusters / surviving rows equal the single-process oracle result.  With 5 ranks
    and 4 shards one rank has no documents at all."""
    import gzip
    from oracle import ref_py as R
    n = len(mini_docs)
    ids = [f"id{(i * 37) % n:04d}" for i in range(n)]
    inp, out = tmp_path / "in", tmp_path / "out"
    inp.mkdir()
    bounds = [0, 40, 95, 150, n]
    for f in range(4):
        # dotted shard names that differ only after the first dot (they used to collapse to ONE output file)
        with open(inp / f"part.{f:05d}-of-00004.jsonl", "w", encoding="utf8") as fh:
            for i in range(bounds[f], bounds[f + 1]):
                fh.write(json.dumps({"doc_id": ids[i], "raw_content": mini_docs[i]}, ensure_ascii=False) + "\n")
    script = tmp_path / "cli_worker.py"
    script.write_text(CLI_WORKER.format(root=ROOT, inp=str(inp), out=str(out)))
    port = 31500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port + world), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lab = R.dedup(mini_docs)["labels"]
    comp = {}
    for k in range(n):
        comp.setdefault(int(lab[k]), []).append(k)
    exp_removed, exp_cluster = set(), {}
    for members in comp.values():
        if len(members) > 1:
            keep = min(ids[k] for k in members)
            for k in members:
                exp_cluster[ids[k]] = keep
                if ids[k] != keep:
                    exp_removed.add(ids[k])
    assert sorted(os.listdir(out)) == ["clusters.jsonl", "kept_ids.txt", "removed_ids.txt", "shards", "summary.json"]
    assert set(open(out / "removed_ids.txt").read().split()) == exp_removed
    kept = open(out / "kept_ids.txt").read().split()
    assert set(kept) == set(ids) - exp_removed and len(kept) + len(exp_removed) == n
    got = {c["doc_id"]: c["cluster"] for c in map(json.loads, open(out / "clusters.jsonl", encoding="utf8"))}
    assert got == exp_cluster
    assert sorted(os.listdir(out / "shards")) == [f"part.{f:05d}-of-00004.jsonl.gz" for f in range(4)]
    rows = [json.loads(l) for p in sorted(os.listdir(out / "shards")) for l in gzip.open(out / "shards" / p, "rt", encoding="utf8")]
    assert sorted(r["doc_id"] for r in rows) == sorted(set(ids) - exp_removed)
    s = json.load(open(out / "summary.json"))
    assert s["world"] == world and s["n_rows"] == n and s["n_removed_ids"] == len(exp_removed) and s["kept_rows_written"] == len(rows)


COLL_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from map_neo_amd.dedup import _all_to_all_v, _all_gather_v
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
width = 3
def count(src, dst): return (src * 7 + dst * 3) % 11 if (src + dst) % 4 else 0          # some pairs exchange nothing
def rec(src, dst, i): return [src * 1000000 + dst * 10000 + i, -(src + 1), dst]
send_counts = [count(rank, p) for p in range(world)]
recv_counts = [count(p, rank) for p in range(world)]
t = torch.tensor([v for p in range(world) for i in range(send_counts[p]) for v in rec(rank, p, i)], dtype=torch.int32)
exp = torch.tensor([v for p in range(world) for i in range(recv_counts[p]) for v in rec(p, rank, i)], dtype=torch.int32)
for limit in (None, 4 * width * world * 2, 1):                                           # one round, 2 records per peer per round, 1
    got = _all_to_all_v(t, send_counts, recv_counts, width, limit_bytes=limit)
    assert torch.equal(got, exp), (rank, limit)
mine = torch.arange(rank * 5 + (rank % 2) * 3, dtype=torch.int32) + 100 * rank
exp_g = torch.cat([torch.arange(p * 5 + (p % 2) * 3, dtype=torch.int32) + 100 * p for p in range(world)])
for limit in (None, 4 * world * 2, 1):
    assert torch.equal(_all_gather_v(mine, limit_bytes=limit), exp_g), (rank, limit)
dist.barrier()
dist.destroy_process_group()
'''


def test_collectives_in_rounds_gloo(tmp_path):
    """the bounded-size rounds of _all_to_all_v / _all_gather_v (RCCL drops the tail of > 1 GiB messages) give the
    same result as one call: ragged counts, empty pairs, 3 ranks"""
    script = tmp_path / "coll_worker.py"
    script.write_text(COLL_WORKER.format(root=ROOT))
    port = 33500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
