"""The library's native multi-rank exchange (csrc/mdd_comm.hip: mdd_comm_init, mdd_exchange_counts / _bands,
mdd_allgather_counts / _edges -- what replaces Spark's shuffle, EN:136,140-142) with W > 1 peers on ONE GPU: W threads,
one mdd context and stream each, over the loop-back transport tests/native/fake_rccl.cpp (MDD_RCCL_LIB test hook).
The checks live in tests/loopback_worker.py (fresh process per world size: the RCCL binding is made once per process)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "native", "libfake_rccl.so")


@pytest.mark.gpu
@pytest.mark.parametrize("world,msg_bytes", [(2, 65536), (3, 32768), (8, 4096)])
def test_native_exchange_with_peers_over_loopback(world, msg_bytes):
    assert os.path.exists(FAKE), "build it first: make -C map_neo_amd/csrc fake_rccl (__graft_entry__.build does)"
    env = dict(os.environ, MDD_RCCL_LIB=FAKE, MDD_COMM_MSG_BYTES=str(msg_bytes), FAKE_RCCL_TIMEOUT_S="8")
    p = subprocess.run([sys.executable, "-u", os.path.join(ROOT, "tests", "loopback_worker.py"), str(world)], env=env,
                       capture_output=True, text=True, timeout=240)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert lines, "no report:\n" + p.stdout[-2000:] + p.stderr[-4000:]
    rep = json.loads(lines[-1])
    assert rep.get("ok"), json.dumps(rep, indent=1)[:6000] + p.stderr[-3000:]
    assert p.returncode == 0
    assert rep["world"] == world and rep["rounds_of_largest_pair"] >= 3 and rep["fake_largest_message"] <= msg_bytes
    assert rep["records_between_ranks"] > 0 and sum(rep["forest_edges"]) > 0
    if world >= 3:
        assert rep["shards"][1] == 0        # a rank without documents: zero send counts, still receives and gathers
