"""Worker of tests/test_gpu_comm_loopback.py (run as a script in a FRESH process: the RCCL binding of libmdd_hip.so
is made once per process, so the loop-back transport must be chosen before anything binds the real one).

W "ranks" = W threads of this process, each with its own torch stream and mdd context on GPU 0, drive the library's
native exchange (mdd_comm_init, mdd_exchange_counts / _bands, mdd_allgather_counts / _edges) through
tests/native/libfake_rccl.so (MDD_RCCL_LIB test hook).  Checks, printed as one JSON line:
  * what every rank received in the band-record all-to-all == the definition of a variable all-to-all applied to what
    the ranks sent (the semantics of dedup.TorchComm._all_to_all_v), byte for byte;
  * the gathered edges == the ranks' forests back to back;
  * labels of every rank == dedup_local's on the whole corpus; removed masks likewise;
  * the exchange ran in >= 3 rounds per pair (MDD_COMM_MSG_BYTES lowered) and no message exceeded the limit;
  * an injected ncclSend failure comes back as an error on every rank, leaves no group open, and the next exchange works.
"""
import ctypes as C
import json
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FAKE = os.path.join(ROOT, "tests", "native", "libfake_rccl.so")


class Recorder:
    """wraps a comm object and keeps host copies of what went in and came out"""

    def __init__(self, inner):
        self.inner, self.world, self.rank, self.name = inner, inner.world, inner.rank, inner.name
        self.log = {}

    def exchange(self, tuples_s, meta_s, send_counts, rows):
        out = self.inner.exchange(tuples_s, meta_s, send_counts, rows)
        self.log["ex"] = dict(ts=tuples_s.cpu().numpy().copy(), ms=meta_s.cpu().numpy().copy(), rows=rows,
                              tr=out[0].cpu().numpy().copy(), mr=out[1].cpu().numpy().copy(), send=list(out[2]), recv=list(out[3]))
        return out

    def allgather_edges(self, eu, ev):
        out = self.inner.allgather_edges(eu, ev)
        self.log["ag"] = dict(eu=eu.cpu().numpy().copy(), ev=ev.cpu().numpy().copy(), au=out[0].cpu().numpy().copy(),
                              av=out[1].cpu().numpy().copy())
        return out


def main():
    world = int(sys.argv[1])
    msg_bytes = int(os.environ.get("MDD_COMM_MSG_BYTES", "0"))
    assert os.environ.get("MDD_RCCL_LIB") == FAKE, "run through tests/test_gpu_comm_loopback.py"
    import torch

    from map_neo_amd import corpus
    from map_neo_amd.dedup import NativeComm, cluster_distributed, dedup_local
    from map_neo_amd.lib import MddError
    from map_neo_amd.minhash import Engine, as_u32

    fake = C.CDLL(FAKE)
    fake.fake_rccl_stats.argtypes = [C.c_void_p, C.c_int]
    fake.fake_rccl_fail_send.argtypes = [C.c_uint64]
    main_eng = Engine(0)
    n_docs = 6000
    spec = corpus.CorpusSpec(n_docs, 500, lang="mixed", dup_frac=0.3, edit_rate=0.01, short_frac=0.01, seed=77)
    text, offs = corpus.generate(spec, device=main_eng.device)
    ref = dedup_local(main_eng, text, offs)
    sig_all = ref["sig"]
    ref_labels = as_u32(ref["labels"])
    ref_removed = ref["removed"].cpu().numpy()
    torch.cuda.synchronize()
    # unequal shards; with >= 3 ranks one rank holds NO documents (zero counts on the send side)
    cuts = [0]
    weights = [(3 + 2 * r) if not (world >= 3 and r == 1) else 0 for r in range(world)]
    for r in range(world):
        cuts.append(cuts[-1] + n_docs * weights[r] // sum(weights))
    cuts[-1] = n_docs
    uid = [None]
    barrier = threading.Barrier(world)
    results = [None] * world
    errors = []

    def run(rank):
        try:
            stream = torch.cuda.Stream(device=main_eng.device)
            with torch.cuda.stream(stream):
                eng = Engine(0)
                if rank == 0:
                    uid[0] = eng.comm_unique_id()
                barrier.wait()
                comm = Recorder(NativeComm.from_unique_id(eng, uid[0], rank, world))
                lo, hi = cuts[rank], cuts[rank + 1]
                sig = sig_all[lo:hi].clone()
                out = cluster_distributed(eng, sig, lo, n_docs, comm=comm)
                stream.synchronize()
                res = dict(labels=as_u32(out["labels"]).copy(), removed=out["removed"].cpu().numpy().copy(), log=comm.log, lo=lo, hi=hi)
                # ---- error path: the 3rd ncclSend of every rank fails
                barrier.wait()
                if rank == 0:
                    fake.fake_rccl_fail_send(3)
                barrier.wait()
                fake.fake_rccl_reset_thread()
                failed = ""
                try:
                    cluster_distributed(eng, sig, lo, n_docs, comm=comm.inner)
                except MddError as e:
                    failed = str(e)
                res["fault_error"] = failed
                res["group_depth_after_fault"] = int(fake.fake_rccl_group_depth())
                stream.synchronize()
                barrier.wait()
                if rank == 0:
                    fake.fake_rccl_fail_send(0)
                barrier.wait()
                out2 = cluster_distributed(eng, sig, lo, n_docs, comm=comm.inner)
                stream.synchronize()
                res["labels_after_fault"] = as_u32(out2["labels"]).copy()
                results[rank] = res
                eng.close()
        except BaseException as e:      # noqa: BLE001
            import traceback
            errors.append("rank %d: %s\n%s" % (rank, e, traceback.format_exc()))
            try:
                barrier.abort()
            except Exception:
                pass

    fake.fake_rccl_stats(None, 1)
    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        print(json.dumps({"ok": False, "errors": errors}))
        return 1
    stats = (C.c_uint64 * 6)()
    fake.fake_rccl_stats(stats, 0)
    rep = {"world": world, "msg_bytes": msg_bytes, "shards": [cuts[r + 1] - cuts[r] for r in range(world)],
           "fake_groups": int(stats[0]), "fake_sends": int(stats[1]), "fake_recvs": int(stats[2]), "fake_bytes": int(stats[3]),
           "fake_largest_message": int(stats[4]), "fake_allgathers": int(stats[5])}
    bad = []
    # --- the all-to-all against its definition
    rows = results[0]["log"]["ex"]["rows"]
    n_sent_other = 0
    for r in range(world):
        ex = results[r]["log"]["ex"]
        exp_t, exp_m = [], []
        for p in range(world):
            sp = results[p]["log"]["ex"]
            soff = int(sum(sp["send"][:r]))
            cnt = int(sp["send"][r])
            if cnt != int(ex["recv"][p]):
                bad.append("rank %d expects %d records from %d, which sent %d" % (r, ex["recv"][p], p, cnt))
            exp_t.append(sp["ts"][soff * rows:(soff + cnt) * rows])
            exp_m.append(sp["ms"][soff:soff + cnt])
            if p != r:
                n_sent_other += cnt
        exp_t, exp_m = np.concatenate(exp_t), np.concatenate(exp_m)
        if exp_t.tobytes() != ex["tr"].tobytes():
            bad.append("rank %d: received tuples differ from the all-to-all's definition" % r)
        if exp_m.tobytes() != ex["mr"].tobytes():
            bad.append("rank %d: received meta differ from the all-to-all's definition" % r)
    rep["records_between_ranks"] = n_sent_other
    rep["max_pair_records"] = int(max(results[p]["log"]["ex"]["send"][r] for p in range(world) for r in range(world) if p != r))
    per_round = max(1, msg_bytes // (4 * rows)) if msg_bytes else None
    rep["rounds_of_largest_pair"] = -(-rep["max_pair_records"] // per_round) if per_round else 1
    # --- the all-gather
    exp_u = np.concatenate([results[p]["log"]["ag"]["eu"] for p in range(world)])
    exp_v = np.concatenate([results[p]["log"]["ag"]["ev"] for p in range(world)])
    rep["forest_edges"] = [int(results[p]["log"]["ag"]["eu"].size) for p in range(world)]
    for r in range(world):
        ag = results[r]["log"]["ag"]
        if exp_u.tobytes() != ag["au"].tobytes() or exp_v.tobytes() != ag["av"].tobytes():
            bad.append("rank %d: gathered edges differ from the ranks' forests back to back" % r)
    # --- end result
    for r in range(world):
        res = results[r]
        if not (res["labels"] == ref_labels).all():
            bad.append("rank %d: labels differ from dedup_local" % r)
        if not (res["removed"] == ref_removed[res["lo"]:res["hi"]]).all():
            bad.append("rank %d: removed mask differs from dedup_local" % r)
        if not res["fault_error"]:
            bad.append("rank %d: the injected ncclSend failure was not reported" % r)
        if res["group_depth_after_fault"] != 0:
            bad.append("rank %d: a group was left open after the failure (depth %d)" % (r, res["group_depth_after_fault"]))
        if not (res["labels_after_fault"] == ref_labels).all():
            bad.append("rank %d: labels differ after the recovered fault" % r)
    rep["fault_error"] = results[0]["fault_error"][:160]
    rep["n_removed"] = int(ref_removed.sum())
    if msg_bytes:
        if rep["fake_largest_message"] > msg_bytes:
            bad.append("a message of %d bytes exceeds the limit %d" % (rep["fake_largest_message"], msg_bytes))
        if rep["rounds_of_largest_pair"] < 3:
            bad.append("the exchange did not need 3 rounds: lower MDD_COMM_MSG_BYTES")
    rep["ok"] = not bad
    rep["errors"] = bad
    print(json.dumps(rep))
    return 0 if not bad else 1


if __name__ == "__main__":
    sys.exit(main())
