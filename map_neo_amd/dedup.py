"""Whole-path drivers: signatures -> band records -> buckets/edges -> components -> removed.

`dedup_local`       one GPU, everything stays in HBM.
`dedup_distributed` one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI):
                    documents are sharded by contiguous global index range, band records are
                    routed to their owner rank with an all-to-all so that every (band, tuple)
                    bucket lives on ONE GPU (EN:136's groupBy shuffle, ZH-1:58-61's per-band
                    directories), the edges every rank finds are all-gathered and every rank
                    runs the same union-find, so labels are identical on all ranks and equal
                    to the one-GPU result.

Both take a `backend` object implementing the stage methods of minhash.Engine.  The product
backend is Engine (HIP); tests may inject another backend to exercise the sharding / exchange
logic on CPU with gloo.  Nothing in this module computes a hash or a signature itself.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from .minhash import BANDS, NUM_PERM, ROWS, SEED, WIDTH


class StageTimer:
    """Optional per-stage timing with events on the current stream (bench only)."""

    def __init__(self):
        self.marks = []

    def mark(self, name: str):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self.marks.append((name, ev))

    def result_ms(self) -> dict:
        torch.cuda.synchronize()
        out = {}
        for (_, e0), (name, e1) in zip(self.marks[:-1], self.marks[1:]):
            out[name] = out.get(name, 0.0) + e0.elapsed_time(e1)
        return out


def verified_edges(be, sig: torch.Tensor, eu: torch.Tensor, ev: torch.Tensor, min_jaccard: float):
    """The edges whose two documents agree on at least ceil(min_jaccard * K) signature components (datasketch's
    MinHash.jaccard() >= min_jaccard).  An extension: the reference has no such step (EN:136-145 unions every pair)."""
    import math
    need = int(math.ceil(float(min_jaccard) * sig.shape[1] - 1e-9))
    keep = be.signature_matches(sig, eu, ev) >= need
    return eu[keep], ev[keep]


def cluster_local(be, sig: torch.Tensor, bands: int = BANDS, rows: int = ROWS, key_bits: int = 64,
                  want_buckets: bool = False, timer: Optional[StageTimer] = None, min_jaccard: Optional[float] = None):
    """Band records -> buckets/edges -> union-find -> removed mask for signatures already in HBM
    (EN:133-145 / ZH-2..ZH-4).  Document index = row of `sig`.  min_jaccard (default None = the reference's behaviour)
    drops candidate edges (document, smallest document of its bucket) whose estimated Jaccard similarity is lower."""
    n, num_perm = sig.shape
    mark = timer.mark if timer is not None else (lambda name: None)
    key, meta = be.band_keys(sig, bands, rows, 0, key_bits)
    mark("band_keys")
    eu, ev, n_coll, rb = be.bucket_edges(key, meta, sig, bands, num_perm, rows, rows, want_buckets)
    mark("bucket_edges")
    if min_jaccard is not None:
        eu, ev = verified_edges(be, sig, eu, ev, min_jaccard)
        mark("verify")
    labels = be.components_init(n)
    be.components_union(labels, eu, ev)
    be.components_finish(labels)
    removed, n_removed = be.removed_mask(labels, 0, n)
    mark("components")
    return {"sig": sig, "key": key, "meta": meta, "edge_u": eu, "edge_v": ev, "labels": labels,
            "removed": removed, "n_removed": n_removed, "n_collisions": int(n_coll),
            "rec_bucket": rb}


def dedup_local(be, text: torch.Tensor, offsets: torch.Tensor, num_perm: int = NUM_PERM, width: int = WIDTH,
                unit: str = "char", bands: int = BANDS, rows: int = ROWS, seed: int = SEED,
                text_bytes: Optional[int] = None, key_bits: int = 64, want_buckets: bool = False,
                timer: Optional[StageTimer] = None, min_jaccard: Optional[float] = None):
    """One-GPU path.  Returns a dict of device tensors: sig [n,K], edge_u/edge_v, labels [n]
    (label = smallest doc index of the component), removed [n] uint8, n_removed (device
    int64[1]), n_collisions (host int)."""
    mark = timer.mark if timer is not None else (lambda name: None)
    mark("start")
    sig = be.signatures(text, offsets, num_perm, width, unit, seed, text_bytes=text_bytes)
    mark("signatures")
    return cluster_local(be, sig, bands, rows, key_bits, want_buckets, timer, min_jaccard)


def signatures_streamed(be, blocks, n_docs: int, num_perm=NUM_PERM, width: int = WIDTH, unit: str = "char", seed: int = SEED,
                        on_block=None):
    """Signatures of a corpus that does not (have to) fit in HBM as text: `blocks` yields (text, offsets, text_bytes) of
    consecutive document ranges (packed device tensors, as for Engine.signatures); every block is hashed into its rows of
    ONE resident signature matrix and may then be dropped -- at 100 M documents the text is ~220 GB, the signatures 51 GB.
    num_perm may be a tuple: several matrices from one pass over the text (band sweeps with different K).
    on_block(first_doc, text, offsets, text_bytes, ms) is called after each block (ms: kernel time by events).
    Returns the int32 matrix [n_docs, K] (a dict K -> matrix when num_perm is a tuple)."""
    many = isinstance(num_perm, (tuple, list))
    Ks = list(num_perm) if many else [num_perm]
    sigs = {K: torch.empty((n_docs, K), dtype=torch.int32, device=be.device) for K in Ks}
    d0 = 0
    for text, offs, nbytes in blocks:
        nd = offs.numel() - 1
        if d0 + nd > n_docs:
            raise ValueError("signatures_streamed: the blocks hold more than n_docs documents")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for K in Ks:
            be.signatures(text, offs, K, width, unit, seed, text_bytes=nbytes, out=sigs[K][d0:d0 + nd])
        e1.record()
        if on_block is not None:
            e1.synchronize()
            on_block(d0, text, offs, nbytes, e0.elapsed_time(e1))
        d0 += nd
    if d0 != n_docs:
        raise ValueError("signatures_streamed: the blocks hold %d documents, expected %d" % (d0, n_docs))
    return sigs if many else sigs[Ks[0]]


# RCCL on this stack silently drops the tail of an all_to_all_single message beyond 1 GiB (measured with one rank:
# everything past the first half of a 1 GiB + 8 KiB message arrives untouched, tools/diag_a2a.py), so no single
# collective call here moves more than COLLECTIVE_BYTES (half the observed limit) in total; larger exchanges run in rounds.
COLLECTIVE_BYTES = 512 << 20


def _host_staged(t: torch.Tensor, group=None) -> bool:
    """gloo carries CUDA tensors through the host anyway and was seen to dead-lock with several ranks on one GPU
    (rehearsals of the multi-rank path on a one-GPU box): with that backend the exchanges below are staged
    explicitly through host memory.  RCCL (backend "nccl") always works on the device buffers."""
    import os
    import torch.distributed as dist
    if os.environ.get("MDD_GLOO_DEVICE_TENSORS") == "1":      # diagnosis only (tools/gloo_probe.py)
        return False
    return t.is_cuda and dist.get_backend(group) == "gloo"


def _all_to_all_v(t: torch.Tensor, send_counts, recv_counts, width: int, group=None, limit_bytes: Optional[int] = None):
    """Variable all-to-all of records of `width` elements: t holds the records for peer 0, peer 1, ... back to back
    (send_counts records each); returns what the peers sent here, grouped by peer (recv_counts)."""
    import torch.distributed as dist
    if _host_staged(t, group):
        return _all_to_all_v(t.cpu(), send_counts, recv_counts, width, group, limit_bytes).to(t.device)
    world = len(send_counts)
    send_counts = [int(c) for c in send_counts]
    recv_counts = [int(c) for c in recv_counts]
    out = torch.empty(sum(recv_counts) * width, dtype=t.dtype, device=t.device)
    per_peer = max(1, (limit_bytes or COLLECTIVE_BYTES) // (t.element_size() * width * world))   # records per peer per round
    mx = torch.tensor([max(send_counts + recv_counts + [0])], dtype=torch.int64, device=t.device)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)       # every rank must run the same number of rounds
    rounds = max(1, -(-int(mx.item()) // per_peer))
    if rounds == 1:
        dist.all_to_all_single(out, t, [c * width for c in recv_counts], [c * width for c in send_counts], group=group)
        return out
    s_off = [0] * world
    r_off = [0] * world
    for p in range(1, world):
        s_off[p] = s_off[p - 1] + send_counts[p - 1]
        r_off[p] = r_off[p - 1] + recv_counts[p - 1]
    for k in range(rounds):
        s_cnt = [min(max(c - k * per_peer, 0), per_peer) for c in send_counts]
        r_cnt = [min(max(c - k * per_peer, 0), per_peer) for c in recv_counts]
        sbuf = torch.cat([t[(s_off[p] + k * per_peer) * width: (s_off[p] + k * per_peer + s_cnt[p]) * width] for p in range(world)])
        rbuf = torch.empty(sum(r_cnt) * width, dtype=t.dtype, device=t.device)
        dist.all_to_all_single(rbuf, sbuf, [c * width for c in r_cnt], [c * width for c in s_cnt], group=group)
        pos = 0
        for p in range(world):
            out[(r_off[p] + k * per_peer) * width: (r_off[p] + k * per_peer + r_cnt[p]) * width] = rbuf[pos: pos + r_cnt[p] * width]
            pos += r_cnt[p] * width
    return out


def _all_gather_v(t: torch.Tensor, group=None, limit_bytes: Optional[int] = None):
    """All-gather of variable-length 1-D tensors (sizes exchanged first), in rounds of bounded size."""
    import torch.distributed as dist
    if _host_staged(t, group):
        return _all_gather_v(t.cpu(), group, limit_bytes).to(t.device)
    world = dist.get_world_size(group)
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    per_rank = max(1, (limit_bytes or COLLECTIVE_BYTES) // (t.element_size() * world))
    parts = [[] for _ in range(world)]
    for lo in range(0, max(max(sizes), 1), per_rank):
        cnt = [min(max(s - lo, 0), per_rank) for s in sizes]
        mx = max(max(cnt), 1)
        pad = torch.zeros(mx, dtype=t.dtype, device=t.device)
        mine = t[lo: lo + per_rank]
        pad[: mine.numel()] = mine
        outs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(outs, pad, group=group)
        for p in range(world):
            parts[p].append(outs[p][: cnt[p]])
    return torch.cat([x for p in range(world) for x in parts[p]])


class TorchComm:
    """The exchanges of the multi-GPU path over torch.distributed.  This is the TEST / rehearsal transport: gloo on CPU
    tensors (tests/test_distributed_cpu.py) and gloo with every rank on one GPU (rehearsals on a one-GPU box).  The
    product transport is NativeComm (collectives issued by libmdd_hip.so against RCCL)."""

    name = "torch.distributed"

    def __init__(self, group=None):
        self.group = group

    def exchange(self, tuples_s: torch.Tensor, meta_s: torch.Tensor, send_counts: torch.Tensor, rows: int):
        import torch.distributed as dist
        send = [int(c) for c in send_counts.tolist()]
        sc = torch.tensor(send, dtype=torch.int64, device=meta_s.device)
        rc = torch.empty_like(sc)
        if _host_staged(sc, self.group):
            sc_h, rc_h = sc.cpu(), rc.cpu()
            dist.all_to_all_single(rc_h, sc_h, group=self.group)
            rc = rc_h
        else:
            dist.all_to_all_single(rc, sc, group=self.group)
        recv = [int(c) for c in rc.tolist()]
        tuples_r = _all_to_all_v(tuples_s, send, recv, rows, self.group)
        meta_r = _all_to_all_v(meta_s, send, recv, 1, self.group)
        return tuples_r, meta_r, send, recv

    def allgather_edges(self, eu: torch.Tensor, ev: torch.Tensor):
        return _all_gather_v(eu, self.group), _all_gather_v(ev, self.group)


class CommInitError(RuntimeError):
    """mdd_comm_init did not succeed on every rank"""


class NativeComm:
    """The exchanges issued by libmdd_hip.so itself (include/mdd.h: mdd_comm_init, mdd_exchange_counts / _bands,
    mdd_allgather_counts / _edges) against RCCL, on the engine's stream; torch only owns the buffers.  The
    communicator's id travels from rank 0 to the others through the torch.distributed process group that launched
    the job (object broadcast: plumbing).  One host synchronisation per exchange (the counts), inside the library."""

    name = "libmdd_hip + RCCL"

    def __init__(self, engine, group=None):
        import torch.distributed as dist
        self.eng = engine
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        box = [engine.comm_unique_id() if self.rank == 0 else None]
        if self.world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        # mdd_comm_init is collective, but a rank can still fail on its own after ncclCommInitRank (staging allocation):
        # agree on the outcome, so that either every rank uses the native transport or none does
        ok, why = 1, ""
        try:
            engine.comm_init(box[0], self.rank, self.world)
        except Exception as e:      # noqa: BLE001 -- the other ranks must hear of it whatever it was
            ok, why = 0, repr(e)
        if self.world > 1:
            flag = torch.tensor([ok], dtype=torch.int32, device=engine.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            all_ok = int(flag.item()) == 1
        else:
            all_ok = ok == 1
        if not all_ok:
            if ok:
                engine.comm_destroy()
            raise CommInitError(why or "mdd_comm_init failed on another rank")

    @classmethod
    def from_unique_id(cls, engine, unique_id: bytes, rank: int, world: int):
        """For launchers without torch.distributed (one thread or process per GPU that hands rank 0's
        Engine.comm_unique_id() to the others by its own means): collective, every rank calls it."""
        self = cls.__new__(cls)
        self.eng, self.rank, self.world = engine, int(rank), int(world)
        engine.comm_init(unique_id, self.rank, self.world)
        return self

    def exchange(self, tuples_s, meta_s, send_counts, rows):
        return self.eng.exchange_bands(tuples_s, meta_s, send_counts, rows)

    def allgather_edges(self, eu, ev):
        return self.eng.allgather_edges(eu, ev)


def default_comm(be, group=None):
    """NativeComm for the HIP engine under the RCCL ("nccl") backend, TorchComm otherwise (gloo: CPU tests, one-GPU
    rehearsals; MDD_COMM=torch forces it).  Cached on the engine: a context has one communicator."""
    import os
    import torch.distributed as dist
    native = (hasattr(be, "comm_init") and dist.get_backend(group) == "nccl" and os.environ.get("MDD_COMM", "native") != "torch")
    if not native:
        return TorchComm(group)
    comm = getattr(be, "_native_comm", None)
    if comm is None:
        # RCCL must be loadable on EVERY rank before any rank enters ncclCommInitRank (a rank that cannot load it would
        # leave the others waiting there for ever): probe locally, agree, and only then initialise.  Without agreement
        # all ranks take torch.distributed's RCCL communicator instead -- loudly; it is the same library and the same
        # wire traffic, issued from Python.
        ok, why = 1, ""
        try:
            be.comm_unique_id()
        except Exception as e:      # noqa: BLE001 -- whatever went wrong, the other ranks must hear of it
            ok, why = 0, repr(e)
        flag = torch.tensor([ok], dtype=torch.int32, device=be.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) == 0:
            import sys
            print("map_neo_amd: libmdd_hip could not bind RCCL on every rank (%s): the exchanges go through "
                  "torch.distributed on this run" % (why or "another rank failed"), file=sys.stderr, flush=True)
            comm = TorchComm(group)
        else:
            try:
                comm = NativeComm(be, group)
            except CommInitError as e:
                import sys
                print("map_neo_amd: mdd_comm_init did not succeed on every rank (%s): the exchanges go through "
                      "torch.distributed on this run" % e, file=sys.stderr, flush=True)
                comm = TorchComm(group)
        be._native_comm = comm
    return comm


def dedup_distributed(be, text: torch.Tensor, offsets: torch.Tensor, doc_base: int, n_global: int,
                      num_perm: int = NUM_PERM, width: int = WIDTH, unit: str = "char", bands: int = BANDS,
                      rows: int = ROWS, seed: int = SEED, text_bytes: Optional[int] = None, key_bits: int = 64,
                      group=None, timings: Optional[dict] = None, timer: Optional[StageTimer] = None, comm=None):
    """Multi-GPU path; every rank calls it with its own shard (documents doc_base ..
    doc_base+n_local of n_global).  Returns the same dict as dedup_local; labels cover all
    n_global documents (replicated), removed covers the local shard."""
    if timer is not None:
        timer.mark("start")
    sig = be.signatures(text, offsets, num_perm, width, unit, seed, text_bytes=text_bytes)
    if timer is not None:
        timer.mark("signatures")
    return cluster_distributed(be, sig, doc_base, n_global, bands, rows, key_bits, group, timings, timer, comm)


def cluster_distributed(be, sig: torch.Tensor, doc_base: int, n_global: int, bands: int = BANDS, rows: int = ROWS,
                        key_bits: int = 64, group=None, timings: Optional[dict] = None, timer: Optional[StageTimer] = None,
                        comm=None):
    """Multi-GPU clustering of signatures already in HBM (row i of `sig` = global document doc_base + i):
    band records -> all-to-all by bucket owner -> local buckets -> spanning forest -> all-gather ->
    replicated union-find.  Same return value as dedup_distributed.  `comm` carries the two exchanges
    (default_comm: libmdd_hip + RCCL for the HIP engine, torch.distributed for tests)."""
    if comm is None:
        comm = default_comm(be, group)
    if getattr(comm, "world", None) is not None:      # NativeComm knows its communicator (and may run without torch.distributed)
        world, rank = comm.world, comm.rank
    else:
        import torch.distributed as dist
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    n_local = sig.shape[0]
    mark = timer.mark if timer is not None else (lambda name: None)
    key, meta = be.band_keys(sig, bands, rows, doc_base, key_bits)
    mark("band_keys")
    # --- route band records to their owner rank (all-to-all over xGMI); the counts stay on the device
    tuples_s, meta_s, send_counts = be.route(sig, key, meta, bands, rows, world)
    mark("route")
    tuples_r, meta_r, send_list, recv_list = comm.exchange(tuples_s, meta_s, send_counts, rows)
    mark("all_to_all")
    del tuples_s, meta_s
    # --- local buckets over the records this rank owns
    key_r = be.packed_keys(tuples_r, meta_r, rows, key_bits)
    eu, ev, n_coll, _ = be.bucket_edges(key_r, meta_r, tuples_r, 1, rows, 0, rows, False)
    mark("bucket_edges")
    # --- merge: reduce this rank's edges to a spanning forest (at most one edge per touched
    # document), all-gather the forests, identical union-find everywhere
    n_local_edges = int(eu.numel())
    if world > 1 and n_local_edges:
        lab_local = be.components_init(n_global)
        be.components_union(lab_local, eu, ev)
        be.components_finish(lab_local)
        eu, ev = be.forest_edges(lab_local, n_local_edges)
        del lab_local
    mark("forest")
    eu_all, ev_all = comm.allgather_edges(eu, ev)
    mark("all_gather")
    labels = be.components_init(n_global)
    be.components_union(labels, eu_all, ev_all)
    be.components_finish(labels)
    removed, n_removed = be.removed_mask(labels, doc_base, n_local)
    mark("components")
    if timings is not None:
        rec_bytes = 4 * rows + 8
        timings["transport"] = comm.name
        timings["records_sent"] = int(sum(send_list))
        timings["records_received"] = int(sum(recv_list))
        timings["records_sent_to_other_ranks"] = int(sum(send_list) - send_list[rank])
        timings["all_to_all_bytes_sent"] = int(sum(send_list)) * rec_bytes
        timings["all_to_all_bytes_received"] = int(sum(recv_list)) * rec_bytes
        timings["record_bytes"] = rec_bytes
        timings["edges_local"] = n_local_edges
        timings["forest_edges_local"] = int(eu.numel())
        timings["edges_global"] = int(eu_all.numel())
        timings["all_gather_bytes_received"] = int(eu_all.numel()) * 8
    return {"sig": sig, "edge_u": eu_all, "edge_v": ev_all, "labels": labels, "removed": removed,
            "n_removed": n_removed, "n_collisions": int(n_coll)}
