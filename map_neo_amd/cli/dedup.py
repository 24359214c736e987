#!/usr/bin/env python3
"""Stage entry point with the I/O contract of BASELINE.json's north_star: jsonl / jsonl.gz /
parquet shards in, kept-document ids and cluster assignments out.

    python -m map_neo_amd.cli.dedup --input_dir I --output_dir O \
        [--id_field doc_id] [--content_field raw_content] [--normalize en|none] [--write_shards]

Outputs in O:
  kept_ids.txt / removed_ids.txt   one document id per line
  clusters.jsonl                   {"doc_id": ..., "cluster": <id of the surviving member>} for every
                                   document of a cluster with more than one member
  summary.json                     counts and timings
  shards/<name>.jsonl.gz           (with --write_shards) the surviving rows of every input shard
Documents without an id field get "<file>@<row>" (the ZH convention, ZH-1:45).  Rows whose
content is null / missing / not utf-8 encodable are never hashed and always kept (EN:124,
ZH-1:52-54).  --normalize en applies EN:31-40 before hashing (the EN pipeline), none hashes
the raw field (the ZH pipeline).  The survivor of a cluster is its smallest id when ids come
from the data (EN:145) and its first document in input order otherwise.
LSH parameters default to the reference's constants (EN:17-20 / ZH-1:14-17).
Multi-GPU: `torchrun --nproc-per-node N -m map_neo_amd.cli.dedup ...` -- one process per GPU, each ingests a
contiguous block of the shards; same outputs.
"""
from __future__ import annotations

import argparse
import gzip
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser()
    p.add_argument("--input_dir", required=True)
    p.add_argument("--output_dir", required=True)
    p.add_argument("--id_field", default="doc_id")
    p.add_argument("--content_field", default="raw_content")
    p.add_argument("--normalize", choices=["en", "none"], default="en")
    p.add_argument("--write_shards", action="store_true")
    p.add_argument("--distributed", action="store_true",
                   help="use the multi-GPU path even with one rank (implied by WORLD_SIZE > 1 under torchrun)")
    p.add_argument("--width", type=int, default=13)
    p.add_argument("--ngram_unit", choices=["char", "word"], default="char")
    p.add_argument("--num_perm", type=int, default=128)
    p.add_argument("--bands", type=int, default=9)
    p.add_argument("--rows", type=int, default=13)
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--min_jaccard", type=float, default=None,
                   help="extension (the reference has no such step): drop candidate pairs whose signatures agree on fewer than "
                        "ceil(min_jaccard * num_perm) components before the union-find; one-GPU path only.  Only the STAR edges of a bucket "
                        "are checked (each document against the smallest document of its bucket), not all pairs: two documents "
                        "that are similar to each other but not to that document are split, unlike pairwise LSH verification")
    p.add_argument("--device", type=int, default=0)
    p.add_argument("--workers", type=int, default=min(16, os.cpu_count() or 1))
    p.add_argument("--checkpoint_dir", default=None,
                   help="write signatures / edges / labels + manifest.json there (map_neo_amd/checkpoint.py; single process)")
    p.add_argument("--resume_from", choices=["signatures", "edges", "labels"], default=None,
                   help="start from that artifact of --checkpoint_dir instead of parsing and hashing the input")
    return p


def _parse_shard(task):
    """pool worker: one shard -> ids of all rows, candidate rows, packed text ready to hash and (Parquet,
    --normalize en) the pure-ASCII documents that the device still has to normalise (hostio)."""
    path, id_field, content_field, normalize = task
    from .. import hostio
    name = os.path.basename(path)
    if path.lower().endswith((".parquet", ".pq")):
        r = hostio.read_parquet_shard(path, content_field, id_field, normalize_en=(normalize == "en"))
        row_of = None                                 # row k is row k
    else:
        j = hostio.read_jsonl_shard(path, content_field, id_field, normalize_en=(normalize == "en"), on_error="skip")
        r = {"ids": j["ids"], "cand": j["cand"], "text": j["text"], "offsets": j["offsets"],
             "ready": np.arange(len(j["cand"]), dtype=np.int64), "raw_text": np.zeros(0, np.uint8),
             "raw_offsets": np.zeros(1, np.uint64), "raw": np.zeros(0, np.int64)}
        row_of = j["line_no"].tolist()
    r["synthetic"] = False
    if None in r["ids"]:                              # (a scan in C: nearly every shard has all its ids)
        for k, rid in enumerate(r["ids"]):
            if rid is None:
                r["ids"][k], r["synthetic"] = f"{name}@{k if row_of is None else int(row_of[k])}", True
    return r


def _loads_or_none(line):
    try:
        return json.loads(line)
    except ValueError:
        return None


def _write_shard(task):
    path, out_path, id_field, remove_ids = task
    from .. import hostio
    name = os.path.basename(path)
    kept = 0
    is_pq = path.lower().endswith((".parquet", ".pq"))
    rows = enumerate(hostio.iter_records(path)) if is_pq else \
        ((k, _loads_or_none(line)) for k, line in enumerate(hostio.iter_lines(path)) if line.strip())
    with gzip.open(out_path, "wt", encoding="utf8", compresslevel=6) as f:
        for k, row in rows:
            if not isinstance(row, dict):
                continue
            rid = row.get(id_field)
            if rid is None:
                rid = f"{name}@{k}"                       # k = physical line number (jsonl) / row number (parquet)
            if rid not in remove_ids:
                f.write(json.dumps(row, ensure_ascii=False, default=str) + "\n")
                kept += 1
    return kept


def run(args, engine_factory=None) -> dict:
    """Single process: one GPU.  Under torchrun (WORLD_SIZE > 1, or --distributed): one process per GPU, every rank
    ingests a contiguous block of the shards, band records are exchanged by all-to-all, the union-find is
    replicated (map_neo_amd.dedup.cluster_distributed), every rank writes the outputs of its own shards and rank 0
    stitches the id / cluster lists together.  `engine_factory(device)` lets tests inject a CPU backend."""
    from .. import hostio
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or getattr(args, "distributed", False)
    all_files = hostio.list_files(args.input_dir)
    f_lo, f_hi = (len(all_files) * rank) // world, (len(all_files) * (rank + 1)) // world
    files = all_files[f_lo:f_hi]
    if rank == 0:
        os.makedirs(args.output_dir, exist_ok=True)
    t0 = time.time()
    n_workers = max(1, args.workers // world)
    ckpt_dir, resume = getattr(args, "checkpoint_dir", None), getattr(args, "resume_from", None)
    if (ckpt_dir or resume) and distributed:           # flags are validated BEFORE the worker pool exists
        raise SystemExit("--checkpoint_dir / --resume_from: single-process runs only")
    if resume and not ckpt_dir:
        raise SystemExit("--resume_from needs --checkpoint_dir")
    if files and all(f.lower().endswith((".parquet", ".pq")) for f in files):
        # Arrow decodes without the GIL and hands its buffers over without a copy: threads, not processes
        from multiprocessing.pool import ThreadPool
        pool = ThreadPool(n_workers)
    else:
        pool = mp.get_context("fork").Pool(n_workers)                      # forked before this process touches the GPU
    try:
        parsed = pool.imap(_parse_shard, [(p, args.id_field, args.content_field, args.normalize) for p in ([] if resume else files)])
        import torch
        from .. import checkpoint as ckpt

        from ..dedup import cluster_distributed, cluster_local
        from ..minhash import as_u32, to_padded_device
        backend = os.environ.get("MDD_DIST_BACKEND", "nccl")     # "gloo": rehearsal with every rank on --device (one-GPU box)
        if engine_factory is not None:
            eng = engine_factory(local_rank)
        else:
            from ..minhash import Engine
            eng = Engine(local_rank if (distributed and backend == "nccl") else args.device)
        dist = None
        if distributed:
            import torch.distributed as dist
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29533")
                os.environ.setdefault("RANK", "0")
                os.environ.setdefault("WORLD_SIZE", "1")
                if eng.device.type == "cuda" and backend == "nccl":
                    dist.init_process_group("nccl", device_id=eng.device)
                else:
                    dist.init_process_group("gloo")
        sigs, all_ids, cand_ids, any_synth = [], [], [], False

        def padded(text):
            return to_padded_device(eng, text)

        for r in parsed:
            ids, cand = r["ids"], r["cand"]
            any_synth |= r["synthetic"]
            all_ids.extend(ids)
            if len(cand) == 0:
                continue
            sig = torch.empty((len(cand), args.num_perm), dtype=torch.int32, device=eng.device)
            if len(r["ready"]):
                s_ready = eng.signatures(padded(r["text"]), eng.to_device(r["offsets"]), args.num_perm, args.width,
                                         args.ngram_unit, args.seed, text_bytes=int(r["text"].size))
                sig[eng.to_device(r["ready"])] = s_ready
            if len(r["raw"]):                        # EN:31-40 on the device (flagged documents: Python steps)
                s_raw, _ = eng.signatures_en(padded(r["raw_text"]), eng.to_device(r["raw_offsets"]), args.num_perm,
                                             args.width, args.ngram_unit, args.seed)
                sig[eng.to_device(r["raw"])] = s_raw
            sigs.append(sig)
            cand_ids.extend(ids if len(cand) == len(ids) else [ids[i] for i in cand.tolist()])
        t1 = time.time()
        state = None
        if resume:
            state = ckpt.load(ckpt_dir, resume, ckpt.params_of(args), args.input_dir)
            cand_ids = state["doc_ids"]
            all_ids, any_synth = state["manifest"]["extra"]["all_ids"], state["manifest"]["extra"]["any_synth"]
        n = len(cand_ids)
        if resume:
            sig = eng.to_device(state["signatures"].reshape(-1)).view(n, args.num_perm) if resume == "signatures" else None
        else:
            sig = torch.cat(sigs) if sigs else torch.empty((0, args.num_perm), dtype=torch.int32, device=eng.device)
        ck_extra = {"all_ids": all_ids, "any_synth": any_synth, "input_dir": os.path.abspath(args.input_dir)}
        if ckpt_dir and not resume:
            ckpt.save(ckpt_dir, ckpt.params_of(args), cand_ids, signatures=as_u32(sig).reshape(n, args.num_perm), extra=ck_extra)
        if not distributed:
            doc_base, n_global = 0, n
            if resume == "labels":
                labels = state["labels"]
            elif resume == "edges":
                e = state["edges"]
                lab = eng.components_init(n)
                if len(e):
                    eng.components_union(lab, eng.to_device(np.ascontiguousarray(e[:, 0])), eng.to_device(np.ascontiguousarray(e[:, 1])))
                eng.components_finish(lab)
                labels = as_u32(lab).reshape(n)
            else:
                out = cluster_local(eng, sig, args.bands, args.rows, min_jaccard=args.min_jaccard)
                labels = as_u32(out["labels"]).reshape(n)
                if ckpt_dir:
                    edges = np.ascontiguousarray(np.stack([as_u32(out["edge_u"]), as_u32(out["edge_v"])], axis=1))
                    rewrite = resume == "signatures" and state["manifest"]["params"] != ckpt.params_of(args)
                    ckpt.save(ckpt_dir, ckpt.params_of(args), cand_ids, signatures=state["signatures"] if rewrite else None,
                              edges=edges, labels=labels, extra=ck_extra)
        else:
            counts = [None] * world
            dist.all_gather_object(counts, (n, any_synth))
            doc_base, n_global = sum(c for c, _ in counts[:rank]), sum(c for c, _ in counts)
            any_synth = any(s for _, s in counts)
            labels = as_u32(cluster_distributed(eng, sig, doc_base, n_global, args.bands, args.rows)["labels"])
        # documents of clusters with more than one member (global indices), and this rank's share of them
        gidx = np.arange(n_global, dtype=np.uint32)
        in_cluster = labels != gidx
        in_cluster[labels[in_cluster]] = True
        mine = np.flatnonzero(in_cluster[doc_base: doc_base + n])
        mine_l = mine.tolist()
        pairs = list(zip(labels[doc_base + mine].tolist(), [cand_ids[k] for k in mine_l], (mine + doc_base).tolist()))
        if distributed:
            # in rounds of 200 k entries per rank: all_gather_object pickles into one message per call, and no
            # collective call here should come near 1 GiB (map_neo_amd.dedup.COLLECTIVE_BYTES)
            chunk = 200_000
            lens = [None] * world
            dist.all_gather_object(lens, len(pairs))
            all_parts = [[] for _ in range(world)]
            for lo in range(0, max(max(lens), 1), chunk):
                gathered = [None] * world
                dist.all_gather_object(gathered, pairs[lo: lo + chunk])
                for r, part in enumerate(gathered):
                    all_parts[r].extend(part)
            all_pairs = [p for part in all_parts for p in part]
        else:
            all_pairs = pairs
        # representative: smallest id (EN:145) unless ids were synthesised, then first in input order -- all_pairs is in
        # global document order (ranks in order, each rank's documents in order), so that is the first one met
        rep = {}
        if any_synth:
            for l, d, _ in all_pairs:
                if l not in rep:
                    rep[l] = d
        else:
            get = rep.get
            for l, d, _ in all_pairs:
                r = get(l)
                if r is None or d < r:
                    rep[l] = d
        remove_ids = {d for l, d, _ in all_pairs if d != rep[l]}
        t2 = time.time()
        suffix = ".part%05d" % rank if distributed else ""
        # the id lists are 10^5 ... 10^6 entries per rank: one encoder, one join and one write per file (json.dumps with a
        # keyword builds an encoder per call, an f-string per id is a Python-level loop)
        enc = json.JSONEncoder(ensure_ascii=False).encode          # == json.dumps(x, ensure_ascii=False)
        with open(os.path.join(args.output_dir, "clusters.jsonl" + suffix), "w", encoding="utf8") as f:
            f.write("".join(['{"doc_id": %s, "cluster": %s}\n' % (enc(d), enc(rep[l])) for l, d, _ in pairs]))
        with open(os.path.join(args.output_dir, "removed_ids.txt" + suffix), "w", encoding="utf8") as fr, \
                open(os.path.join(args.output_dir, "kept_ids.txt" + suffix), "w", encoding="utf8") as fk:
            removed = [d for d in all_ids if d in remove_ids]
            kept = [d for d in all_ids if d not in remove_ids] if removed else all_ids
            for fh, lst in ((fr, removed), (fk, kept)):
                if lst:
                    fh.write("\n".join(map(str, lst)) + "\n")
        kept_rows = None
        if args.write_shards:
            os.makedirs(os.path.join(args.output_dir, "shards"), exist_ok=True)
            # names are derived from ALL input files (every rank computes the same list), so they are unique across ranks
            out_names = hostio.shard_output_names(all_files)[f_lo:f_hi]
            tasks = [(p, os.path.join(args.output_dir, "shards", nm), args.id_field, remove_ids) for p, nm in zip(files, out_names)]
            kept_rows = sum(pool.imap_unordered(_write_shard, tasks))
        t3 = time.time()
    finally:
        pool.close()
        pool.join()
    summary = {"n_rows": len(all_ids), "n_hashed": n, "n_removed_ids": len(remove_ids),
               "n_clusters": len(rep), "n_clustered_docs": len(all_pairs), "kept_rows_written": kept_rows,
               "t_parse_signatures_s": t1 - t0, "t_cluster_s": t2 - t1, "t_write_s": t3 - t2, "world": world}
    if distributed:
        parts = [None] * world
        dist.all_gather_object(parts, (len(all_ids), n, kept_rows))            # also the barrier before stitching
        summary["n_rows"], summary["n_hashed"] = sum(p[0] for p in parts), sum(p[1] for p in parts)
        summary["kept_rows_written"] = sum(p[2] for p in parts) if args.write_shards else None
        if rank == 0:
            for name in ("clusters.jsonl", "removed_ids.txt", "kept_ids.txt"):
                with open(os.path.join(args.output_dir, name), "wb") as out:
                    for r in range(world):
                        part = os.path.join(args.output_dir, name + ".part%05d" % r)
                        with open(part, "rb") as f:
                            out.write(f.read())
                        os.remove(part)
        dist.barrier()
    if rank == 0:
        with open(os.path.join(args.output_dir, "summary.json"), "w") as f:
            json.dump(summary, f)
        print(json.dumps(summary), file=sys.stderr)
    return summary


def main(argv=None):
    run(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
