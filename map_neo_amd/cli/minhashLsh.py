#!/usr/bin/env python3
"""Drop-in for Matrix/english_data_clean_pipeline/dedup/Minhash_Lsh_Deduplication/minhashLsh.py
(EN in SURVEY.md): same flags, same outputs, no Spark.

    python -m map_neo_amd.cli.minhashLsh --input_dir I --inter_dir M --output_dir O \
        [--content_field_name raw_content]

Input : directory of JSON-lines files (optionally .gz) with `doc_id` and the content field
        (EN:120-123).
Inter : signatures keyed by doc_id (EN:128-129 writes a Spark pickle SequenceFile of
        (uint64[128], doc_id); here: signatures.npy (uint32 [n, K]) + doc_ids.json, rows in
        input order).
Output: gzip JSON lines of the surviving rows with all original columns (EN:149-150), one
        part file per input file, and the line "the number of doc need to be remove is  N"
        (EN:148).
Rows whose content is null are never hashed and therefore always kept (EN:124).  Within a
connected component the row(s) carrying the smallest doc_id (Python ordering) survive
(EN:145: remove_list = first element of every final (v, min) edge).

Two streaming passes, so memory holds signatures and doc_ids but never the corpus:
  pass 1  a process pool parses + normalises one file per task (host work, EN:31-40) and
          returns packed utf-8; the main process feeds the GPU file by file (signatures stay
          in HBM);   then bands -> buckets -> union-find on the GPU;
  pass 2  the pool re-reads every file and writes the rows whose doc_id survives.
Extension flags (defaults = the reference's constants EN:17-20): --width --ngram_unit
--num_perm --bands --rows --seed --device --workers.

Checkpoints (the reference persists its signatures at EN:128-129 but never reloads them): inter_dir
also receives edges.npy, labels.npy and a manifest.json naming every artifact (map_neo_amd/checkpoint.py), and
--resume_from signatures|edges|labels restarts from that artifact instead of re-hashing the corpus; the outputs are
identical to an uninterrupted run.
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
    p.add_argument("--input_dir", help="input directories")
    p.add_argument("--inter_dir", help="output directory")
    p.add_argument("--output_dir", help="output directory")
    p.add_argument("--content_field_name", help="content field name", default="raw_content")
    p.add_argument("--width", type=int, default=13)
    p.add_argument("--ngram_unit", choices=["char", "word"], default="char")
    p.add_argument("--num_perm", type=int, default=128)
    p.add_argument("--bands", type=int, default=9)
    p.add_argument("--rows", type=int, default=13)
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--device", type=int, default=0)
    p.add_argument("--workers", type=int, default=min(16, os.cpu_count() or 1))
    p.add_argument("--resume_from", choices=["signatures", "edges", "labels"], default=None,
                   help="start from an artifact of a previous run in --inter_dir (manifest.json) instead of hashing the input")
    p.add_argument("--reader", choices=["auto", "threads", "processes"], default="processes",
                   help="who parses the shards in pass 1.  threads: no pickling of the packed text (about 20 %% faster on "
                        "ASCII corpora with libmdd_io.so), but lines that fall back to the Python steps (non-ASCII text "
                        "under EN normalisation) then serialise on the GIL; auto = threads when libmdd_io.so is available")
    p.add_argument("--normalize_on", choices=["auto", "device", "host"], default="auto",
                   help="where normalize_text (EN:31-40) runs.  device: the workers hand over the raw content and the GPU "
                        "normalises it (non-ASCII text included; the few documents it cannot vouch for take the Python steps); "
                        "host: the workers normalise (native code for ASCII lines, the Python steps for every non-ASCII line); "
                        "auto = device unless ftfy is installed (ftfy rewrites text in ways only Python can)")
    return p


def _parse_file(task):
    """pool worker, pass 1: (doc_ids of all rows, indices of candidate rows, packed text, offsets).
    Native scanner (+ ASCII normaliser when the host normalises) where it is sure, the Python steps elsewhere (hostio)."""
    path, field, host_norm = task
    from .. import hostio
    r = hostio.read_jsonl_shard(path, field, "doc_id", normalize_en=host_norm, on_error="skip")
    if r["n_bad"]:
        print(f"{os.path.basename(path)}: {r['n_bad']} lines are not JSON objects and were skipped", file=sys.stderr)
    return path, r["ids"], r["cand"], r["text"], r["offsets"]


def _write_file(task):
    """pool worker, pass 2: copy the rows whose doc_id is not removed (left_anti join, EN:149)"""
    path, out_path, remove_ids = task
    from .. import hostio
    kept = 0
    with gzip.open(out_path, "wt", encoding="utf8", compresslevel=6) as f:      # zlib's default level, as Spark's gzip codec
        for line in hostio.iter_lines(path):
            if not line.strip():
                continue
            try:
                row = json.loads(line)
            except ValueError:
                continue
            if not isinstance(row, dict):
                continue
            if row.get("doc_id") not in remove_ids:
                f.write(json.dumps(row, ensure_ascii=False) + "\n")
                kept += 1
    return kept


def edges_of(eng, sig, bands: int, rows: int) -> np.ndarray:
    """EN:133-136 on signatures in HBM: bucket collisions as document index pairs, uint32 [m, 2]"""
    from ..dedup import cluster_local
    from ..minhash import as_u32
    out = cluster_local(eng, sig, bands, rows)
    return np.ascontiguousarray(np.stack([as_u32(out["edge_u"]), as_u32(out["edge_v"])], axis=1))


def labels_of(eng, edges: np.ndarray, cand_ids) -> np.ndarray:
    """EN:138-144: component label (smallest member index) per candidate row.  Rows sharing one doc_id are ONE node
    of the reference's graph (its edges carry doc_ids), so they are linked first."""
    from ..minhash import as_u32
    n = len(cand_ids)
    first_of, extra = {}, []
    for k, d in enumerate(cand_ids):
        j = first_of.setdefault(d, k)
        if j != k:
            extra.append((k, j))
    lab = eng.components_init(n)
    if len(edges):
        eng.components_union(lab, eng.to_device(np.ascontiguousarray(edges[:, 0])), eng.to_device(np.ascontiguousarray(edges[:, 1])))
    if extra:
        e = np.asarray(extra, dtype=np.uint32)
        eng.components_union(lab, eng.to_device(np.ascontiguousarray(e[:, 0])), eng.to_device(np.ascontiguousarray(e[:, 1])))
    eng.components_finish(lab)
    return as_u32(lab).reshape(n)


def remove_set_of(labels: np.ndarray, cand_ids):
    """EN:145: survivor of a component = smallest doc_id (Python ordering), whatever its position in the input"""
    n = len(cand_ids)
    members = np.flatnonzero(labels != np.arange(n, dtype=np.uint32))
    comp_min = {}
    for k in set(labels[members].tolist()) | set(members.tolist()):
        l = int(labels[k])
        d = cand_ids[k]
        if l not in comp_min or d < comp_min[l]:
            comp_min[l] = d
    return {cand_ids[k] for k in comp_min_members(labels, members) if cand_ids[k] != comp_min[int(labels[k])]}


def removed_doc_ids(eng, sig, cand_ids, bands: int, rows: int):
    """EN:133-147 on signatures in HBM: (set of doc_ids to remove, component label per candidate row)."""
    labels = labels_of(eng, edges_of(eng, sig, bands, rows), cand_ids)
    return remove_set_of(labels, cand_ids), labels


def minhashLsh(args) -> dict:
    from .. import checkpoint as ckpt
    from .. import hostio
    files = hostio.list_files(args.input_dir)
    params = ckpt.params_of(args)
    resume = getattr(args, "resume_from", None)
    t0 = time.time()
    ctx = mp.get_context("fork")                 # forked BEFORE the GPU is touched in this process
    pool = ctx.Pool(max(1, args.workers))
    tpool = None
    try:
        import torch

        from ..minhash import Engine, as_u32, to_padded_device
        sig_h = edges = labels = None
        if resume:
            state = ckpt.load(args.inter_dir, resume, params, args.input_dir)
            cand_ids, n_rows = state["doc_ids"], state["manifest"]["extra"].get("n_rows")
            sig_h, edges, labels = state.get("signatures"), state.get("edges"), state.get("labels")
            eng = Engine(args.device)
            print(f"resuming from {resume} of {args.inter_dir} ({len(cand_ids)} candidate documents)", file=sys.stderr)
        else:
            # pass 1 on threads when the native reader is there: it inflates, scans and normalises without the GIL and
            # its buffers reach this process without pickling; the forked pool above serves the Python-heavy pass 2
            from multiprocessing.pool import ThreadPool
            tpool = ThreadPool(max(1, args.workers)) if (args.reader == "threads" or (args.reader == "auto" and hostio._io() is not None)) else None
            host_norm = args.normalize_on == "host" or (args.normalize_on == "auto" and hostio._ftfy is not None)
            parsed = (tpool or pool).imap(_parse_file, [(p, args.content_field_name, host_norm) for p in files])
            eng = Engine(args.device)
            sigs, cand_ids, n_rows, n_host_docs = [], [], 0, 0
            for path, ids, cand, text, offs in parsed:
                n_rows += len(ids)
                if len(cand):
                    buf = to_padded_device(eng, text)
                    if host_norm:
                        sigs.append(eng.signatures(buf, eng.to_device(offs), args.num_perm, args.width,
                                                   args.ngram_unit, args.seed, text_bytes=int(text.size)))
                    else:
                        s, nh = eng.signatures_en(buf, eng.to_device(offs), args.num_perm, args.width,
                                                  args.ngram_unit, args.seed)
                        sigs.append(s)
                        n_host_docs += nh
                    cand_ids.extend(ids[i] for i in cand.tolist())
            if not host_norm:
                print(f"normalised on the device; {n_host_docs} documents took the Python steps", file=sys.stderr)
            sig = torch.cat(sigs) if sigs else torch.empty((0, args.num_perm), dtype=torch.int32, device=eng.device)
            sig_h = as_u32(sig).reshape(len(cand_ids), args.num_perm)
            # the signatures are on disk before anything else happens to them (EN:128-129), so that a failure in a
            # later step costs no re-hashing: --resume_from signatures
            ckpt.save(args.inter_dir, params, cand_ids, signatures=sig_h, extra={"n_rows": n_rows, "input_dir": os.path.abspath(args.input_dir)})
        t1 = time.time()
        n = len(cand_ids)
        print("finish generate hash value")                                   # EN:132
        if edges is None and labels is None:
            sig = eng.to_device(sig_h.reshape(-1)).view(n, args.num_perm) if resume else sig
            edges = edges_of(eng, sig, args.bands, args.rows)
        if labels is None:
            labels = labels_of(eng, edges, cand_ids)
        remove_ids = remove_set_of(labels, cand_ids)
        t2 = time.time()
        if resume != "labels":
            # complete the manifest (artifacts this run did not hold, e.g. the signatures after --resume_from edges, stay listed)
            rewrite_sig = resume == "signatures" and state["manifest"]["params"] != params      # e.g. another (bands, rows)
            ckpt.save(args.inter_dir, params, cand_ids, signatures=sig_h if rewrite_sig else None, edges=edges, labels=labels,
                      extra={"n_rows": n_rows, "input_dir": os.path.abspath(args.input_dir)})
        print("the number of doc need to be remove is ", len(remove_ids))     # EN:148 (distinct doc_ids)
        os.makedirs(args.output_dir, exist_ok=True)
        tasks = [(p, os.path.join(args.output_dir, "part-%05d.json.gz" % i), remove_ids) for i, p in enumerate(files)]
        kept = sum(pool.imap_unordered(_write_file, tasks))
        t3 = time.time()
    finally:
        pool.close()
        pool.join()
        if tpool is not None:
            tpool.close()
    stats = {"n_rows": n_rows, "n_candidates": n, "n_remove": len(remove_ids), "n_kept_rows": kept,
             "n_components": int((labels == np.arange(n)).sum()) if n else 0, "resumed_from": resume,
             "t_parse_normalise_signatures_s": t1 - t0, "t_cluster_s": t2 - t1, "t_write_s": t3 - t2}
    print(json.dumps(stats), file=sys.stderr)
    return stats


def comp_min_members(labels: np.ndarray, members: np.ndarray):
    """indices of every document that belongs to a component with more than one member"""
    roots = set(labels[members].tolist())
    return sorted(set(members.tolist()) | roots)


def main(argv=None):
    args = build_parser().parse_args(argv)
    minhashLsh(args)


if __name__ == "__main__":
    main()
