#!/usr/bin/env python3
"""Drop-in for Matrix/english_data_clean_pipeline/dedup/Minhash_Lsh_Deduplication/minhashLsh.py
(EN in SURVEY.md): same flags, same outputs, no Spark.

    python -m map_neo_amd.cli.minhashLsh --input_dir I --inter_dir M --output_dir O \
        [--content_field_name raw_content]

Input : directory of JSON-lines files (optionally .gz) with `doc_id` and the content field
        (EN:120-123).
Inter : signatures keyed by doc_id (EN:128-129 writes a Spark pickle SequenceFile of
        (uint64[128], doc_id); here: signatures.npy (uint32 [n, K]) + doc_ids.json).
Output: gzip JSON lines of the surviving rows with all original columns (EN:149-150) and
        the line "the number of doc need to be remove is  N" (EN:148).
Rows whose content is null are never hashed and therefore always kept (EN:124).  Within a
connected component the row(s) carrying the smallest doc_id (Python ordering) survive
(EN:145: remove_list = first element of every final (v, min) edge).
Extension flags (defaults = the reference's constants EN:17-20): --width --ngram_unit
--num_perm --bands --rows --seed --device.
"""
from __future__ import annotations

import argparse
import gzip
import json
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
    return p


def minhashLsh(args) -> dict:
    import torch

    from .. import hostio
    from ..dedup import dedup_local
    from ..minhash import Engine, as_u32, pack_texts

    t0 = time.time()
    rows, srcs = [], []
    for path in hostio.list_files(args.input_dir):
        for line in hostio.iter_lines(path):
            if line.strip():
                rows.append(json.loads(line))
    field = args.content_field_name
    # candidates = rows with non-null content (EN:124); sorted by doc_id so that
    # "smallest dense index of a component" == "smallest doc_id of a component" (EN:145)
    cand = [i for i, r in enumerate(rows) if r.get(field) is not None and hostio.encodable(r[field])]
    cand.sort(key=lambda i: rows[i]["doc_id"])
    texts = [hostio.normalize_text(rows[i][field]) for i in cand]
    ids = [rows[i]["doc_id"] for i in cand]
    t1 = time.time()

    eng = Engine(args.device)
    buf, offs = pack_texts(texts)
    out = dedup_local(eng, eng.to_device(buf), eng.to_device(offs), args.num_perm, args.width, args.ngram_unit,
                      args.bands, args.rows, args.seed, text_bytes=int(offs[-1]))
    labels = as_u32(out["labels"])
    # rows sharing one doc_id are one node of the reference's graph: they are adjacent after
    # the sort; link them, then re-label through the union-find on the host (tiny)
    dup_id_edges = [(k + 1, k) for k in range(len(ids) - 1) if ids[k] == ids[k + 1]]
    if dup_id_edges:
        lab = eng.components_init(len(ids))
        eng.components_union(lab, out["edge_u"], out["edge_v"])
        eu = eng.to_device(np.asarray([e[0] for e in dup_id_edges], np.uint32))
        ev = eng.to_device(np.asarray([e[1] for e in dup_id_edges], np.uint32))
        eng.components_union(lab, eu, ev)
        eng.components_finish(lab)
        labels = as_u32(lab)
    sig = as_u32(out["sig"])
    t2 = time.time()

    os.makedirs(args.inter_dir, exist_ok=True)
    np.save(os.path.join(args.inter_dir, "signatures.npy"), sig)
    with open(os.path.join(args.inter_dir, "doc_ids.json"), "w", encoding="utf8") as f:
        json.dump(ids, f, ensure_ascii=False)
    print("finish generate hash value")                                   # EN:132

    remove_ids = {ids[k] for k in range(len(ids)) if ids[int(labels[k])] != ids[k]}
    n_remove = len(remove_ids)
    print("the number of doc need to be remove is ", n_remove)            # EN:148
    os.makedirs(args.output_dir, exist_ok=True)
    kept = 0
    with gzip.open(os.path.join(args.output_dir, "part-00000.json.gz"), "wt", encoding="utf8") as f:
        for r in rows:                                                     # left_anti join on doc_id (EN:149)
            if r.get("doc_id") not in remove_ids:
                f.write(json.dumps(r, ensure_ascii=False) + "\n")
                kept += 1
    t3 = time.time()
    stats = {"n_rows": len(rows), "n_candidates": len(ids), "n_remove": n_remove, "n_kept_rows": kept,
             "n_components": int((labels == np.arange(len(ids))).sum()),
             "t_read_normalise_s": t1 - t0, "t_gpu_s": t2 - t1, "t_write_s": t3 - t2}
    print(json.dumps(stats), file=sys.stderr)
    return stats


def main(argv=None):
    args = build_parser().parse_args(argv)
    minhashLsh(args)


if __name__ == "__main__":
    main()
