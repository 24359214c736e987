#!/usr/bin/env python3
"""Drop-in for Matrix/english_data_clean_pipeline/dedup/Paragraph_Deduplication/paragraph.py:
same flags, same outputs, no Spark.

    python -m map_neo_amd.cli.paragraph --input_dir I --output_dir O

Output: gzip JSON lines with the two columns the reference keeps (`doc_id`, `raw_content`;
PAR:17,26-27), documents in doc_id order, and the line "after paragraph dedup: N" (PAR:35).
A paragraph survives in the row with the smallest (doc_id, pos) that holds it (PAR:21-23);
documents shorter than 20 characters afterwards are dropped (PAR:34 min_len=20).  Rows with
null content vanish (posexplode of null).  doc_ids are expected to be unique; rows repeating a
doc_id are treated as separate documents in input order (Spark would merge their paragraphs)."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

from .en_common import load_corpus, reorder_packed, write_docs


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--input_dir", required=True)
    p.add_argument("--output_dir", required=True)
    p.add_argument("--min_len", type=int, default=20)
    p.add_argument("--device", type=int, default=0)
    p.add_argument("--workers", type=int, default=min(16, os.cpu_count() or 1))
    return p


def run(args) -> dict:
    from .. import exact
    t0 = time.time()
    c = load_corpus(args.input_dir, args.workers, args.device)
    eng, ids = c["eng"], c["doc_ids"]
    t1 = time.time()
    from ..hostio import id_order_key
    order = sorted(range(len(ids)), key=lambda k: id_order_key(ids[k]))    # Window.orderBy(doc_id, pos); stable; nulls first
    perm = eng.to_device(np.asarray(order, dtype=np.int64))
    text, offs = reorder_packed(c["eng"], c["text"], c["offs"], perm)
    out_text, out_offs, docs = exact.paragraph_dedup_packed(eng, text, offs, args.min_len)
    kept_ids = [ids[order[d]] for d in docs.cpu().tolist()]
    t2 = time.time()
    print("after paragraph dedup: %d" % len(kept_ids))                     # PAR:35
    write_docs(args.output_dir, kept_ids, out_text, out_offs, args.workers)
    stats = {"n_docs_in": len(ids), "n_docs_out": len(kept_ids), "bytes_in": int(offs[-1].item()) if len(ids) else 0,
             "bytes_out": int(out_offs[-1].item()), "t_read_s": t1 - t0, "t_dedup_s": t2 - t1, "t_write_s": time.time() - t2}
    print(json.dumps(stats), file=sys.stderr)
    return stats


def main(argv=None):
    run(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
