#!/usr/bin/env python3
"""Drop-in for Matrix/english_data_clean_pipeline/dedup/Exact_Substring_Deduplication/
exact_substr_dedup.py: same flags, same outputs, no Spark.

    python -m map_neo_amd.cli.exact_substr_dedup --input_dir I --output_dir O

Output: gzip JSON lines {`doc_id`, `raw_content`} (SUB:92-94) of the documents that keep at
least min_length = 20 characters after every repeated 50-character window has been deleted
(SUB:10-11,34-47,55-70,90), plus the two count lines (SUB:79,95).  Among the occurrences of a
window the one at the smallest (document in input order, offset) stays; Spark's groupByKey
leaves that choice to its shuffle order.  Rows with null content (the reference fails on them,
SUB:38) are skipped."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

from .en_common import load_corpus, write_docs


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--input_dir", required=True)
    p.add_argument("--output_dir", required=True)
    p.add_argument("--duplicate_len", type=int, default=50)
    p.add_argument("--min_length", type=int, default=20)
    p.add_argument("--device", type=int, default=0)
    p.add_argument("--workers", type=int, default=min(16, os.cpu_count() or 1))
    return p


def run(args) -> dict:
    from .. import exact
    t0 = time.time()
    c = load_corpus(args.input_dir, args.workers, args.device)
    ids = c["doc_ids"]
    print("the number of docs is ", len(ids))                              # SUB:79
    t1 = time.time()
    out_text, out_offs, docs = exact.exact_substr_dedup_packed(c["eng"], c["text"], c["offs"], args.duplicate_len, args.min_length)
    kept_ids = [ids[d] for d in docs.cpu().tolist()]
    t2 = time.time()
    write_docs(args.output_dir, kept_ids, out_text, out_offs, args.workers)
    print("the number of deduped docs is ", len(kept_ids))                 # SUB:95
    stats = {"n_docs_in": len(ids), "n_docs_out": len(kept_ids), "bytes_in": int(c["offs"][-1].item()) if len(ids) else 0,
             "bytes_out": int(out_offs[-1].item()), "t_read_s": t1 - t0, "t_dedup_s": t2 - t1, "t_write_s": time.time() - t2}
    print(json.dumps(stats), file=sys.stderr)
    return stats


def main(argv=None):
    run(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
