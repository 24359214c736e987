#!/usr/bin/env python3
"""Drop-in for Matrix/english_data_clean_pipeline/dedup/Exact_Document_Deduplication/
exact_document_deduplication.py: same flags, same outputs, no Spark.

    python -m map_neo_amd.cli.exact_document_deduplication --input_dir I --output_dir O

One row per distinct `raw_content` survives with all its columns (EXD:16-18 dropDuplicates +
gzip JSON).  Spark keeps an arbitrary row of each group; here the first in (file, line) order.
Rows whose content is null / missing / not a string form one group, as Spark's null does."""
from __future__ import annotations

import argparse
import gzip
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

from .en_common import CONTENT_FIELD_NAME, load_corpus


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--input_dir", help="input directories")
    p.add_argument("--output_dir", help="output directory")
    p.add_argument("--device", type=int, default=0)
    p.add_argument("--workers", type=int, default=min(16, os.cpu_count() or 1))
    return p


def _rewrite(task):
    path, out_path, keep_lines = task
    from .. import hostio
    kept = 0
    with gzip.open(out_path, "wt", encoding="utf8", compresslevel=6) as f:
        for k, line in enumerate(hostio.iter_lines(path)):
            if k in keep_lines:
                f.write(json.dumps(json.loads(line), ensure_ascii=False) + "\n")
                kept += 1
    return kept


def run(args) -> dict:
    from .. import exact
    t0 = time.time()
    c = load_corpus(args.input_dir, args.workers, args.device, id_field=None)
    t1 = time.time()
    keep = exact.exact_document_dedup_packed(c["eng"], c["text"], c["offs"]).cpu().numpy()
    t2 = time.time()
    keep_lines = [set() for _ in c["files"]]
    for fi, row in zip(c["doc_file"][keep].tolist(), c["doc_row"][keep].tolist()):
        keep_lines[fi].add(row)
    seen_null = False
    for fi, (ids, line_no, cand) in enumerate(c["rows_per_file"]):       # the null group: its first row survives
        if not seen_null and len(cand) < len(ids):
            is_cand = np.zeros(len(ids), dtype=bool)
            is_cand[cand] = True
            keep_lines[fi].add(int(line_no[np.flatnonzero(~is_cand)[0]]))
            seen_null = True
    os.makedirs(args.output_dir, exist_ok=True)
    tasks = [(p, os.path.join(args.output_dir, "part-%05d.json.gz" % i), keep_lines[i]) for i, p in enumerate(c["files"])]
    if args.workers <= 1 or len(tasks) <= 1:
        kept = sum(map(_rewrite, tasks))
    else:
        with mp.get_context("spawn").Pool(min(args.workers, len(tasks))) as pool:
            kept = sum(pool.map(_rewrite, tasks))
    stats = {"n_rows": sum(len(ids) for ids, _, _ in c["rows_per_file"]), "n_docs": len(c["doc_file"]), "n_kept_rows": kept,
             "t_read_s": t1 - t0, "t_group_s": t2 - t1, "t_write_s": time.time() - t2}
    print(json.dumps(stats), file=sys.stderr)
    return stats


def main(argv=None):
    run(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
