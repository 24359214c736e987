#!/usr/bin/env python3
"""Drop-in for Matrix/chinese_data_clean_pipeline/deduplication/full_text_dedup/full_text_dedup.py (FTD):
same flags, same output layout, exact instead of probabilistic.

    python -m map_neo_amd.cli.full_text_dedup --input_dir I --output_dir O [--content_filed_name text]

The reference streams every line through a ScalableBloomFilter on the content field (FTD:62-71): the first line
with a given text is written, later ones are dropped -- plus the filter's false positives (error_rate 0.001 drops
about one unique document in a thousand) and whatever order its five reader processes happen to interleave.  Here
the grouping is exact (`mdd_exact_group`, equal bytes) and "first" is (file in sorted order, line): the outcome
the reference approaches as its error rate goes to zero.  Output: the surviving lines, byte for byte, in files
part_<k>.jsonl of --batch_size lines (FTD:74-79; the reference's last, short part has no extension, FTD:86: kept).
Lines that are not JSON are skipped (FTD:36-39); lines whose content field is missing end the reference's
consumer loop with a KeyError -- here they are skipped and counted.  The Bloom-filter flags are accepted and ignored."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

from .en_common import load_corpus


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--input_dir", type=str, default="./dedup_partition/", help="input dir where contatin jsonl file")
    p.add_argument("--output_dir", type=str, default="./deduped_dataset", help="output dir where to save deduplicated jsonl file")
    p.add_argument("--queue_size", type=int, default=1000000)
    p.add_argument("--read_worker_num", type=int, default=5)
    p.add_argument("--write_worker_num", type=int, default=2)
    p.add_argument("--bf_init_capacity", type=int, default=100000)
    p.add_argument("--bf_error_rate", type=float, default=0.001)
    p.add_argument("--batch_size", type=int, default=100000, help="specify the number of lines of output jonsl files")
    p.add_argument("--content_filed_name", type=str, default="text", help="specify the field name of content in jsonl file")
    p.add_argument("--device", type=int, default=0)
    return p


def run(args) -> dict:
    from .. import exact, hostio
    t0 = time.time()
    c = load_corpus(args.input_dir, max(1, args.read_worker_num), args.device, content_field=args.content_filed_name, id_field=None)
    keep = exact.exact_document_dedup_packed(c["eng"], c["text"], c["offs"]).cpu().numpy()
    t1 = time.time()
    keep_lines = [set() for _ in c["files"]]
    for fi, row in zip(c["doc_file"][keep].tolist(), c["doc_row"][keep].tolist()):
        keep_lines[fi].add(row)
    os.makedirs(args.output_dir, exist_ok=True)
    batch, part_id, written = [], 0, 0

    def flush(last):
        nonlocal batch, part_id, written
        name = f"part_{part_id}" if last else f"part_{part_id}.jsonl"       # FTD:74 vs FTD:86
        with open(os.path.join(args.output_dir, name), "w", encoding="utf8") as f:
            f.writelines(batch)
        written += len(batch)
        part_id += 1
        batch = []

    for fi, path in enumerate(c["files"]):
        for k, line in enumerate(hostio.iter_lines(path)):
            if k in keep_lines[fi]:
                batch.append(line if line.endswith("\n") else line + "\n")
                if len(batch) >= args.batch_size:
                    flush(False)
    if batch:
        flush(True)
    n_rows = sum(len(ids) for ids, _, _ in c["rows_per_file"])
    stats = {"n_rows": n_rows, "n_docs": int(len(keep)), "n_without_content": n_rows - int(len(keep)), "n_kept": written,
             "n_parts": part_id, "t_read_group_s": t1 - t0, "t_write_s": time.time() - t1}
    print(json.dumps(stats), file=sys.stderr)
    return stats


def main(argv=None):
    run(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
