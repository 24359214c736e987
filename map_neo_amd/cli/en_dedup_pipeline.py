#!/usr/bin/env python3
"""The four dedup steps of the EN pipeline in ONE process, with the corpus staying in HBM between them
(an extension: the reference runs four Spark jobs with gzip JSON in between,
Matrix/english_data_clean_pipeline/README.md "Deduplication"):

    exact document dedup -> MinHash-LSH near dedup -> paragraph dedup -> exact substring dedup

    python -m map_neo_amd.cli.en_dedup_pipeline --input_dir I --output_dir O

Every step is the function its own drop-in command uses, applied in the order and with the document
order those commands would see when chained through files, so the output equals running
`exact_document_deduplication`, `minhashLsh`, `paragraph`, `exact_substr_dedup` one after the other
(tests/test_gpu_cli.py checks that).  Output: gzip JSON lines {doc_id, raw_content} -- the columns that
survive the paragraph step (PAR:17,26-27).  Rows with null content never reach the output (PAR:17)."""
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
    p.add_argument("--device", type=int, default=0)
    p.add_argument("--workers", type=int, default=min(16, os.cpu_count() or 1))
    return p


def _normalised_signatures(eng, text, offs, num_perm=128, width=13, seed=1):
    """EN:31-40 + EN:56-64 for a packed corpus in HBM (Engine.signatures_en: device normalisation, Python steps
    for the documents the device flags -- all of them when ftfy is installed)."""
    return eng.signatures_en(text, offs, num_perm, width, "char", seed)[0]


def run(args) -> dict:
    import torch

    from .. import exact
    from .minhashLsh import removed_doc_ids
    t = [time.time()]
    c = load_corpus(args.input_dir, args.workers, args.device)
    eng, ids = c["eng"], c["doc_ids"]
    text, offs = c["text"], c["offs"]
    counts = {"rows_with_content": len(ids)}
    t.append(time.time())
    # 1. exact document dedup (EXD:17): first row of every content
    keep = exact.exact_document_dedup_packed(eng, text, offs)
    sel = torch.nonzero(keep).flatten()
    text, offs = reorder_packed(eng, text, offs, sel)
    ids = [ids[k] for k in sel.cpu().tolist()]
    counts["after_exact_document"] = len(ids)
    t.append(time.time())
    # 2. MinHash-LSH (EN:120-149): rows whose doc_id is removed disappear, smallest doc_id of a component stays
    sig = _normalised_signatures(eng, text, offs)
    remove_ids, _ = removed_doc_ids(eng, sig, ids, 9, 13)
    del sig
    sel = eng.to_device(np.asarray([k for k, d in enumerate(ids) if d not in remove_ids], dtype=np.int64))
    text, offs = reorder_packed(eng, text, offs, sel)
    ids = [d for d in ids if d not in remove_ids]
    counts["after_minhash"] = len(ids)
    t.append(time.time())
    # 3. paragraph dedup (PAR:15-30): documents in doc_id order
    from ..hostio import id_order_key
    order = sorted(range(len(ids)), key=lambda k: id_order_key(ids[k]))
    text, offs = reorder_packed(eng, text, offs, eng.to_device(np.asarray(order, dtype=np.int64)))
    text, offs, docs = exact.paragraph_dedup_packed(eng, text, offs, 20)
    ids = [ids[order[d]] for d in docs.cpu().tolist()]
    counts["after_paragraph"] = len(ids)
    t.append(time.time())
    # 4. exact substring dedup (SUB:34-95): documents in the order the paragraph step wrote them
    text, offs, docs = exact.exact_substr_dedup_packed(eng, text, offs, 50, 20)
    ids = [ids[d] for d in docs.cpu().tolist()]
    counts["after_exact_substring"] = len(ids)
    t.append(time.time())
    write_docs(args.output_dir, ids, text, offs, args.workers)
    t.append(time.time())
    stats = dict(counts, bytes_out=int(offs[-1].item()),
                 **{k: t[i + 1] - t[i] for i, k in enumerate(["t_read_s", "t_exact_document_s", "t_minhash_s", "t_paragraph_s",
                                                               "t_exact_substring_s", "t_write_s"])})
    print(json.dumps(stats), file=sys.stderr)
    return stats


def main(argv=None):
    run(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
