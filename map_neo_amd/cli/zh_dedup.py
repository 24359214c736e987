#!/usr/bin/env python3
"""Fused ZH command: the five steps of the reference (ZH-1..ZH-5) in one process with every
intermediate kept in HBM -- no band pickles, no pair files.  Output = the filtered jsonl
files (same names as the inputs) plus kept/removed ids and cluster assignments as JSON.

    python -m map_neo_amd.cli.zh_dedup --input_dir I --output_dir O [--content_field_name text]
"""
from __future__ import annotations

import argparse
import json
import os

import numpy as np

from . import zh_common as Z


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--input_dir")
    p.add_argument("--output_dir")
    p.add_argument("--content_field_name", default="text")
    p.add_argument("--report", default=None, help="optional JSON file: removed ids + cluster of every clustered doc")
    p.add_argument("--width", type=int, default=Z.WIDTH)
    p.add_argument("--ngram_unit", choices=["char", "word"], default="char")
    p.add_argument("--num_perm", type=int, default=Z.NUM_PERM)
    p.add_argument("--bands", type=int, default=Z.BANDS)
    p.add_argument("--rows", type=int, default=Z.ROWS)
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--device", type=int, default=0)
    return p


def run(args):
    from .. import hostio
    from ..dedup import cluster_local
    from ..minhash import Engine, as_u32, padded_size
    files = hostio.list_files(args.input_dir)
    import torch
    eng = Engine(args.device)
    ids, sigs = [], []
    for path in files:                                   # one shard at a time: signatures stay in HBM, text does not
        name = os.path.basename(path)
        line_no, text, offs = Z.read_jsonl_texts(path, args.content_field_name)
        ids.extend((name, int(i)) for i in line_no.tolist())
        if len(line_no):
            buf = np.zeros(padded_size(text.size), dtype=np.uint8)
            buf[: text.size] = text
            sigs.append(eng.signatures(eng.to_device(buf), eng.to_device(offs), args.num_perm, args.width,
                                       args.ngram_unit, args.seed, text_bytes=int(text.size)))
    sig = torch.cat(sigs) if sigs else torch.empty((0, args.num_perm), dtype=torch.int32, device=eng.device)
    out = cluster_local(eng, sig, args.bands, args.rows)
    labels = as_u32(out["labels"])
    removed = labels != np.arange(len(ids), dtype=np.uint32)
    drop = {}
    for k in np.flatnonzero(removed).tolist():
        drop.setdefault(ids[k][0], set()).add(ids[k][1])
    os.makedirs(args.output_dir, exist_ok=True)
    for path in files:
        name = os.path.basename(path)
        d = drop.get(name, set())
        with open(path, "rb") as f, open(os.path.join(args.output_dir, name), "wb") as fw:
            for idx, line in enumerate(f):
                if idx not in d:
                    fw.write(line)
    n_comp = int((labels == np.arange(len(ids))).sum())
    print("number of connected components:", n_comp)
    print("number of duplicate documents that will be removed:", int(removed.sum()))
    if args.report:
        clustered = np.flatnonzero(removed | np.isin(np.arange(len(ids)), labels[removed]))
        rep = {"removed": [f"{ids[k][0]}@{ids[k][1]}" for k in np.flatnonzero(removed).tolist()],
               "cluster_of": {f"{ids[k][0]}@{ids[k][1]}": f"{ids[int(labels[k])][0]}@{ids[int(labels[k])][1]}"
                              for k in clustered.tolist()}}
        with open(args.report, "w", encoding="utf8") as f:
            json.dump(rep, f, ensure_ascii=False)
    return labels, removed


def main(argv=None):
    run(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
