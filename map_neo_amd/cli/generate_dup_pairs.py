#!/usr/bin/env python3
"""Drop-in for .../minhash_lsh/generate_dup_pairs.py (ZH-2): per band, documents whose
band key equals an earlier document's are written as "doc :: first" lines to
<output_dir>/dup_paris_<band>.txt.  Flags as the reference (ZH-2:53-54).  Bucketing runs on
the GPU (mdd_bucket_edges on PACKED records); "first" is the first document in sorted file
order, one legal instance of the reference's os.listdir order."""
from __future__ import annotations

import argparse
import os

import numpy as np

from . import zh_common as Z


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--input_dir")
    p.add_argument("--output_dir")
    p.add_argument("--device", type=int, default=0)
    return p


def process_partition(args):
    from .. import hostio
    from ..minhash import Engine, as_u64
    eng = Engine(args.device)
    os.makedirs(args.output_dir, exist_ok=True)
    for band_name in sorted(os.listdir(args.input_dir)):
        band_dir = os.path.join(args.input_dir, band_name)
        if not os.path.isdir(band_dir):
            continue
        ids, keys = [], []
        for path in hostio.list_files(band_dir):
            print(f"begin process {os.path.basename(path)}")
            try:
                for rec in Z.read_own_pickle(path):
                    ids.append(rec["doc_id"])
                    keys.append(rec["hash"])
                print(f"finish process file {os.path.basename(path)}")
            except Exception as e:  # noqa: BLE001
                print(f"process file {os.path.basename(path)} fail, erros: {e}")
        out_path = os.path.join(args.output_dir, f"dup_paris_{band_name}.txt")
        with open(out_path, "w", encoding="utf8") as fw:
            if not ids:
                continue
            rows = len(keys[0]) // 8
            tup = Z.key_bytes_to_tuples(keys, rows)
            n = len(ids)
            meta = np.arange(n, dtype=np.uint64)                 # one band per call: band field 0
            t_d, m_d = eng.to_device(np.ascontiguousarray(tup.reshape(-1))), eng.to_device(meta)
            k_d = eng.packed_keys(t_d, m_d, rows)
            # ZH-2:30-37 writes "doc :: first doc seen with this key": the bucket's lowest record (rec_bucket)
            _, _, _, rb = eng.bucket_edges(k_d, m_d, t_d, 1, rows, 0, rows, want_buckets=True)
            first = as_u64(rb).astype(np.int64)
            for u in np.flatnonzero(first != np.arange(n)).tolist():   # in document order like the sequential scan
                fw.write(f"{ids[u]} :: {ids[int(first[u])]}\n")


def main(argv=None):
    process_partition(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
