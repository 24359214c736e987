#!/usr/bin/env python3
"""Drop-in for .../deduplication/minhash_lsh/generate_minhash.py (ZH-1): signatures and
band keys of every jsonl file of --input_dir, written per band under --output_dir.
Flags as the reference (ZH-1:82-85); --workers is accepted and ignored (the GPU does the
work the pool did).  Extension flags: --width --ngram_unit --num_perm --bands --rows --seed
--device."""
from __future__ import annotations

import argparse
import os

import numpy as np

from . import zh_common as Z


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--workers", default=42, type=int)
    p.add_argument("--input_dir", default="/root/data/dedup_dataset_after")
    p.add_argument("--output_dir", default="/root/data/minhash_value")
    p.add_argument("--content_field_name", default="text", type=str, help="field name of content in json")
    p.add_argument("--width", type=int, default=Z.WIDTH)
    p.add_argument("--ngram_unit", choices=["char", "word"], default="char")
    p.add_argument("--num_perm", type=int, default=Z.NUM_PERM)
    p.add_argument("--bands", type=int, default=Z.BANDS)
    p.add_argument("--rows", type=int, default=Z.ROWS)
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--device", type=int, default=0)
    return p


def process_dir(args):
    from .. import hostio
    from ..minhash import Engine, as_u32, padded_size
    eng = Engine(args.device)
    os.makedirs(args.output_dir, exist_ok=True)
    for band in range(args.bands):
        os.makedirs(os.path.join(args.output_dir, str(band)), exist_ok=True)
    files = hostio.list_files(args.input_dir)
    # shards are parsed on threads ahead of the device; --workers (the reference's pool size, default 42) is capped
    for fi, line_no, text, offs in Z.iter_shards(files, args.content_field_name, min(16, args.workers)):   # no normalisation in the ZH path (ZH-1:46)
        name = os.path.basename(files[fi])
        print(f"process file {name}")
        buf = np.zeros(padded_size(text.size), dtype=np.uint8)
        buf[: text.size] = text
        sig = as_u32(eng.signatures(eng.to_device(buf), eng.to_device(offs), args.num_perm, args.width,
                                    args.ngram_unit, args.seed, text_bytes=int(text.size)))
        sig = sig.reshape(len(line_no), args.num_perm)
        ids = [f"{name}@{idx}" for idx in line_no.tolist()]
        for band in range(args.bands):
            keys = Z.band_key_bytes(sig, band, args.rows)
            recs = [{"doc_id": d, "hash": k} for d, k in zip(ids, keys)]
            Z.write_pickle(os.path.join(args.output_dir, str(band), name + ".gz"), recs)
        print(f"finish process file {name}")


def main(argv=None):
    process_dir(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
