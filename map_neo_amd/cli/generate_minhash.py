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
    p.add_argument("--ingest", choices=["device", "host"], default="device",
                   help="who extracts the content field (ZH-1:40-46): the GPU from the shard's raw bytes (default; unsure lines, "
                        "gzip shards go through the host reader) or host threads")
    return p


def process_dir(args):
    from .. import hostio
    from ..minhash import Engine, as_u32, to_padded_device
    eng = Engine(args.device)
    os.makedirs(args.output_dir, exist_ok=True)
    for band in range(args.bands):
        os.makedirs(os.path.join(args.output_dir, str(band)), exist_ok=True)
    files = hostio.list_files(args.input_dir)
    # shards are parsed on threads ahead of the device; --workers (the reference's pool size, default 42) is capped
    def per_file():
        """(file index, line numbers, signatures u32 [n, K]) for every file with documents"""
        if args.ingest == "device":
            for fidx, line_no, text, offs, nbytes in Z.iter_shards_device(eng, files, args.content_field_name, min(16, args.workers)):
                sig = as_u32(eng.signatures(text, offs, args.num_perm, args.width, args.ngram_unit, args.seed, text_bytes=nbytes))
                sig = sig.reshape(len(line_no), args.num_perm)
                for fi in np.unique(fidx).tolist():           # a batch holds several files: split its rows
                    m = fidx == fi
                    yield fi, line_no[m], sig[m]
        else:
            for fi, line_no, text, offs in Z.iter_shards(files, args.content_field_name, min(16, args.workers)):
                sig = as_u32(eng.signatures(to_padded_device(eng, text), eng.to_device(offs), args.num_perm, args.width, args.ngram_unit,
                                            args.seed, text_bytes=int(text.size)))
                yield fi, line_no, sig.reshape(len(line_no), args.num_perm)

    seen = set()

    def with_empty_files():      # ZH-1:58-61 writes its (empty) band lists for a file without documents too
        for item in per_file():
            seen.add(item[0])
            yield item
        for fi in range(len(files)):
            if fi not in seen:
                yield fi, np.zeros(0, np.uint64), np.zeros((0, args.num_perm), np.uint32)

    for fi, line_no, sig in with_empty_files():   # no normalisation in the ZH path (ZH-1:46)
        name = os.path.basename(files[fi])
        print(f"process file {name}")
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
