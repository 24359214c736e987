#!/usr/bin/env python3
"""Drop-in for .../minhash_lsh/remove_dup.py (ZH-5): copies every input file to
--output_dir without the lines listed in <dup_line_id_dir>/<file>.pkl (byte-for-byte line
copy).  Deliberate deviation: a file with no .pkl (no duplicates) is copied
unchanged, whereas the reference raises inside apply_async and silently drops the file."""
from __future__ import annotations

import argparse
import os
from concurrent.futures import ThreadPoolExecutor

from . import zh_common as Z


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--input_dir")
    p.add_argument("--output_dir")
    p.add_argument("--dup_line_id_dir", type=str, help="dir where contain the dup line id file")
    p.add_argument("--workers", type=int, default=8)
    return p


def remove_dup(file_path, dup_line_id_path, output_path):
    drop = Z.read_own_pickle(dup_line_id_path) if os.path.exists(dup_line_id_path) else set()
    Z.copy_without_lines(file_path, output_path, drop)


def remove_dup_in_dir(args):
    from .. import hostio
    os.makedirs(args.output_dir, exist_ok=True)
    jobs = []
    with ThreadPoolExecutor(max_workers=max(1, args.workers or 1)) as pool:
        for path in hostio.list_files(args.input_dir):
            name = os.path.basename(path)
            jobs.append(pool.submit(remove_dup, path, os.path.join(args.dup_line_id_dir, name + ".pkl"),
                                    os.path.join(args.output_dir, name)))
        for j in jobs:
            j.result()


def main(argv=None):
    remove_dup_in_dir(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
