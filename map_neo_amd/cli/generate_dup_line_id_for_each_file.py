#!/usr/bin/env python3
"""Drop-in for .../minhash_lsh/generate_dup_line_id_for_each_file.py (ZH-4): every member
of a component except the first is a duplicate; its "file@idx" id is split into a per-file
set of line indices, pickled to <output_dir>/<file>.pkl.  Host bookkeeping only."""
from __future__ import annotations

import argparse
import os
from collections import defaultdict

from . import zh_common as Z


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--input_file")
    p.add_argument("--output_dir")
    return p


def generate_duplicates(args):
    print("Processing duplicates!!!")
    components, _n_components, reversed_mapper = Z.read_own_pickle(args.input_file)
    per_file = defaultdict(set)
    n_dup = 0
    for comp in components:
        for node in comp[1:]:
            name, _, idx = reversed_mapper[node].rpartition("@")
            per_file[name].add(int(idx))
            n_dup += 1
    print("number of duplicate documents that will be removed:", n_dup)
    os.makedirs(args.output_dir, exist_ok=True)
    for name, idxs in per_file.items():
        Z.write_pickle(os.path.join(args.output_dir, name + ".pkl"), idxs)
    return n_dup


def main(argv=None):
    generate_duplicates(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
