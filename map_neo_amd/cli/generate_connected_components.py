#!/usr/bin/env python3
"""Drop-in for .../minhash_lsh/generate_connected_components.py (ZH-3): reads every
"a :: b" pair file of --input_dir, drops self pairs (ZH-3:41), computes connected components
on the GPU (union-find instead of networkit) and pickles
(components, n_components, {node: doc_id}) to --output_file (ZH-3:55-59).
Node ids are assigned in (file name, line index) order and each component lists its nodes in
ascending order, so component[0] -- the document the next stage keeps (ZH-4:18) -- is the
smallest (file, line): a legal instance of the reference's hash-seed dependent choice."""
from __future__ import annotations

import argparse
import time
from glob import glob

import numpy as np

from . import zh_common as Z


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--input_dir")
    p.add_argument("--output_file")
    p.add_argument("--num_workers", default=90, type=int)
    p.add_argument("--device", type=int, default=0)
    return p


def generate_connected_components_mp(args):
    from ..minhash import Engine, as_u32
    start = time.time()
    print("Started graph building")
    pairs = set()
    for fp in sorted(glob(f"{args.input_dir}/*")):
        with open(fp, "r", encoding="utf8") as f:
            for line in f:
                pair = tuple(line.strip().split(" :: "))
                if len(pair) == 2 and pair[0] != pair[1]:
                    pairs.add(pair)
    print("length of the set of duplicates:", len(pairs), time.time() - start)
    names = sorted({d for p in pairs for d in p}, key=Z.natural_key)
    node = {d: i for i, d in enumerate(names)}
    n = len(names)
    eng = Engine(args.device)
    labels = eng.components_init(n)
    if pairs:
        e = np.asarray([(node[a], node[b]) for a, b in pairs], dtype=np.uint32)
        eng.components_union(labels, eng.to_device(np.ascontiguousarray(e[:, 0])),
                             eng.to_device(np.ascontiguousarray(e[:, 1])))
    eng.components_finish(labels)
    lab = as_u32(labels)
    order = np.argsort(lab, kind="stable")
    bounds = np.flatnonzero(np.diff(lab[order])) + 1
    components = [c.tolist() for c in np.split(order, bounds)] if n else []
    n_components = len(components)
    print("number of connected components:", n_components, time.time() - start)
    reversed_mapper = {i: d for d, i in node.items()}
    Z.write_pickle(args.output_file, (components, n_components, reversed_mapper))
    print("Graph generated duplicates list!!!", time.time() - start)


def main(argv=None):
    generate_connected_components_mp(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
