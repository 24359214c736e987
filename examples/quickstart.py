#!/usr/bin/env python3
"""Smallest useful tour of the library on one MI355X (run from the repository root after
`python -c "import __graft_entry__ as g; g.build()"`):

    python examples/quickstart.py

1. MinHash-LSH near-dedup of a handful of documents with the reference's constants (char 13-grams,
   128 permutations, 9 bands x 13 rows): signatures -> clusters -> which documents to remove;
2. the exact-match stages around it (exact document, paragraph, exact substring dedup)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from map_neo_amd import exact                                     # noqa: E402
from map_neo_amd.dedup import dedup_local                         # noqa: E402
from map_neo_amd.minhash import Engine, as_u32, pack_texts        # noqa: E402


def main():
    eng = Engine(0)
    base = ("the quick brown fox jumps over the lazy dog while the small grey cat watches from the warm window sill "
            "and nobody in the quiet house notices that the kettle has been boiling for quite some time now")
    docs = [base,
            base.replace("quick", "quiet"),                       # near duplicate of 0
            "an entirely different document about hash tables, atomics and memory transactions on a large accelerator",
            base,                                                 # exact duplicate of 0
            "在日军的重重封锁之下，行路难是此时的中国广大人民最适合的诠释。" * 3]
    text, offs = pack_texts(docs)
    out = dedup_local(eng, eng.to_device(text), eng.to_device(offs))
    labels = as_u32(out["labels"]).tolist()                       # labels[d] = smallest document index of d's cluster
    removed = out["removed"].cpu().numpy().astype(bool).tolist()
    print("signature of document 0 (first 4 of 128):", as_u32(out["sig"])[0, :4].tolist())
    print("cluster labels:", labels)
    print("removed       :", removed)
    assert labels == [0, 0, 2, 0, 4] and removed == [False, True, False, True, False]

    print("exact document dedup keeps:", exact.exact_document_dedup(eng, docs).tolist())
    rows = [(1, "intro\nshared paragraph that is long enough\nonly in one"), (2, "shared paragraph that is long enough\nonly in two")]
    print("paragraph dedup:", exact.paragraph_dedup(eng, rows, min_len=0))
    sub = [("x" * 10 + "a repeated passage of more than fifty characters, repeated verbatim!" + "y" * 30, "A"),
           ("z" * 25 + "a repeated passage of more than fifty characters, repeated verbatim!" + "w" * 30, "B")]
    for doc_id, cleaned in exact.exact_substr_dedup(eng, sub, duplicate_len=50, min_length=20):
        print("substring dedup:", doc_id, repr(cleaned))
    return 0


if __name__ == "__main__":
    sys.exit(main())
