"""Golden fixture for the exact-match stages, generated from the CPU oracle
oracle/ref_exact.py on a small paragraph-structured corpus.  RESTATEMENT-DERIVED like the MinHash fixtures
(the reference ships none and Spark is absent): pins the oracle against itself over time and the HIP path
against the oracle.

Run from the repo root:  python tests/golden/make_golden_exact.py   ->  tests/golden/exact_expected.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_exact as RX  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def corpus():
    rng = np.random.Generator(np.random.PCG64(19))
    words = ("data cleaning removes repeated paragraphs and boilerplate from crawled pages so that language models "
             "do not memorise navigation bars cookie banners licence footers 数据 清洗 去重 段落 模板 页脚 😀").split()
    paras = [" ".join(rng.choice(words, int(rng.integers(4, 30)))) for _ in range(50)] + ["", "short"]
    docs = ["\n".join(rng.choice(paras, int(rng.integers(1, 9)))) for _ in range(120)]
    docs += [docs[int(i)] for i in rng.integers(0, 120, 20)]
    docs += ["", "\n", "x" * 50, "y" * 51]
    ids = [f"doc-{int(i):04d}" for i in rng.permutation(len(docs))]
    return ids, docs


def main():
    ids, docs = corpus()
    out = {
        "ids": ids, "docs": docs,
        "exact_document_keep": RX.exact_document_dedup(docs),
        "paragraph_min_len_20": RX.paragraph_dedup(list(zip(ids, docs)), 20),
        "paragraph_min_len_0": RX.paragraph_dedup(list(zip(ids, docs)), 0),
        "substr_50_20": RX.exact_substr_dedup(list(zip(docs, ids)), 50, 20),
        "substr_13_0": RX.exact_substr_dedup(list(zip(docs, ids)), 13, 0),
    }
    with open(os.path.join(HERE, "exact_expected.json"), "w", encoding="utf8") as f:
        json.dump(out, f, ensure_ascii=False)
    print({k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
