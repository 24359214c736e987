"""Generates the committed golden fixtures from the CPU oracle (oracle/ref_py.py).

The reference ships no fixtures for this path and its dependencies (datasketch, nltk) cannot
be imported here (ModuleNotFoundError), so these vectors are
RESTATEMENT-DERIVED: they pin the oracle against itself over time and pin the HIP path
against the oracle, not against a running datasketch ("parity unpinned").

Run from the repo root:  python tests/golden/make_golden.py
Outputs: tests/golden/perms_seed1_K256.npy, mini_corpus.json, mini_expected.npz
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_py as R  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def mini_corpus():
    rng = np.random.Generator(np.random.PCG64(7))
    en_words = ("the quick brown fox jumps over lazy dog data pipeline removes near duplicate documents from "
                "web crawl using minhash locality sensitive hashing with thirteen character shingles and nine "
                "bands of rows each so that similar pages collide in at least one bucket").split()
    zh_chars = "在日军的重重封锁之下行路难是此时中国广大人民最适合诠释数据清洗去重文本相似哈希分桶连通分量保留一篇文档网络爬取语料库质量过滤"
    emoji = "😀🎉🚀🌍🔥"
    docs = []

    def en(n):
        return " ".join(rng.choice(en_words, n))

    def zh(n):
        return "".join(rng.choice(list(zh_chars), n))

    # edge lengths in code points: 0, 1, 12, 13, 14 (ASCII, CJK, emoji)
    docs += ["", "a", "abcdefghijkl", "abcdefghijklm", "abcdefghijklmn", "too short"]
    docs += ["中", "中" * 12, zh_chars[:13], zh_chars[:14], emoji[:1], (emoji * 3)[:13], (emoji * 3)[:14]]
    docs += ["the quick brown fox jumps over the lazy dog",
             "在日军的重重封锁之下，行路难是此时的中国广大人民最适合的诠释。"]
    # disjoint documents
    for _ in range(40):
        docs.append(en(int(rng.integers(5, 120))))
    for _ in range(30):
        docs.append(zh(int(rng.integers(13, 200))))
    # mixed-script documents (1-, 2-, 3- and 4-byte code points interleaved)
    for _ in range(20):
        parts = []
        for _ in range(int(rng.integers(3, 12))):
            c = rng.integers(0, 4)
            parts.append(en(int(rng.integers(1, 6))) if c == 0 else zh(int(rng.integers(1, 9))) if c == 1
                         else "".join(rng.choice(list(emoji), int(rng.integers(1, 4)))) if c == 2
                         else "héllo wörld ñandú çédille")
        docs.append(" ".join(parts))
    base = len(docs)
    # exact duplicates
    for i in range(10):
        docs.append(docs[20 + i])
    # near duplicates: one token / character edited in a long document
    longs = [en(300) for _ in range(8)] + [zh(400) for _ in range(6)]
    docs += longs
    for t in longs:
        for _ in range(3):
            if " " in t:
                w = t.split(" ")
                w[int(rng.integers(0, len(w)))] = "edited"
                docs.append(" ".join(w))
            else:
                k = int(rng.integers(0, len(t)))
                docs.append(t[:k] + "改" + t[k + 1:])
    # a chain a~b~c where a and c are far apart (tests transitive components)
    t = en(400).split(" ")
    docs.append(" ".join(t))
    docs.append(" ".join(t[:200] + [w[::-1] for w in t[200:230]] + t[230:]))
    docs.append(" ".join(t[:200] + [w[::-1] for w in t[200:260]] + t[260:]))
    while len(docs) < 200:
        docs.append(en(int(rng.integers(1, 40))))
    assert len(docs) == 200, len(docs)
    return docs


def main():
    a, b = R.make_perms(1, 256)
    np.save(os.path.join(HERE, "perms_seed1_K256.npy"), np.stack([a, b]))
    docs = mini_corpus()
    with open(os.path.join(HERE, "mini_corpus.json"), "w", encoding="utf8") as f:
        json.dump(docs, f, ensure_ascii=False, indent=0)
    out = {}
    sig256 = R.signatures(docs, 256, 13, "char")
    out["sig_char13_K256"] = sig256
    out["sig_word5_K128"] = R.signatures(docs, 128, 5, "word")
    out["sig_char5_K128"] = R.signatures(docs, 128, 5, "char")
    out["sig_char20_K64"] = R.signatures(docs, 64, 20, "char")
    for (bnd, rows, K) in [(9, 13, 128), (14, 9, 128), (20, 12, 256)]:
        sig = sig256[:, :K]
        bk = R.buckets(sig, bnd, rows)
        edges = set()
        for members in bk.values():
            edges.update(R.generate_edges(members))
        labels = R.components_min_label(len(docs), edges)
        # literal EN:82-115,138-145 star loop: remove_list = first element of every final edge
        star = R.large_star_small_star(edges) if edges else set()
        assert {u for u, _ in star} == {int(v) for v in range(len(docs)) if labels[v] != v}
        out[f"labels_b{bnd}_r{rows}_K{K}"] = labels
        # bucket id per record (band-major: rec = band * n + doc): smallest doc in the bucket
        bid = np.zeros((bnd, len(docs)), np.uint32)
        for (j, _key), members in bk.items():
            for d in members:
                bid[j, d] = members[0]
        out[f"bucketmin_b{bnd}_r{rows}_K{K}"] = bid
    np.savez_compressed(os.path.join(HERE, "mini_expected.npz"), **out)
    print("docs", len(docs), "removed(9x13):", int((out["labels_b9_r13_K128"] != np.arange(200)).sum()))


if __name__ == "__main__":
    main()
