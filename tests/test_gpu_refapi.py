"""GPU suite: the reference's function names (map_neo_amd.refapi) against the oracle's per-document restatement --
written the way the reference would be exercised: one MinHash object per document, _H of hashvalues slices,
generate_edges, the literal large-star / small-star loop."""
import numpy as np
import pytest

from map_neo_amd import refapi
from oracle import ref_py as R

pytestmark = pytest.mark.gpu


def test_generate_minhash_value_and_bands(mini_docs):
    pairs = [(t, f"doc{i}") for i, t in enumerate(mini_docs[:60])]
    got = refapi.generate_minhash_value(pairs)
    assert len(got) == len(pairs)
    for (hv, doc_id), (text, exp_id) in zip(got, pairs):
        m = R.MinHash(num_perm=128)                       # EN:62-63
        for s in R.shingles_char(text, 13):
            m.update(s)
        assert doc_id == exp_id and hv.dtype == np.uint64 and (hv == m.hashvalues).all()
    bands = refapi.generate_bands(got)
    assert len(bands) == 9 * len(pairs)
    for k, (idx, key, doc_id) in enumerate(bands):
        d, j = divmod(k, 9)
        assert idx == j and doc_id == pairs[d][1] and len(key) == 104
        assert key == R._H(got[d][0][j * 13:(j + 1) * 13])


def test_generate_hash_values_zh(mini_docs):
    texts = [t for t in mini_docs if any(ord(c) > 0x2E80 for c in t)][:20] + ["too short", ""]
    got = refapi.generate_hash_values(texts)
    for text, keys in zip(texts, got):
        assert keys == R.band_keys(R.signature(text).astype(np.uint64))          # ZH-1:23-33
    assert got[-1] == got[-2] and got[-1][0] == b"\x00\x00\x00\x00\xff\xff\xff\xff" * 13     # no shingles: all hashvalues 2^32-1


def test_normalize_edges_and_remove_list():
    assert refapi.normalize_text([("  Hello,   World!\n", 7)]) == [("hello world", 7)]
    ids = ["d", "a", "c", "c", "b"]
    assert refapi.generate_edges(ids) == R.generate_edges(ids) == [("d", "a"), ("c", "a"), ("c", "c"), ("c", "b")]
    rng = np.random.Generator(np.random.PCG64(3))
    edges = [tuple(sorted((f"id{int(a):03d}", f"id{int(b):03d}"), reverse=True)) for a, b in rng.integers(0, 300, (250, 2))]
    edges = [e for e in edges if e[0] != e[1]]
    star = R.large_star_small_star(edges)                                        # the literal loop of EN:138-144
    assert refapi.remove_list(edges) == {u for u, _ in star}
    assert refapi.remove_list([]) == set()


def test_lone_surrogate_raises_like_encode():
    with pytest.raises(UnicodeEncodeError):
        refapi.generate_minhash_value([("bad \ud800 text that is long enough", 1)])
