"""CPU suite: pins the oracle (oracle/ref_py.py, oracle/mdd_ref.c).  The reference holds no
tests/fixtures for this path, so the pins are: FIPS 180 SHA-1 vectors, numpy's
RandomState stream, the KATs of SURVEY.md 8(c) and the committed golden fixtures."""
import hashlib

import numpy as np
import pytest

from oracle import ref_c as RC
from oracle import ref_py as R


def test_sha1_fips_vectors():
    # FIPS 180-4 / RFC 3174 test vectors
    vec = {
        b"abc": "a9993e364706816aba3e25717850c26c9cd0d89d",
        b"": "da39a3ee5e6b4b0d3255bfef95601890afd80709",
        b"abcdbcdecdefdefgefghfghighijhijkijkljklmklmnlmnomnopnopq": "84983e441c3bd26ebaae4aa1f95129e5e54670f1",
        b"a" * 1000000: "34aa973cd4c4daa4f61eeb2bdbad27316534016f",
    }
    for m, h in vec.items():
        assert RC.sha1(m).hex() == h
        assert hashlib.sha1(m).hexdigest() == h
    for n in (54, 55, 56, 57, 63, 64, 65, 119, 120, 128):
        m = bytes(range(256))[:n] * 1
        assert RC.sha1(m) == hashlib.sha1(m).digest()


def test_shingle_hash_kat():
    # sha1(b"the quick bro") = 7e39e288... => hv = 2296527230
    assert hashlib.sha1(b"the quick bro").hexdigest() == "7e39e2886158ecd1c1dd291ff270b0bafa27ee04"
    assert R.sha1_hash32(b"the quick bro") == 2296527230
    assert RC.lib().ref_sha1_hash32(b"the quick bro", 13) == 2296527230


def test_perm_table_pins(golden_perms):
    a, b = R.make_perms(1, 256)
    assert a[:4].tolist() == [775169054918279404, 2109959069025162, 401325382989534145, 1130051441076870728]
    assert b[:4].tolist() == [1758426461858698312, 965365488286768773, 1703346441743126657, 1762784241922636284]
    assert (int(a[127]), int(b[127])) == (1931671111240692334, 1454448473341514576)
    assert (int(a[255]), int(b[255])) == (1369100512183171272, 1825998840733297350)
    t256 = np.stack([a, b]).astype("<u8")
    assert hashlib.sha256(t256.tobytes()).hexdigest() == "127cdd98f137eb662e81e0574798d6f333e736ac009f170f61a4c9908ca3dc98"
    a1, b1 = R.make_perms(1, 128)
    assert hashlib.sha256(np.stack([a1, b1]).astype("<u8").tobytes()).hexdigest() == \
        "d4d3ade4e025058ec11116259daaf7131d09adce90cf9a0872dc9f0f9d405349"
    assert (a1 == a[:128]).all() and (b1 == b[:128]).all()          # prefix-consistent
    assert (golden_perms == np.stack([a, b])).all()
    for seed in (0, 1, 2, 99991):
        ac, bc = RC.make_perms(seed, 256)
        ap, bp = R.make_perms(seed, 256)
        assert (ac == ap).all() and (bc == bp).all()


def test_wraparound_semantics():
    # numpy uint64 wraps mod 2^64 BEFORE % p; the big-integer result differs
    a, b = R.make_perms(1, 128)
    hv = 2296527230
    with np.errstate(over="ignore"):
        got = (a * np.uint64(hv) + b) % R.MERSENNE_PRIME & R.MAX_HASH
    p = (1 << 61) - 1
    wrap = [(((int(x) * hv + int(y)) & ((1 << 64) - 1)) % p) & 0xFFFFFFFF for x, y in zip(a, b)]
    big = [((int(x) * hv + int(y)) % p) & 0xFFFFFFFF for x, y in zip(a, b)]
    assert got.tolist() == wrap
    assert wrap != big


SIG_KATS = [
    ("the quick brown fox jumps over the lazy dog", [16577316, 19445529, 41971003, 321773557],
     "c7c1b681d030a4ba291adad4f5119013", "2b3233b6e38a2f86dbe0e44818d34443"),
    ("在日军的重重封锁之下，行路难是此时的中国广大人民最适合的诠释。", [63320824, 76680309, 231010711, 1234772337],
     "d87c81eee12c7f03d2ebfe37fec8363e", None),
    ("abcdefghijklm", [2077423985, 3498858485, 3325784592, 2590120213], "728e622360ef3e759d3d0058e5a1cbdc", None),
    ("too short", [4294967295] * 4, "2d9ad807780d7e7ede906600643c4545", None),
]


@pytest.mark.parametrize("text,head,h128,h256", SIG_KATS)
def test_signature_kats(text, head, h128, h256):
    for literal in (False, True):
        sig = R.signature(text, 128, literal=literal)
        assert sig[:4].tolist() == head
        assert hashlib.sha256(sig.astype("<u8").tobytes()).hexdigest()[:32] == h128
    if h256:
        s256 = R.signature(text, 256)
        assert hashlib.sha256(s256.astype("<u8").tobytes()).hexdigest()[:32] == h256
        assert (s256[:128] == R.signature(text, 128)).all()


def _pack(docs):
    enc = [d.encode("utf8") for d in docs]
    offs = np.cumsum([0] + [len(e) for e in enc]).astype(np.uint64)
    return np.frombuffer(b"".join(enc) + b"\0", np.uint8), offs


def test_golden_signatures_both_oracles(mini_docs, mini_expected):
    text, offs = _pack(mini_docs)
    for name, K, w, unit in [("sig_char13_K256", 256, 13, "char"), ("sig_word5_K128", 128, 5, "word"),
                             ("sig_char5_K128", 128, 5, "char"), ("sig_char20_K64", 64, 20, "char")]:
        exp = mini_expected[name]
        got_c = RC.signatures(text, offs, K, w, unit)
        assert (got_c == exp).all(), name
    # python oracle on a subset (slow path), incl. the literal per-shingle update loop
    exp = mini_expected["sig_char13_K256"]
    for i in list(range(0, 20)) + [60, 100, 150]:
        assert (R.signature(mini_docs[i], 256).astype(np.uint32) == exp[i]).all()
    for i in (13, 14, 16):
        assert (R.signature(mini_docs[i], 256, literal=True).astype(np.uint32) == exp[i]).all()


def test_band_key_serialisation():
    sig = R.signature("the quick brown fox jumps over the lazy dog", 128)
    keys = R.band_keys(sig, 9, 13)
    assert len(keys) == 9 and all(len(k) == 104 for k in keys)
    assert keys[0][:8] == int(sig[0]).to_bytes(8, "big")
    allmax = R.band_keys(R.signature("too short", 128), 9, 13)
    assert len(set(allmax)) == 1


def test_golden_clusters_c_oracle(mini_docs, mini_expected):
    sig256 = mini_expected["sig_char13_K256"]
    n = len(mini_docs)
    for (b, r, K) in [(9, 13, 128), (14, 9, 128), (20, 12, 256)]:
        sig = np.ascontiguousarray(sig256[:, :K])
        labels, ne, bucket_of = RC.cluster(sig, b, r, want_buckets=True)
        assert (labels == mini_expected[f"labels_b{b}_r{r}_K{K}"]).all()
        # identical LSH buckets: smallest doc of each record's bucket
        bmin = (bucket_of % n).reshape(b, n)
        assert (bmin == mini_expected[f"bucketmin_b{b}_r{r}_K{K}"]).all()
    lab = mini_expected["labels_b9_r13_K128"]
    short = [i for i, d in enumerate(mini_docs) if len(d) < 13]
    assert len({int(lab[i]) for i in short}) == 1            # docs below 13 code points: one component


def test_structural_invariants(mini_docs):
    # order / multiplicity invariance of the shingle set, edge-shape invariance of components
    t = mini_docs[20]
    assert (R.signature(t + t) <= R.signature(t)).all()
    docs = mini_docs[:80]
    d1 = R.dedup(docs, edge_shape="chain")
    d2 = R.dedup(docs, edge_shape="star")
    assert (d1["labels"] == d2["labels"]).all()
    perm = np.random.Generator(np.random.PCG64(3)).permutation(len(docs))
    d3 = R.dedup([docs[i] for i in perm])
    # same partition under input permutation
    part1 = {frozenset(np.nonzero(d1["labels"] == l)[0].tolist()) for l in set(d1["labels"].tolist())}
    part3 = {frozenset(int(perm[j]) for j in np.nonzero(d3["labels"] == l)[0]) for l in set(d3["labels"].tolist())}
    assert part1 == part3
    assert d1["removed"].sum() == sum(len(p) - 1 for p in part1)


def test_star_loop_removal_rule():
    edges = {(1, 0), (2, 1), (3, 2), (5, 4), (9, 3), (9, 5), (12, 11)}
    st = R.large_star_small_star(edges)
    lab = R.components_min_label(13, edges)
    assert {u for u, _ in st} == {i for i in range(13) if lab[i] != i}


def test_normalize_text():
    assert R.normalize_text("  Hello,   World!\n\tFoo-bar  ") == "hello world foobar"
    assert R.normalize_text("ＡＢＣ　ｄｅｆ") == "ａｂｃ ｄｅｆ"      # NFC keeps full-width forms; \s+ matches U+3000


def test_c_oracle_multithreaded_equals_single(mini_docs):
    from map_neo_amd.minhash import pack_texts
    from oracle import ref_c as RC
    buf, offs = pack_texts(mini_docs)
    assert (RC.signatures(buf, offs, threads=4) == RC.signatures(buf, offs)).all()


def test_exact_stage_oracle_matches_golden():
    """oracle/ref_exact.py against the committed fixture (tests/golden/make_golden_exact.py)"""
    import json
    import os
    from oracle import ref_exact as RX
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "exact_expected.json"), encoding="utf8"))
    ids, docs = g["ids"], g["docs"]
    assert RX.exact_document_dedup(docs) == g["exact_document_keep"]
    assert [list(x) for x in RX.paragraph_dedup(list(zip(ids, docs)), 20)] == g["paragraph_min_len_20"]
    assert [list(x) for x in RX.paragraph_dedup(list(zip(ids, docs)), 0)] == g["paragraph_min_len_0"]
    assert [list(x) for x in RX.exact_substr_dedup(list(zip(docs, ids)), 50, 20)] == g["substr_50_20"]
    assert [list(x) for x in RX.exact_substr_dedup(list(zip(docs, ids)), 13, 0)] == g["substr_13_0"]


def test_word_unit_on_runs_of_spaces_both_oracles():
    """The word unit (an extension, BASELINE.json config 1) is defined on normalised text; on runs of spaces, leading and
    trailing spaces both restatements hash the text from the first token's start to the last token's end"""
    from map_neo_amd.minhash import pack_texts

    assert R.shingles_word("a b  c", 2) == [b"a b", b"b  c"]
    assert R.shingles_word("  a   ", 1) == [b"a"] and R.shingles_word("   ", 1) == []
    assert R.shingles_word("x y z", 2) == [b"x y", b"y z"]  # normalised text: ' '.join of the tokens
    rng = np.random.Generator(np.random.PCG64(1))
    docs = []
    for ln in (0, 1, 5, 255, 256, 257, 513):
        for p in (0.0, 0.3, 0.97):
            ch = np.where(rng.random(ln) < p, " ", np.where(rng.random(ln) < 0.1, "中", "a"))
            s = "".join(ch.tolist())
            docs += [s, " " + s + "  "]
    buf, offs = pack_texts(docs)
    for w in (1, 2, 5):
        a = RC.signatures(buf, offs, 64, w, "word")
        b = np.stack([R.signature(d, 64, w, "word") for d in docs])
        assert (a == b).all(), w


def test_word_unit_golden_vector_built_independently():
    """one vector for the word unit whose expectation does not come from either restatement -- the five word
    5-grams of the pangram typed out literally (nltk.ngrams over text.split(' '), joined with ' '), their SHA-1 digests
    as published by `sha1sum`, and datasketch's sha1_hash32 = little-endian u32 of digest[0:4].  Only valid on
    single-space-normalised text: that is where the word unit is defined (include/mdd.h, MDD_UNIT_WORD)."""
    text = "the quick brown fox jumps over the lazy dog"
    grams = [b"the quick brown fox jumps", b"quick brown fox jumps over", b"brown fox jumps over the", b"fox jumps over the lazy",
             b"jumps over the lazy dog"]
    digests = ["51fa0448c76be2aa3a49f6075538840570b2a775", "4f820428fcb18cbf72445e9b09edf5140a7a60ae",
               "28f7fce3b192225884f4f32f06ed1e0b784b4fce", "3df76e625831339639c6a60f83f4af4195b8a50e",
               "c767d6926710d7bd08432685bba6a2a675a0dfec"]
    hv = [1208285777, 671384143, 3825006376, 1651439421, 2463524807]
    assert [" ".join(text.split(" ")[i:i + 5]).encode() for i in range(5)] == grams
    assert R.shingles_word(text, 5) == grams
    import hashlib
    assert [hashlib.sha1(g).hexdigest() for g in grams] == digests
    assert [int.from_bytes(bytes.fromhex(d)[:4], "little") for d in digests] == hv
    assert [R.sha1_hash32(g) for g in grams] == hv
    a, b = R.make_perms(1, 128)
    exp = R.permute_min(np.array(hv, dtype=np.uint64), a, b)
    from map_neo_amd.minhash import pack_texts
    buf, offs = pack_texts([text])
    assert (RC.signatures(buf, offs, 128, 5, "word")[0] == exp.astype(np.uint32)).all()
    assert (R.signature(text, 128, 5, "word") == exp.astype(np.uint32)).all()
