"""GPU suite: exact-match stages against their oracle restatement."""
import numpy as np
import pytest

from map_neo_amd import exact
from map_neo_amd.minhash import as_u32, pack_texts
from oracle import ref_exact as RX

pytestmark = pytest.mark.gpu


def _corpus(rng, n):
    words = ["alpha", "beta", "gamma", "delta", "epsilon", "zeta", "eta", "theta", "数据", "清洗", "去重", "😀", "naïve"]
    paras = [" ".join(rng.choice(words, int(rng.integers(1, 25)))) for _ in range(60)] + ["", "", "short"]
    docs = []
    for _ in range(n):
        k = int(rng.integers(0, 9))
        docs.append("\n".join(rng.choice(paras, k)) if k else "")
    for _ in range(n // 5):
        docs.append(docs[int(rng.integers(0, n))])                           # exact copies
    return docs


def test_exact_group_primitive(engine):
    rng = np.random.Generator(np.random.PCG64(3))
    units = [bytes(rng.integers(97, 100, int(rng.integers(0, 40))).astype(np.uint8)) for _ in range(5000)]
    units += [b"x" * 300, b"x" * 300, b"x" * 299, b"", b""]
    buf, offs = pack_texts(units)
    first = {}
    exp = np.array([first.setdefault(u, i) for i, u in enumerate(units)], dtype=np.uint32)
    for key_bits in (64, 6):                                                  # 6 bits: massive tag collisions, settled on the bytes
        got = as_u32(engine.exact_group(engine.to_device(buf), engine.to_device(offs[:-1].copy()),
                                        engine.to_device(offs[1:].copy()), key_bits, text_host=buf))
        assert (got == exp).all(), key_bits
    assert engine.last_exact_collisions > 0
    # wave-per-unit hashing (long units) gives the same groups
    long_units = [bytes(rng.integers(97, 99, 700).astype(np.uint8)) for _ in range(50)]
    long_units += long_units[:10] + [long_units[0][:-1] + b"!"]
    buf, offs = pack_texts(long_units)
    first = {}
    exp = np.array([first.setdefault(u, i) for i, u in enumerate(long_units)], dtype=np.uint32)
    got = as_u32(engine.exact_group(engine.to_device(buf), engine.to_device(offs[:-1].copy()), engine.to_device(offs[1:].copy())))
    assert (got == exp).all()


def test_exact_document_dedup(engine):
    rng = np.random.Generator(np.random.PCG64(4))
    docs = _corpus(rng, 400)
    docs[7] = None
    docs[90] = None
    keep = exact.exact_document_dedup(engine, docs)
    assert keep.tolist() == RX.exact_document_dedup(docs)
    assert 0 < keep.sum() < len(docs)


@pytest.mark.parametrize("key_bits,fused", [(64, True), (5, True), (64, False), (5, False)])
def test_paragraph_dedup(engine, key_bits, fused):
    rng = np.random.Generator(np.random.PCG64(5))
    docs = _corpus(rng, 300)
    ids = [f"doc-{int(i):05d}" for i in rng.permutation(len(docs))]
    rows = list(zip(ids, docs))
    rows[11] = (rows[11][0], None)
    got = exact.paragraph_dedup(engine, rows, min_len=20, key_bits=key_bits, fused=fused)
    exp = RX.paragraph_dedup(rows, min_len=20)
    assert got == exp
    assert 0 < len(got) < len(rows)


@pytest.mark.parametrize("key_bits,fused", [(64, True), (64, False), (7, False)])
def test_exact_substr_dedup(engine, key_bits, fused):
    rng = np.random.Generator(np.random.PCG64(6))
    base = ["".join(chr(int(c)) for c in rng.integers(0x61, 0x7B, 400)) for _ in range(12)]
    zh = ["".join(chr(int(c)) for c in rng.integers(0x4E00, 0x4F00, 200)) for _ in range(6)]
    rows = []
    for i in range(60):
        parts = []
        for _ in range(int(rng.integers(1, 4))):
            src = base[int(rng.integers(0, 12))] if rng.random() < 0.7 else zh[int(rng.integers(0, 6))]
            a = int(rng.integers(0, len(src) - 60))
            parts.append(src[a:a + int(rng.integers(30, 150))])
        rows.append((" 😀 ".join(parts), f"id{i}"))
    rows += [("short text", "s1"), ("", "s2"), ("x" * 50, "s3"), ("y" * 51, "s4"), ("y" * 120, "s5")]
    got = exact.exact_substr_dedup(engine, rows, 50, 20, key_bits=key_bits, fused=fused)
    exp = RX.exact_substr_dedup(rows, 50, 20)
    assert got == exp
    assert sum(len(t) for _, t in got) < sum(len(t) for t, _ in rows)


def test_paragraph_and_substr_edge_cases(engine):
    cases = [
        [(1, "")], [(1, ""), (2, "")], [(1, "\n")], [(1, "\n\n\n"), (2, "\n")], [(1, "a\n"), (2, "\na")],
        [(1, "same\nsame\nsame")], [(2, "x\ny"), (1, "y\nx")], [(1, "tail\n"), (2, "tail"), (3, "\ntail\n\n")],
        [(1, "only one paragraph that is long enough to stay")], [(1, None), (2, None)], [],
        [(1, "多字节\n多字节\nüber\n"), (2, "über\n多字节\nnew")],
    ]
    for rows in cases:
        for min_len in (0, 3, 20):
            for fused in (True, False):
                assert exact.paragraph_dedup(engine, rows, min_len=min_len, fused=fused) == RX.paragraph_dedup(rows, min_len=min_len), \
                    (rows, min_len, fused)
    sub = [
        [("", "a")], [("x" * 50, "a")], [("x" * 51, "a")], [("x" * 52, "a")], [("ab" * 60, "a"), ("ab" * 60, "b")],
        [("é" * 120, "a")], [("0123456789" * 6, "a"), ("z" + "0123456789" * 6, "b"), ("0123456789" * 5 + "0", "c")], [],
    ]
    for rows in sub:
        for dl, ml in ((50, 20), (5, 0), (50, 0)):
            for fused in (True, False):
                assert exact.exact_substr_dedup(engine, rows, dl, ml, fused=fused) == RX.exact_substr_dedup(rows, dl, ml), (rows, dl, ml, fused)


def test_exact_substr_dedup_long_and_mixed_documents(engine):
    """fused kernels: documents longer than a segment (split across waves), multi-byte text, many repeats"""
    rng = np.random.Generator(np.random.PCG64(9))
    alpha = [chr(c) for c in range(0x61, 0x7B)] + [" "] * 5 + ["é", "数", "据", "😀"]
    blocks = ["".join(rng.choice(alpha, int(rng.integers(40, 400)))) for _ in range(30)]
    rows = []
    for i in range(40):
        k = int(rng.integers(1, 12)) if i % 8 else int(rng.integers(150, 260))     # every 8th document is long (> 16 KiB)
        rows.append(("".join(rng.choice(blocks, k)), i))
    try:
        engine.set_segment_bytes(1024)
        small_seg = exact.exact_substr_dedup(engine, rows, 50, 20)
    finally:
        engine.set_segment_bytes(0)
    got = exact.exact_substr_dedup(engine, rows, 50, 20)
    exp = RX.exact_substr_dedup(rows, 50, 20)
    assert got == exp and small_seg == exp
    assert exact.exact_substr_dedup(engine, rows, 50, 20, fused=False) == exp
    # window table smaller than the windows need: several passes partitioned by content key
    from map_neo_amd.minhash import pack_texts
    buf, offs = pack_texts([t for t, _ in rows])
    t, o, docs = engine.substr_dedup(engine.to_device(buf), engine.to_device(offs), 50, 20, table_bytes=1 << 20)
    assert engine.last_substr_parts > 3
    raw, oo = t.cpu().numpy().tobytes(), o.cpu().numpy()
    assert [(rows[d][1], raw[int(oo[k]):int(oo[k + 1])].decode("utf8")) for k, d in enumerate(docs.cpu().tolist())] == exp


def test_fused_paths_equal_generic_paths_under_heavy_repetition(engine):
    """many identical windows / paragraphs inserted concurrently: the one-pass marking (atomicMin names the loser)
    and the fused compaction must give exactly what the independent generic route gives"""
    import torch
    from map_neo_amd.minhash import pack_texts
    rng = np.random.Generator(np.random.PCG64(17))
    words = ["w%d" % i for i in range(300)]
    paras = [" ".join(rng.choice(words, int(rng.integers(5, 40)))) for _ in range(400)]
    base_docs = ["\n".join(rng.choice(paras, int(rng.integers(1, 10)))) for _ in range(300)]
    docs = [base_docs[int(i)] for i in rng.integers(0, 300, 20000)]          # ~66 copies of every document
    for i in range(0, 20000, 7):
        docs[i] = docs[i] + " tail%d" % i
    buf, offs = pack_texts(docs)
    t, o = engine.to_device(buf), engine.to_device(offs)
    for fn, args in ((exact.exact_substr_dedup_packed, (50, 20)), (exact.paragraph_dedup_packed, (20,))):
        a = fn(engine, t, o, *args, fused=True)
        b = fn(engine, t, o, *args, fused=False)
        assert torch.equal(a[2], b[2]) and torch.equal(a[1], b[1])
        n = int(a[1][-1].item())
        assert torch.equal(a[0][:n], b[0][:n]) and n > 0


def test_exact_group_more_than_2_pow_26_long_units(engine):
    """70 M overlapping 300-byte units: 64 lanes per unit would exceed the 2^32 work-items of one launch
    (regression: the wave-per-unit hash kernel is grid-stride)"""
    import torch
    n = 70_000_000
    rng = np.random.Generator(np.random.PCG64(23))
    buf = np.zeros(4096 + 128, dtype=np.uint8)
    buf[:4096] = rng.integers(0, 256, 4096)
    text = engine.to_device(buf)
    u = torch.arange(n, dtype=torch.int64, device=engine.device)
    ubeg = (u * 7) % 1000
    uend = ubeg + 300
    first = engine.exact_group(text, ubeg, uend)
    assert engine.last_exact_collisions == 0
    assert torch.equal(first.to(torch.int64), u % 1000)


def test_substr_partitioned_passes_equal_single_pass_at_scale(engine):
    """200 k synthetic documents (0.4 GB, 0.4 G windows): one table pass vs seven key-partitioned passes"""
    import torch
    from map_neo_amd import corpus
    spec = corpus.CorpusSpec(200_000, 2048, lang="en", seed=5)
    text, offs = corpus.generate_chunked(spec, engine.device)
    a = engine.substr_dedup(text, offs, 50, 20, method="table")
    assert engine.last_substr_parts == 1
    n_win = int(torch.clamp(offs[1:] - offs[:-1] - 50, min=0).sum().item())
    b = engine.substr_dedup(text, offs, 50, 20, table_bytes=16 * n_win // 7 + 1, method="table")
    assert engine.last_substr_parts == 7
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    n = int(a[1][-1].item())
    assert torch.equal(a[0][:n], b[0][:n]) and 0 < n < int(offs[-1].item())
    del b
    # and the independent generic route (explicit unit ranges through mdd_exact_group + torch plumbing)
    g = exact.exact_substr_dedup_packed(engine, text, offs, 50, 20, fused=False)
    assert torch.equal(a[1], g[1]) and torch.equal(a[2], g[2]) and torch.equal(a[0][:n], g[0][:n])
    del g
    # paragraphs: every 40th space becomes a newline
    par = text.clone()
    body = par[: int(offs[-1].item())]
    r = torch.rand(body.numel(), device=engine.device, generator=torch.Generator(device=engine.device).manual_seed(3))
    body.copy_(torch.where((body == 32) & (r < 0.025), torch.full_like(body, 10), body))
    pa = exact.paragraph_dedup_packed(engine, par, offs, 20, fused=True)
    pg = exact.paragraph_dedup_packed(engine, par, offs, 20, fused=False)
    m = int(pa[1][-1].item())
    assert torch.equal(pa[1], pg[1]) and torch.equal(pa[2], pg[2]) and torch.equal(pa[0][:m], pg[0][:m]) and m > 0


def test_exact_stages_match_golden_fixture(engine):
    """HIP path against the committed fixture tests/golden/exact_expected.json"""
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "exact_expected.json"), encoding="utf8"))
    ids, docs = g["ids"], g["docs"]
    assert exact.exact_document_dedup(engine, docs).tolist() == g["exact_document_keep"]
    for fused in (True, False):
        assert [list(x) for x in exact.paragraph_dedup(engine, list(zip(ids, docs)), 20, fused=fused)] == g["paragraph_min_len_20"]
        assert [list(x) for x in exact.paragraph_dedup(engine, list(zip(ids, docs)), 0, fused=fused)] == g["paragraph_min_len_0"]
        assert [list(x) for x in exact.exact_substr_dedup(engine, list(zip(docs, ids)), 50, 20, fused=fused)] == g["substr_50_20"]
        assert [list(x) for x in exact.exact_substr_dedup(engine, list(zip(docs, ids)), 13, 0, fused=fused)] == g["substr_13_0"]


def test_substr_undersized_table_is_an_error_not_a_miss(engine):
    """a window table with fewer slots than distinct windows: mdd_substr_check reports MDD_ERR_INVALID and the
    number of windows that found no slot (the kernel used to treat them as unique); a correctly sized table on
    the same input then passes the check and equals the oracle"""
    import ctypes as C
    import torch
    from map_neo_amd import lib as _lib
    from map_neo_amd.minhash import TEXT_PAD
    rng = np.random.Generator(np.random.PCG64(44))
    docs = ["".join(chr(int(c)) for c in rng.integers(97, 123, 400)) for _ in range(50)]     # ~17 500 distinct windows
    buf, offs = pack_texts(docs)
    text, o = engine.to_device(buf), engine.to_device(offs)
    tb = text.numel() - TEXT_PAD
    dup = torch.zeros(tb, dtype=torch.uint8, device=engine.device)
    small = torch.empty(1000, dtype=torch.int64, device=engine.device)
    p = lambda t: C.c_void_p(t.data_ptr())
    _lib.check(engine.L.mdd_substr_mark(engine.ctx, p(text), tb, p(o), len(docs), 50, 0, 1, p(small), small.numel(), p(dup)))
    n_over = C.c_uint64(0)
    rc = engine.L.mdd_substr_check(engine.ctx, C.byref(n_over))
    assert rc == -1 and n_over.value >= 17_500 - 1000 and b"tab_size below 2 x the windows" in engine.L.mdd_last_error()
    assert engine.L.mdd_substr_check(engine.ctx, C.byref(n_over)) == 0 and n_over.value == 0       # the count was cleared
    rows = [(d, i) for i, d in enumerate(docs + docs[:5])]
    assert exact.exact_substr_dedup(engine, rows, 50, 20) == RX.exact_substr_dedup(rows, 50, 20)


@pytest.fixture()
def sorted_method(engine):
    engine.substr_method = "sorted"
    yield engine
    engine.substr_method = "auto"


def test_substr_sorted_method_vs_oracle(sorted_method):
    """mdd_substr_mark_sorted (windows partitioned into leaves, grouped in LDS) == the oracle: random utf-8 documents built
    from shared pieces, windows 1 / 13 / 50 / 128, tiny segments, exact copies, ASCII + CJK + emoji"""
    engine = sorted_method
    rng = np.random.Generator(np.random.PCG64(77))
    for rep in range(6):
        docs = _corpus(rng, 60)
        docs += ["".join(chr(int(c)) for c in rng.integers(0x4E00, 0x4E40, 300)) for _ in range(5)]
        docs += [docs[-1], docs[-2][:120] + "尾", "😀" * 80, "😀" * 81]
        window = [50, 13, 1, 128, 50, 2][rep]
        seg = [0, 256, 0, 1024, 512, 0][rep]
        engine.set_segment_bytes(seg)
        try:
            rows = [(d, i) for i, d in enumerate(docs)]
            assert exact.exact_substr_dedup(engine, rows, window, 20) == RX.exact_substr_dedup(rows, window, 20), (window, seg)
            assert engine.last_substr_method == "sorted"
        finally:
            engine.set_segment_bytes(0)


def test_substr_sorted_method_heavy_repetition_and_parts(sorted_method):
    """one leaf with hundreds of thousands of identical windows (boilerplate), several key-partitioned passes, and the
    same survivors as the table method on 100 k synthetic documents"""
    import torch
    from map_neo_amd import corpus
    engine = sorted_method
    boiler = "this site uses cookies to improve your experience please accept our terms " * 3
    rng = np.random.Generator(np.random.PCG64(5))
    docs = [boiler + " ".join("w%d" % int(x) for x in rng.integers(0, 50, 12)) + boiler for _ in range(300)]
    rows = [(d, i) for i, d in enumerate(docs)]
    assert exact.exact_substr_dedup(engine, rows, 50, 20) == RX.exact_substr_dedup(rows, 50, 20)
    spec = corpus.CorpusSpec(100_000, 2048, lang="mixed", seed=5, boiler_frac=0.05, n_templates=20)
    text, offs = corpus.generate_chunked(spec, engine.device)
    a = engine.substr_dedup(text, offs, 50, 20, method="sorted")
    assert engine.last_substr_parts == 1 and engine.last_substr_method == "sorted"
    n_sort, n_win = engine.last_substr_candidates        # (windows left by the candidate filter, all windows)
    b = engine.substr_dedup(text, offs, 50, 20, method="table")
    n = int(a[1][-1].item())
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[0][:n], b[0][:n]) and 0 < n < int(offs[-1].item())
    assert 0 < n_sort <= n_win == int(torch.clamp(offs[1:] - offs[:-1] - 50, min=0).sum().item())
    c = engine.substr_dedup(text, offs, 50, 20, method="sorted", table_bytes=int(18 * 1.06 * n_sort / 3))
    assert engine.last_substr_parts >= 3
    engine.substr_filter = False
    try:
        c2 = engine.substr_dedup(text, offs, 50, 20, method="sorted", table_bytes=int(18 * 1.06 * n_win / 3))
    finally:
        engine.substr_filter = True
    assert engine.last_substr_parts >= 3
    assert torch.equal(a[1], c2[1]) and torch.equal(a[2], c2[2]) and torch.equal(a[0][:n], c2[0][:n])
    del c2
    assert torch.equal(a[1], c[1]) and torch.equal(a[2], c[2]) and torch.equal(a[0][:n], c[0][:n])
    # leaves above 640 windows are grouped in several rounds over shares of the tag space; the knob lowers the limits so that
    # this corpus (48 windows per leaf on average) takes 4-5 rounds per leaf
    import os
    os.environ["MDD_SUBSTR_LEAF_ROUND"] = "16,12"
    try:
        r = engine.substr_dedup(text, offs, 50, 20, method="sorted")
    finally:
        del os.environ["MDD_SUBSTR_LEAF_ROUND"]
    assert torch.equal(a[1], r[1]) and torch.equal(a[2], r[2]) and torch.equal(a[0][:n], r[0][:n])


def test_substr_sorted_method_big_leaves(sorted_method):
    """leaves above BIG_LEAF (16384 windows) take the sliced path (bigleaf_kernel: per-slice LDS tables joined through a
    global table).  Separator lines are the natural source: every window inside a run of one character is the same
    window.  (a) 60 documents with runs of '-' and '=' plus ordinary text, against the oracle; (b) 30 000 documents
    (28 M identical windows in one leaf, boilerplate in others), the same survivors as the table method."""
    import torch
    engine = sorted_method
    rng = np.random.Generator(np.random.PCG64(11))
    words = ["w%d" % i for i in range(40)]

    def doc(i, run):
        body = " ".join(words[int(x)] for x in rng.integers(0, 40, 30))
        return body + "\n" + "-" * run + "\n" + body[::-1] + "\n" + "=" * (run // 2) + ("\n%d" % i)

    docs = [doc(i, 700 + 10 * (i % 7)) for i in range(60)]            # 60 x ~670 windows of '-' x 50 = 40 k in one leaf
    rows = [(d, i) for i, d in enumerate(docs)]
    assert exact.exact_substr_dedup(engine, rows, 50, 20) == RX.exact_substr_dedup(rows, 50, 20)
    assert engine.last_substr_method == "sorted"
    docs = [doc(i, 1000) for i in range(30_000)]
    enc = [d.encode() for d in docs]
    offs_h = np.zeros(len(enc) + 1, dtype=np.int64)
    np.cumsum([len(b) for b in enc], out=offs_h[1:])
    buf = np.zeros(int(offs_h[-1]) + 3 - (int(offs_h[-1]) + 3) % 4 + 128, dtype=np.uint8)
    buf[: int(offs_h[-1])] = np.frombuffer(b"".join(enc), dtype=np.uint8)
    text = torch.from_numpy(buf).to(engine.device)
    offs = torch.from_numpy(offs_h).to(engine.device)
    a = engine.substr_dedup(text, offs, 50, 20, method="sorted")
    b = engine.substr_dedup(text, offs, 50, 20, method="table")
    n = int(a[1][-1].item())
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[0][:n], b[0][:n]) and 0 < n < int(offs[-1].item())


def test_substr_candidate_filter_adversarial_vs_oracle(sorted_method):
    """mdd_substr_candidates in front of the sorted method (window >= 50): repeats planted at every dword alignment, at the
    first / last hashable window of a document, across the 200-byte steps of the anchor kernel, across segment borders,
    inside multi-byte text, between a pure-ASCII and a non-ASCII document, runs of one character (every position an
    anchor), periodic text -- against the oracle, with the filter on and off"""
    engine = sorted_method
    rng = np.random.Generator(np.random.PCG64(2024))
    alpha = [chr(c) for c in range(0x61, 0x7B)] + [" "] * 4

    def rnd(n, extra=()):
        return "".join(rng.choice(alpha + list(extra), n))

    for rep in range(8):
        window = [50, 50, 64, 50, 128, 50, 50, 57][rep]
        seg = [0, 256, 0, 512, 0, 0, 256, 0][rep]
        passages = [rnd(int(rng.integers(window - 3, window + 40))) for _ in range(6)]
        passages += [rnd(window + 20, ("é", "数", "😀")) for _ in range(3)]
        passages += ["-" * (window + 30), "ab" * (window // 2 + 20), "xyz" * (window // 3 + 25)]
        docs = []
        for i in range(70):
            parts = []
            for _ in range(int(rng.integers(1, 5))):
                parts.append(rnd(int(rng.integers(0, 330)), ("é",) if i % 3 == 0 else ()))      # filler of every length: all alignments, all step phases
                parts.append(passages[int(rng.integers(0, len(passages)))])
            if i % 5 == 0:
                parts.append(rnd(int(rng.integers(0, 3))))
            docs.append("".join(parts))
        p0 = passages[0]
        docs += [p0, p0 + "x", "y" + p0, p0[: window] + "z", p0[: window], p0[1:] + "q", "", "short"]
        docs += [rnd(3000) + p0 + rnd(2500) + passages[1] + rnd(700)]                             # long document: many steps, segments
        rows = [(d, i) for i, d in enumerate(docs)]
        exp = RX.exact_substr_dedup(rows, window, 20)
        engine.set_segment_bytes(seg)
        try:
            for flt in (True, False):
                engine.substr_filter = flt
                got = exact.exact_substr_dedup(engine, rows, window, 20)
                assert got == exp, (rep, window, seg, flt)
                assert engine.last_substr_method == "sorted" and (engine.last_substr_candidates is not None) == flt
        finally:
            engine.set_segment_bytes(0)
            engine.substr_filter = True


def test_substr_candidate_filter_at_scale(sorted_method):
    """200 k synthetic documents (15 % near-copies): the filter keeps under half of the windows, the survivors are those of
    the unfiltered sorted method and of the table method; so are they when the anchors do not fit their array (the filter
    then flags every position)"""
    import torch
    from map_neo_amd import corpus
    engine = sorted_method
    spec = corpus.CorpusSpec(200_000, 2048, lang="en", seed=9)
    text, offs = corpus.generate_chunked(spec, engine.device)
    a = engine.substr_dedup(text, offs, 50, 20, method="sorted")
    n_sort, n_win = engine.last_substr_candidates
    assert 0 < n_sort < 0.5 * n_win
    n = int(a[1][-1].item())
    engine.substr_filter = False
    try:
        b = engine.substr_dedup(text, offs, 50, 20, method="sorted")
        assert engine.last_substr_candidates is None
    finally:
        engine.substr_filter = True
    c = engine.substr_dedup(text, offs, 50, 20, method="table")
    for r in (b, c):
        assert torch.equal(a[1], r[1]) and torch.equal(a[2], r[2]) and torch.equal(a[0][:n], r[0][:n]) and 0 < n < int(offs[-1].item())
    # mixed EN / ZH documents: windows of 50 code points are 50 .. 150 bytes long
    spec = corpus.CorpusSpec(60_000, 2048, lang="mixed", seed=10)
    text, offs = corpus.generate_chunked(spec, engine.device)
    a = engine.substr_dedup(text, offs, 50, 20, method="sorted")
    c = engine.substr_dedup(text, offs, 50, 20, method="table")
    n = int(a[1][-1].item())
    assert torch.equal(a[1], c[1]) and torch.equal(a[2], c[2]) and torch.equal(a[0][:n], c[0][:n]) and 0 < n < int(offs[-1].item())


def test_substr_resolved_anchors_clusters_with_edits_vs_oracle(sorted_method):
    """window 50: repeated anchors are settled in their leaves (anchor_resolve_kernel).  Clusters of 2 .. 45 near-copies of
    a document with point edits, deletions and insertions at random places (every distance between an edit and an anchor;
    groups above 32 members go to the window sort), copies cut so that a repeated passage ends exactly with its document
    (the window that ends with the document is never hashed), begins it, or sits in a document shorter than 68 bytes;
    with and without the resolve step, against the oracle"""
    import os
    engine = sorted_method
    rng = np.random.Generator(np.random.PCG64(31))
    alpha = [chr(c) for c in range(0x61, 0x7B)] + [" "] * 5
    for rep in range(4):
        docs = []
        for cl in range(10):
            base = "".join(rng.choice(alpha, int(rng.integers(60, 900))))
            n_copy = int([1, 2, 3, 5, 9, 17, 33, 45, 2, 4][cl])
            for c in range(n_copy):
                t = list(base)
                for _ in range(int(rng.integers(0, 6))):
                    at = int(rng.integers(0, len(t)))
                    kind = int(rng.integers(0, 4))
                    if kind == 0:
                        t[at] = "#" if rep % 2 else "é"                # an edit that is / is not ASCII
                    elif kind == 1:
                        del t[at]
                    elif kind == 2:
                        t.insert(at, "Q")
                    else:
                        t[at] = chr(int(rng.integers(0x61, 0x7B)))
                t = "".join(t)
                cut = int(rng.integers(0, 4))
                if cut == 1:
                    t = t[int(rng.integers(0, max(1, len(t) - 55))):]           # begins inside the base text
                elif cut == 2:
                    t = t[: int(rng.integers(50, len(t) + 1))]                  # ends inside it
                elif cut == 3:
                    a = int(rng.integers(0, max(1, len(t) - 60)))
                    t = t[a: a + int(rng.integers(50, 70))]                     # a document of 50 .. 69 bytes
                docs.append(t)
        order = rng.permutation(len(docs))
        rows = [(docs[int(i)], k) for k, i in enumerate(order)]
        exp = RX.exact_substr_dedup(rows, 50, 20)
        engine.set_segment_bytes([0, 256, 0, 512][rep])
        try:
            for knob in (None, "1"):
                if knob:
                    os.environ["MDD_SUBSTR_NO_RESOLVE"] = knob
                try:
                    got = exact.exact_substr_dedup(engine, rows, 50, 20)
                finally:
                    os.environ.pop("MDD_SUBSTR_NO_RESOLVE", None)
                assert got == exp, (rep, knob)
                assert engine.last_substr_candidates is not None
        finally:
            engine.set_segment_bytes(0)


def test_substr_anchor_leaves_of_every_size(sorted_method, monkeypatch):
    """the anchors' level-1 digit grows with the text (9 .. 11 bits: 2^20 .. 2^22 leaves) and leaves above 384 anchors are
    resolved in rounds over shares of the tag space (anchor_resolve_kernel<true>: a call with more than ~9 GB of text).
    MDD_SUBSTR_D1_BITS forces both on small inputs: 11 bits, and 4 bits (2^15 leaves: ~500 anchors per leaf for 160 MB of
    text, three rounds each).  Against the oracle on clusters of edited copies, against the table method at scale; and the
    rounds must RESOLVE, not hand everything to the window sort"""
    import torch
    from map_neo_amd import corpus
    engine = sorted_method
    rng = np.random.Generator(np.random.PCG64(77))
    alpha = [chr(c) for c in range(0x61, 0x7B)] + [" "] * 5
    docs = []
    for cl in range(40):
        base = "".join(rng.choice(alpha, int(rng.integers(60, 700))))
        for c in range(int(rng.integers(1, 12))):
            t = list(base)
            for _ in range(int(rng.integers(0, 4))):
                t[int(rng.integers(0, len(t)))] = "#"
            docs.append("".join(t))
    rows = [(docs[int(i)], k) for k, i in enumerate(rng.permutation(len(docs)))]
    exp = RX.exact_substr_dedup(rows, 50, 20)
    for bits in ("4", "10", "11"):
        monkeypatch.setenv("MDD_SUBSTR_D1_BITS", bits)
        assert exact.exact_substr_dedup(engine, rows, 50, 20) == exp, bits
    monkeypatch.delenv("MDD_SUBSTR_D1_BITS")
    spec = corpus.CorpusSpec(80_000, 2048, lang="en", seed=12)
    text, offs = corpus.generate_chunked(spec, engine.device)
    ref = engine.substr_dedup(text, offs, 50, 20, method="table")
    n = int(ref[1][-1].item())
    assert 0 < n < int(offs[-1].item())
    a = engine.substr_dedup(text, offs, 50, 20, method="sorted")
    n_sort_default, n_win = engine.last_substr_candidates
    assert torch.equal(a[1], ref[1]) and torch.equal(a[2], ref[2]) and torch.equal(a[0][:n], ref[0][:n])
    for bits in ("4", "11"):
        monkeypatch.setenv("MDD_SUBSTR_D1_BITS", bits)
        b = engine.substr_dedup(text, offs, 50, 20, method="sorted")
        n_sort, _ = engine.last_substr_candidates
        assert torch.equal(b[1], ref[1]) and torch.equal(b[2], ref[2]) and torch.equal(b[0][:n], ref[0][:n]), bits
        assert n_sort <= 1.5 * n_sort_default + 1000 and n_sort < 0.1 * n_win, (bits, n_sort, n_sort_default, n_win)


def test_para_doc_sums_equal_the_text_pass(engine):
    """mdd_para_doc_sums (per-document sums from the units + the code points mdd_exact_group_cps counts) ==
    mdd_para_compact_count (the same sums from a pass over the text), on multi-byte text with empty paragraphs, empty
    documents and repeated paragraphs; whole_doc marks exactly the documents whose output is the document itself"""
    import ctypes as C
    import torch
    from map_neo_amd import lib as _lib
    from map_neo_amd.minhash import TEXT_PAD
    rng = np.random.Generator(np.random.PCG64(12))
    docs = _corpus(rng, 500) + ["", "\n", "\n\n", "é\né\nü", "x" * 700 + "\n" + "数据" * 200]
    buf, offs = pack_texts(docs)
    text, o = engine.to_device(buf), engine.to_device(offs)
    n, tb = len(docs), text.numel() - TEXT_PAD
    p = lambda t: C.c_void_p(t.data_ptr()) if t.numel() else C.c_void_p(0)
    per_doc = torch.zeros(n, dtype=torch.int64, device=engine.device)
    _lib.check(engine.L.mdd_para_count(engine.ctx, p(text), tb, p(o), n, p(per_doc)))
    base = torch.zeros(n + 1, dtype=torch.int64, device=engine.device)
    torch.cumsum(per_doc, 0, out=base[1:])
    nu = int(base[-1].item())
    ubeg, uend = torch.zeros(nu, dtype=torch.int64, device=engine.device), torch.zeros(nu, dtype=torch.int64, device=engine.device)
    _lib.check(engine.L.mdd_para_units(engine.ctx, p(text), tb, p(o), n, p(base), nu, p(ubeg), p(uend)))
    for avg in (0, 1000):                                  # thread-per-unit and wave-per-unit hashing count alike
        first_of, cps = engine.exact_group(text, ubeg, uend, avg_unit_bytes=avg, want_cps=True)
        raw = buf.tobytes()
        exp_cps = [len(raw[int(b):int(e)].decode("utf8")) for b, e in zip(ubeg.cpu().tolist(), uend.cpu().tolist())]
        assert cps.cpu().tolist() == exp_cps
    res = []
    for which in ("text", "units"):
        kb, kc = torch.zeros(n, dtype=torch.int64, device=engine.device), torch.zeros(n, dtype=torch.int64, device=engine.device)
        hu, whole = torch.zeros(n, dtype=torch.uint8, device=engine.device), torch.zeros(n, dtype=torch.uint8, device=engine.device)
        if which == "text":
            _lib.check(engine.L.mdd_para_compact_count(engine.ctx, p(text), tb, p(o), n, p(base), nu, p(first_of), p(kb), p(kc), p(hu)))
        else:
            _lib.check(engine.L.mdd_para_doc_sums(engine.ctx, p(o), n, tb, p(base), nu, p(ubeg), p(uend), p(cps), p(first_of), p(kb), p(kc),
                                                  p(hu), p(whole)))
        res.append((kb.cpu(), kc.cpu(), hu.cpu()))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    fo = first_of.cpu().numpy().view(np.uint32)
    bs = base.cpu().numpy()
    exp_whole = [int(bs[d + 1] > bs[d] and all(fo[u] == u for u in range(bs[d], bs[d + 1]))) for d in range(n)]
    assert whole.cpu().tolist() == exp_whole and 0 < sum(exp_whole) < n
