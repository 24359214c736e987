"""GPU suite: EN normalisation on the device (ASCII by byte classes, everything else table-driven)
against the ORACLE's statement of EN:31-40 (oracle.ref_py.normalize_text; ftfy is absent here: NFC only)."""
import numpy as np
import pytest

from oracle import ref_py as R
from map_neo_amd.minhash import as_u64, pack_texts

pytestmark = pytest.mark.gpu


def _run(engine, docs):
    buf, offs = pack_texts(docs)
    t, o, f = engine.normalize_en(engine.to_device(buf), engine.to_device(offs))
    t, o, f = t.cpu().numpy(), as_u64(o), f.cpu().numpy()
    return [bytes(t[int(o[k]):int(o[k + 1])]) for k in range(len(docs))], f


def test_normalize_en_matches_python(engine):
    rng = np.random.Generator(np.random.PCG64(11))
    ws = [9, 10, 11, 12, 13, 28, 29, 30, 31, 32]
    docs = []
    for _ in range(3000):
        n = int(rng.integers(0, 400))
        kind = rng.random(n)
        c = np.where(kind < 0.25, rng.choice(ws, n), np.where(kind < 0.45, rng.integers(33, 127, n), rng.integers(1, 128, n)))
        docs.append(bytes(c.astype(np.uint8)))
    docs += [b"", b" ", b"!!!", b" \t\n", b"a", b" a ", b"A" * 64, b" " * 64 + b"x", b"x" + b" " * 63 + b" " * 64 + b"Y!",
             b"," * 63 + b" " + b"," * 64 + b"z", b"a" + b"," * 200 + b"b", b"a " + b"," * 200 + b" b", b"\x1c\x1d\x1e\x1f",
             b"a\x00b", b"Hello,  World!  ", bytes(range(1, 128)) * 3]
    got, flags = _run(engine, docs)
    assert not flags.any()
    for d, g in zip(docs, got):
        assert g == R.normalize_text(d.decode("ascii")).encode("ascii"), d


def _flagged_by_generator():
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import gen_unicode_tables as G
    maybe = G.maybe_set()
    return lambda c: G.flagged(c, maybe)


def test_non_ascii_documents_known_cases(engine):
    """round 3: non-ASCII documents are normalised on the device through the generated Unicode table; only documents
    holding a code point whose Python steps are not per-character come back flagged (and untouched)"""
    ok = ["Plain ASCII, here.", "na\u00efve caf\u00e9", "", "\u4e2d\u6587 \u6587\u6863", "x" * 200 + "\u00e9", "Tail  OK!",
          "\u201cCurly\u201d \u2018quotes\u2019 \u2013 and\u2014dashes\u2026", "A\u00a0B\u3000C\u0085D\u2028E\u2029F\u202fG\u205fH\u1680I",
          "\u00a0\u3000 lead and trail \u2009\u200a", "\u00c9COLE \u00dcBER STRA\u1e9eE \u0410\u0411\u0412 \u0391\u0392\u0393",
          "emoji \U0001f600!\U0001f680 astral \U00010400\U00010401 deseret", "\U00020000\U0002a6dd cjk ext b", "\u023a\u023e grow on lower",
          "\uff21\uff22 fullwidth", "\u2160\u2161 roman", "\u24b6\u24b7 circled", "\ua78d\ua7aa \u13a0\u13a1 far lower"]
    flagged = ["e\u0301 combining", "\u0130stanbul", "\u039f\u0394\u039f\u03a3 sigma", "\u212b angstrom", "\u2000 en quad", "\u1100\u1161 jamo",
               "\u0378 unassigned", "\u0345 ypogegrammeni", "ok ok ok " * 40 + "\u0301"]
    docs = ok + flagged
    got, flags = _run(engine, docs)
    assert flags.tolist() == [0] * len(ok) + [1] * len(flagged)
    for d, g in zip(ok, got):
        assert g == R.normalize_text(d).encode("utf8"), d
    assert all(g == b"" for g in got[len(ok):])


def test_invalid_utf8_is_flagged(engine):
    bad = [b"abc\x80def", b"\xc3", b"tail \xe4\xb8", b"\xc0\x80 overlong", b"\xe0\x80\x80 overlong3", b"\xf0\x80\x80\x80 overlong4",
           b"\xed\xa0\x80 surrogate", b"\xf4\x90\x80\x80 beyond", b"\xf5\x80\x80\x80", b"\xff", b"\xc3\x28", b"\xe4\xb8\x28x",
           b"a" * 255 + b"\xc3", b"a" * 253 + b"\xf0\x9f\x98", b"\x80" + b"a" * 300, b"\xe4\xb8\xad\xad"]
    good = [b"fine", "caf\u00e9".encode(), b"a" * 255 + "\u00e9".encode(), b"a" * 253 + "\U0001f600".encode()]
    # a lead byte at the end of one document whose continuation bytes open the next: both are invalid on their own
    split = ["\u4e2d".encode()[:1], "\u4e2d".encode()[1:] + b" rest"]
    docs = bad + good + split
    got, flags = _run(engine, docs)
    assert flags.tolist() == [1] * len(bad) + [0] * len(good) + [1, 1]
    for d, g in zip(good, got[len(bad):len(bad) + len(good)]):
        assert g == R.normalize_text(d.decode("utf8")).encode("utf8")


def test_random_unicode_documents_vs_oracle(engine):
    """random BMP + astral documents: unflagged documents equal the oracle's normalize_text, and a document is flagged
    exactly when it holds a code point the table generator flags (no over-flagging)"""
    is_flagged = _flagged_by_generator()
    rng = np.random.Generator(np.random.PCG64(21))
    pools = [np.arange(0x20, 0x7f), np.arange(0xa0, 0x250), np.arange(0x370, 0x530), np.arange(0x2000, 0x2070), np.arange(0x3000, 0x3100),
             np.arange(0x4e00, 0x4f00), np.arange(0xac00, 0xac80), np.arange(0x1f600, 0x1f650), np.arange(0x10400, 0x10450),
             np.arange(0x1e900, 0x1e944), np.arange(0xff00, 0xff60), np.array([9, 10, 13, 0x85, 0xa0, 0x1680, 0x2028, 0x3000, 0x202f]),
             np.arange(0x300, 0x310), np.arange(0x1100, 0x1200), np.arange(0x2f800, 0x2f810), np.arange(0xe0100, 0xe0110), np.arange(0x1d00, 0x1e00)]
    docs = []
    for _ in range(4000):
        n = int(rng.integers(0, 300))
        k = rng.integers(1, 5)
        use = rng.choice(len(pools) - 5 if rng.random() < 0.8 else len(pools), k)     # 80 %: pools without combining marks / jamo / ...
        w = rng.random(len(use)) + (use == 0) * 3.0                                    # mostly ASCII with islands of the rest
        which = rng.choice(len(use), n, p=w / w.sum())
        cps = [int(rng.choice(pools[use[j]])) for j in which]
        docs.append("".join(chr(c) for c in cps if not 0xd800 <= c < 0xe000))
    got, flags = _run(engine, docs)
    n_flag = 0
    for d, g, f in zip(docs, got, flags.tolist()):
        want_flag = any(is_flagged(ord(ch)) for ch in d)
        assert bool(f) == want_flag, [hex(ord(ch)) for ch in d if is_flagged(ord(ch))]
        if f:
            n_flag += 1
            assert g == b""
        else:
            assert g == R.normalize_text(d).encode("utf8"), d
    assert 100 < n_flag < 3000


def test_code_points_across_lane_and_step_boundaries(engine):
    """multi-byte code points at every position of a 256-byte step (lane and step straddles), documents at every alignment"""
    docs = []
    for ch in ["\u00e9", "\u4e2d", "\U0001f600", "\u00a0", "\u3000", "\u00c9"]:
        for k in range(0, 264):
            docs.append("a" * k + ch + "B c" + ch + ch + " ." + ch)
    got, flags = _run(engine, docs)
    assert not flags.any()
    for d, g in zip(docs, got):
        assert g == R.normalize_text(d).encode("utf8"), d


def test_web_english_with_latin1_rarely_falls_back(engine):
    """corpus text with Latin-1 punctuation / accents / typographic quotes injected into half of the documents: fewer
    than 1 % of the documents go back to the host, and signatures_en hashes all of them right"""
    import torch
    from map_neo_amd import corpus
    from oracle import ref_c as RC
    spec = corpus.CorpusSpec(3000, 1500, seed=5, lang="en")
    text, offs = corpus.generate(spec, device="cpu")
    t, o = text.numpy(), offs.numpy()
    rng = np.random.Generator(np.random.PCG64(6))
    inject = ["\u2019", "\u201c", "\u201d", "\u2014", "\u2013", "\u00e9", "\u00e8", "\u00fc", "\u00a0", "\u2026", "\u00c9", "\u00df", "\u00b7", "\u00ab", "\u00bb"]
    docs = []
    for d in range(spec.n_docs):
        sdoc = bytes(t[o[d]:o[d + 1]]).decode("ascii")
        if d % 2 == 0 and len(sdoc) > 4:
            cs = list(sdoc)
            for pos in sorted(rng.integers(0, len(cs), max(1, len(cs) // 60)).tolist(), reverse=True):
                cs.insert(pos, inject[int(rng.integers(0, len(inject)))])
            if d % 200 == 0:
                cs.insert(len(cs) // 2, "e\u0301")          # a decomposed accent: the one kind that must go back
            sdoc = "".join(cs).title() if d % 4 == 0 else "".join(cs)
        docs.append(sdoc)
    buf, offs2 = pack_texts(docs)
    tt, oo = engine.to_device(buf), engine.to_device(offs2)
    _, _, flags = engine.normalize_en(tt, oo)
    n_flag = int(flags.sum().item())
    assert n_flag == len([d for d in range(spec.n_docs) if d % 200 == 0 and len(docs[d]) > 4]) and n_flag < 0.01 * spec.n_docs
    sig, n_host = engine.signatures_en(tt, oo)
    assert n_host == n_flag
    nbuf, noffs = pack_texts([R.normalize_text(d) for d in docs])
    assert (sig.cpu().numpy().view(np.uint32) == RC.signatures(nbuf, noffs)).all()


def test_long_document_and_full_pipeline_equivalence(engine):
    """normalise on device + signatures == signatures of the Python-normalised text"""
    from oracle import ref_c as RC
    rng = np.random.Generator(np.random.PCG64(12))
    words = ["The", "QUICK,", "brown;", "fox", "jumps!", "over", "(the)", "lazy", "dog.", "\t", "\n\n", "  "]
    docs = [" ".join(rng.choice(words, int(rng.integers(0, 3000)))) for _ in range(40)]
    buf, offs = pack_texts(docs)
    t, o, f = engine.normalize_en(engine.to_device(buf), engine.to_device(offs))
    sig = engine.signatures(t, o).cpu().numpy().view(np.uint32)
    nbuf, noffs = pack_texts([R.normalize_text(d) for d in docs])
    exp = RC.signatures(nbuf, noffs)
    assert (sig == exp).all()


def test_gather_docs_matches_python(engine):
    """mdd_gather_docs (row movement between the EN steps): every source / destination alignment, empty documents,
    repeats, reversed order, one long document; against plain Python slicing"""
    import torch
    rng = np.random.Generator(np.random.PCG64(17))
    docs = [bytes(rng.integers(1, 255, int(n), dtype=np.uint8)) for n in
            list(range(0, 40)) + [63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 4099, 70001] + rng.integers(0, 3000, 300).tolist()]
    buf, offs = pack_texts(docs)
    text, o = engine.to_device(buf), engine.to_device(offs)
    n = len(docs)
    for sel in (np.arange(n), np.arange(n)[::-1].copy(), rng.permutation(n), rng.integers(0, n, 1000), np.array([n - 1] * 5 + [0, 0]),
                np.zeros(0, np.int64), np.array([0])):
        sel = sel.astype(np.int64)
        t2, o2 = engine.gather_docs(text, o, engine.to_device(sel))
        o2h = o2.cpu().numpy()
        raw = t2.cpu().numpy().tobytes()
        assert o2h.shape[0] == sel.shape[0] + 1 and o2h[0] == 0
        assert [raw[o2h[k]:o2h[k + 1]] for k in range(sel.shape[0])] == [docs[i] for i in sel.tolist()]
        assert not any(raw[o2h[-1]:])                                            # the padding stays zero
    with pytest.raises(IndexError):
        engine.gather_docs(text, o, engine.to_device(np.array([n], dtype=np.int64)))
    with pytest.raises(IndexError):
        engine.gather_docs(text, o, engine.to_device(np.array([-1], dtype=np.int64)))
    # through the C ABI with a capacity smaller than the selection: nothing is written at or beyond it
    from map_neo_amd.minhash import _ptr
    sel = engine.to_device(np.arange(n, dtype=np.int64))
    lens = (o[1:] - o[:-1])
    oo = torch.zeros(n + 1, dtype=torch.int64, device=engine.device)
    torch.cumsum(lens, 0, out=oo[1:])
    total = int(oo[-1].item())
    cap = total // 2 + 1
    out = torch.full((total + 128,), 0xEE, dtype=torch.uint8, device=engine.device)
    rc = engine.L.mdd_gather_docs(engine.ctx, _ptr(text), total, _ptr(o), n, _ptr(sel), n, _ptr(oo), _ptr(out), cap)
    assert rc == 0
    engine.sync()
    outh = out.cpu().numpy()
    assert (outh[cap:] == 0xEE).all() and outh[:cap].tobytes() == b"".join(docs)[:cap]


def test_unaligned_buffers_take_the_byte_per_lane_kernels_and_agree(engine):
    """the dword-per-lane count / write passes need 4-byte aligned buffers (every device allocation is); a buffer that
    starts at an odd address takes the byte-per-lane kernels -- same results: EN normalisation, the paragraph count (the
    stage as a whole needs 8-byte aligned text for mdd_exact_group), and the substring compaction with an unaligned mark array"""
    import ctypes as C
    import torch
    from map_neo_amd import lib as _lib
    from map_neo_amd.minhash import TEXT_PAD
    rng = np.random.Generator(np.random.PCG64(21))
    words = ["The", "QUICK,", "brown;", "fox", "jumps!", "over", "(the)", "lazy", "dog.", "\t", "\n", "  ", "Zebra"]
    docs = [" ".join(rng.choice(words, int(rng.integers(0, 120)))) for _ in range(400)]
    docs += [docs[3], docs[5], "", "x"]
    buf, offs = pack_texts(docs)
    text, o = engine.to_device(buf), engine.to_device(offs)

    def shifted(t, k):                                  # the same bytes at (aligned address + k)
        big = torch.zeros(t.numel() + 8, dtype=torch.uint8, device=t.device)
        view = big[k: k + t.numel()]
        view.copy_(t)
        assert view.data_ptr() % 4 == k
        return view

    for k in (1, 2, 3):
        tu = shifted(text, k)
        a, b = engine.normalize_en(text, o), engine.normalize_en(tu, o)
        n = int(a[1][-1].item())
        assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[0][:n], b[0][:n]) and n > 0
        per_a = torch.zeros(len(docs), dtype=torch.int64, device=engine.device)
        per_b = torch.zeros_like(per_a)
        tb_ = text.numel() - TEXT_PAD
        _lib.check(engine.L.mdd_para_count(engine.ctx, C.c_void_p(text.data_ptr()), tb_, C.c_void_p(o.data_ptr()), len(docs),
                                           C.c_void_p(per_a.data_ptr())))
        _lib.check(engine.L.mdd_para_count(engine.ctx, C.c_void_p(tu.data_ptr()), tb_, C.c_void_p(o.data_ptr()), len(docs),
                                           C.c_void_p(per_b.data_ptr())))
        assert torch.equal(per_a, per_b) and int(per_a.sum().item()) > len(docs)
    # substring compaction: marks from the table method, then count + write with the mark array at an odd address
    nd = len(docs)
    tb = text.numel() - TEXT_PAD
    p = lambda t: C.c_void_p(t.data_ptr())
    dup = torch.zeros(tb, dtype=torch.uint8, device=engine.device)
    table = torch.empty(4 * tb + 16, dtype=torch.int64, device=engine.device)
    _lib.check(engine.L.mdd_substr_mark(engine.ctx, p(text), tb, p(o), nd, 20, 0, 1, p(table), table.numel(), p(dup)))
    _lib.check(engine.L.mdd_substr_check(engine.ctx, None))
    assert int(dup.sum().item()) > 0
    outs = []
    for d in (dup, shifted(dup, 1), shifted(dup, 3)):
        kb = torch.zeros(nd, dtype=torch.int64, device=engine.device)
        kc = torch.zeros(nd, dtype=torch.int64, device=engine.device)
        _lib.check(engine.L.mdd_substr_count(engine.ctx, p(text), tb, p(o), nd, 20, p(d), p(kb), p(kc)))
        oo = torch.zeros(nd + 1, dtype=torch.int64, device=engine.device)
        torch.cumsum(kb, 0, out=oo[1:])
        total = int(oo[-1].item())
        out = torch.zeros(total + TEXT_PAD, dtype=torch.uint8, device=engine.device)
        keep = torch.ones(nd, dtype=torch.uint8, device=engine.device)
        _lib.check(engine.L.mdd_substr_write(engine.ctx, p(text), tb, p(o), nd, 20, p(d), p(keep), p(oo), p(out), total))
        outs.append((kb.clone(), kc.clone(), out[:total].clone()))
    for x in outs[1:]:
        assert all(torch.equal(u, v) for u, v in zip(outs[0], x))
    assert 0 < outs[0][2].numel() < tb
