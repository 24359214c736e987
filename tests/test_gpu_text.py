"""GPU suite: EN normalisation of pure-ASCII documents on the device
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


def test_non_ascii_documents_are_flagged_not_touched(engine):
    docs = ["Plain ASCII, here.", "naïve café", "", "中文 文档", "x" * 200 + "é", "Tail  OK!"]
    got, flags = _run(engine, docs)
    assert flags.tolist() == [0, 1, 0, 1, 1, 0]
    assert got == [b"plain ascii here", b"", b"", b"", b"", b"tail ok"]


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
