"""GPU suite: seeded random sweep over (K, width, unit, bands, rows, segment length) and random
utf-8 documents; signatures and component labels must equal the C oracle bit for bit."""
import numpy as np
import pytest

from map_neo_amd.dedup import dedup_local
from map_neo_amd.minhash import as_u32, pack_texts
from oracle import ref_c as RC

pytestmark = pytest.mark.gpu

POOLS = [(0x20, 0x7F), (0x80, 0x7FF), (0x800, 0xD7FF), (0xE000, 0xFFFF), (0x10000, 0x10FFFF)]


def _docs(rng, n):
    base = []
    for _ in range(n):
        kind = rng.integers(0, 4)
        ln = int(rng.integers(0, 600)) if rng.random() < 0.8 else int(rng.integers(0, 20))
        if kind == 0:
            cps = rng.integers(0x61, 0x7B, ln)
            cps[rng.random(ln) < 0.18] = 0x20                       # words separated by (runs of) spaces
        elif kind == 1:
            cps = rng.integers(0x4E00, 0x9FFF, ln)
        else:
            pools = rng.integers(0, len(POOLS), ln)
            cps = np.array([rng.integers(*POOLS[p]) for p in pools], dtype=np.int64)
            cps[rng.random(ln) < 0.1] = 0x20
        base.append("".join(chr(int(c)) for c in cps))
    docs = list(base)
    for _ in range(n // 3):                                         # near / exact copies so that clusters exist
        t = base[int(rng.integers(0, n))]
        if len(t) > 30 and rng.random() < 0.7:
            k = int(rng.integers(0, len(t) - 1))
            t = t[:k] + "Z" + t[k + 1:]
        docs.append(t)
    return docs


@pytest.mark.parametrize("seed", range(12))
def test_random_configurations(engine, seed):
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    docs = _docs(rng, 150)
    K = int(rng.choice([1, 7, 64, 65, 100, 128, 129, 192, 200, 256]))
    unit = "char" if rng.random() < 0.6 else "word"
    width = int(rng.integers(1, 41)) if unit == "char" else int(rng.integers(1, 9))
    rows = int(rng.integers(1, min(K, 16) + 1))
    bands = int(rng.integers(1, min(64, K // rows) + 1))
    seg = int(rng.choice([0, 256, 768]))
    sd = int(rng.integers(0, 1 << 31))
    buf, offs = pack_texts(docs)
    engine.set_segment_bytes(seg)
    try:
        out = dedup_local(engine, engine.to_device(buf), engine.to_device(offs), K, width, unit, bands, rows, seed=sd)
    finally:
        engine.set_segment_bytes(0)
    exp_sig = RC.signatures(buf, offs, K, width, unit, seed=sd)
    assert (as_u32(out["sig"]).reshape(len(docs), K) == exp_sig).all(), (K, unit, width, seg)
    exp_labels, _ = RC.cluster(exp_sig, bands, rows)
    assert (as_u32(out["labels"]) == exp_labels).all(), (K, bands, rows)


@pytest.mark.parametrize("seed", range(8))
def test_random_exact_stages(engine, seed):
    """seeded random sweep over window length, minimum lengths, segment length and random utf-8 documents built
    from shared pieces (so that windows / paragraphs repeat): fused kernels == oracle restatement"""
    from map_neo_amd import exact
    from oracle import ref_exact as RX
    from oracle import ref_py as R
    rng = np.random.Generator(np.random.PCG64(2000 + seed))
    pieces = _docs(rng, 40)
    pieces = [p.replace("\n", " ") for p in pieces if p] or ["x"]
    docs = []
    for _ in range(80):
        k = int(rng.integers(0, 6))
        sep = "\n" if rng.random() < 0.7 else ""
        docs.append(sep.join(pieces[int(i)] for i in rng.integers(0, len(pieces), k)) + ("\n" if rng.random() < 0.2 else ""))
    window = int(rng.choice([1, 2, 13, 50, 50, 64, 128]))
    min_length = int(rng.choice([0, 1, 20, 200]))
    seg = int(rng.choice([0, 256, 1024]))
    engine.set_segment_bytes(seg)
    try:
        rows = [(d, i) for i, d in enumerate(docs)]
        assert exact.exact_substr_dedup(engine, rows, window, min_length) == RX.exact_substr_dedup(rows, window, min_length), \
            (window, min_length, seg)
    finally:
        engine.set_segment_bytes(0)
    ids = [int(i) for i in rng.permutation(len(docs))]
    prow = list(zip(ids, docs))
    assert exact.paragraph_dedup(engine, prow, min_length) == RX.paragraph_dedup(prow, min_length), min_length
    assert exact.exact_document_dedup(engine, docs).tolist() == RX.exact_document_dedup(docs)
    # device EN normalisation on the ASCII documents of the sweep
    ascii_docs = [d for d in docs if d.isascii()]
    if ascii_docs:
        buf, offs = pack_texts(ascii_docs)
        t, o, f = engine.normalize_en(engine.to_device(buf), engine.to_device(offs))
        raw, oo = t.cpu().numpy().tobytes(), o.cpu().numpy()
        assert not f.any().item()
        assert [raw[int(oo[k]):int(oo[k + 1])].decode() for k in range(len(ascii_docs))] == [R.normalize_text(d) for d in ascii_docs]


@pytest.mark.parametrize("unit,width", [("char", 128), ("char", 100), ("word", 128), ("word", 32)])
def test_maximum_widths(engine, unit, width):
    """the widest shingles the ABI accepts (ring occupancy 64 + width + 256 boundary entries; multi-block SHA-1)"""
    rng = np.random.Generator(np.random.PCG64(77 + width))
    docs = _docs(rng, 60) + ["a " * 700, "字" * 300, "x" * 127, "x" * 128, "x" * 129, " ".join(["w"] * 127), " ".join(["w"] * 129)]
    buf, offs = pack_texts(docs)
    for seg in (0, 256):
        engine.set_segment_bytes(seg)
        try:
            sig = engine.signatures(engine.to_device(buf), engine.to_device(offs), 64, width, unit)
        finally:
            engine.set_segment_bytes(0)
        assert (as_u32(sig).reshape(len(docs), 64) == RC.signatures(buf, offs, 64, width, unit)).all(), (unit, width, seg)
