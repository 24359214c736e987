"""GPU suite (-m gpu): the permute-phase filter of sig_kernel (mdd_ctx_set_filter_from).

The filter skips the high half of a*h+b for hashes that provably cannot lower any
permutation's minimum; it must never change a signature.  Every test compares the HIP path,
through the C ABI, with the CPU oracle under several filter settings: default (filtering
starts after 256 shingles per 64 permutations), 1 (from the first shingle), an odd start,
and never."""
import numpy as np
import pytest

from map_neo_amd import corpus
from map_neo_amd.minhash import as_u32, pack_texts
from oracle import ref_c as RC

pytestmark = pytest.mark.gpu

NEVER = 0xFFFFFFFF
SETTINGS = (0, 1, 200, NEVER)


def _dev(engine, docs):
    buf, offs = pack_texts(docs)
    return engine.to_device(buf), engine.to_device(offs), buf, offs


def _all_settings(engine, fn):
    out = {}
    try:
        for s in SETTINGS:
            engine.set_filter_from(s)
            out[s] = fn()
    finally:
        engine.set_filter_from(0)
    return out


def test_filter_golden(engine, mini_docs, mini_expected):
    """golden mini-corpus, reference configuration and the K = 256 / word extensions"""
    text, offs_d, _, _ = _dev(engine, mini_docs)
    for name, K, width, unit in (("sig_char13_K256", 256, 13, "char"), ("sig_word5_K128", 128, 5, "word"),
                                 ("sig_char20_K64", 64, 20, "char")):
        got = _all_settings(engine, lambda: as_u32(engine.signatures(text, offs_d, K, width, unit)))
        for s, sig in got.items():
            assert (sig == mini_expected[name]).all(), (name, s)


def test_filter_mixed_lengths_vs_oracle(engine):
    """ragged documents from 0 to 40 000 code points (1..4-byte utf-8), so that work items end
    before, at and long after the point where filtering starts; several K / width / unit"""
    rng = np.random.Generator(np.random.PCG64(77))
    pools = [(0x20, 0x7F), (0x80, 0x7FF), (0x800, 0xD7FF), (0x10000, 0x10FFFF)]
    lens = [0, 5, 13, 14, 76, 77, 268, 269, 524, 525, 1036, 1037, 2500, 6000, 40000] + [int(x) for x in rng.integers(100, 5000, 60)]
    docs = []
    for i, n in enumerate(lens):
        if i % 3 == 0:
            cps = rng.integers(0x61, 0x7B, n)
            cps[rng.random(n) < 0.15] = 0x20
        else:
            mix = rng.integers(0, 4, n)
            cps = np.array([int(rng.integers(*pools[m])) for m in mix], dtype=np.int64)
        docs.append("".join(chr(int(c)) for c in cps))
    text, offs_d, buf, offs = _dev(engine, docs)
    for K, width, unit in ((128, 13, "char"), (256, 13, "char"), (64, 13, "char"), (100, 20, "char"), (128, 3, "word"), (192, 5, "char")):
        exp = RC.signatures(buf, offs, K, width, unit)
        got = _all_settings(engine, lambda: as_u32(engine.signatures(text, offs_d, K, width, unit)))
        for s, sig in got.items():
            assert (sig == exp).all(), (K, width, unit, s)


def test_filter_crafted_thresholds(engine):
    """Permutations with a = 1, so that t' = h + b + 1 and lo32(t') can be placed at will for one
    chosen shingle deep inside a long document (after filtering has started): just below / at /
    above the running minimum, at the top of the 32-bit range where lo32(t') + (t' >> 61) wraps
    (those must reach the full path and the exact pass), with t' >> 61 = 0 and = 7."""
    rng = np.random.Generator(np.random.PCG64(5))
    body = "".join(chr(int(c)) for c in rng.integers(0x61, 0x7B, 3000))
    docs = [body, body[:1500], "abcdefghijklm"]
    text, offs_d, buf, offs = _dev(engine, docs)
    hs = [int(h) for h in RC.shingle_hashes(body.encode(), 13, "char")]
    K = 128
    a = np.ones(K, np.uint64)
    b = np.zeros(K, np.uint64)
    targets = [2 ** 32 - 1, 2 ** 32 - 2, 2 ** 32 - 7, 2 ** 32 - 8, 2 ** 32 - 9, 0, 1, 6, 7, 8, 9, 10, 15, 16, 17, 1000]
    for i in range(K):
        pos = 700 + 17 * i                      # the chosen shingle: well past the default start (512)
        h = hs[pos]
        x = targets[i % len(targets)]           # wanted lo32(t') for that shingle
        c = (0, 7, 3, 1)[(i // len(targets)) % 4]
        low = (x - h - 1) % (1 << 32)           # lo32(b): h + low + 1 == x (mod 2^32); a carry into bit 32 is part of the case
        b[i] = (c << 61) | low
    a[120:] = rng.integers(1, 1 << 61, 8, dtype=np.uint64)          # a few ordinary ones next to them
    b[120:] = rng.integers(0, 1 << 61, 8, dtype=np.uint64)
    perms = (engine.to_device(a), engine.to_device(b))
    exp = RC.signatures(buf, offs, K, 13, "char", perms=(a, b))
    assert (exp[0] <= 8).any() and (exp[0] > 8).any()
    got = _all_settings(engine, lambda: as_u32(engine.signatures(text, offs_d, K, 13, "char", perms=perms)))
    for s, sig in got.items():
        assert (sig == exp).all(), s


def test_filter_full_range_perms(engine):
    """a, b drawn from the whole uint64 range (top bits set: every value of t' >> 61 occurs)"""
    rng = np.random.Generator(np.random.PCG64(6))
    spec = corpus.CorpusSpec(300, 4096, lang="mixed", dup_frac=0.1)
    text, offs = corpus.generate(spec, device=engine.device)
    K = 128
    a = rng.integers(0, 1 << 64, K, dtype=np.uint64)
    b = rng.integers(0, 1 << 64, K, dtype=np.uint64)
    perms = (engine.to_device(a), engine.to_device(b))
    exp = RC.signatures(text.cpu().numpy(), offs.cpu().numpy().astype(np.uint64), K, 13, "char", perms=(a, b))
    got = _all_settings(engine, lambda: as_u32(engine.signatures(text, offs, K, 13, "char", perms=perms)))
    for s, sig in got.items():
        assert (sig == exp).all(), s


def test_filter_with_segments(engine):
    """long documents cut into small segments: each segment is its own work item with its own
    minima and its own filter start; the atomicMin merge must give the unsegmented result"""
    spec = corpus.CorpusSpec(60, 30000, lang="mixed", dup_frac=0.1)
    text, offs = corpus.generate(spec, device=engine.device)
    exp = RC.signatures(text.cpu().numpy(), offs.cpu().numpy().astype(np.uint64), 128, 13, "char")
    try:
        for seg in (1024, 4096, 0):
            engine.set_segment_bytes(seg)
            got = _all_settings(engine, lambda: as_u32(engine.signatures(text, offs, 128, 13, "char")))
            for s, sig in got.items():
                assert (sig == exp).all(), (seg, s)
    finally:
        engine.set_segment_bytes(0)


def test_filter_settings_agree_at_scale(engine):
    """200 k documents of the bench's generator (size-independent property: the setting never
    changes a signature); a sample is checked against the oracle as well"""
    spec = corpus.CorpusSpec(200000, 2048, lang="en", dup_frac=0.15)
    text, offs = corpus.generate(spec, device=engine.device)
    got = _all_settings(engine, lambda: engine.signatures(text, offs, 128, 13, "char").clone())
    base = got[NEVER]
    for s, sig in got.items():
        assert bool((sig == base).all()), s
    n = 400
    o = offs[: n + 1].cpu().numpy().astype(np.uint64)
    exp = RC.signatures(text[: int(o[-1])].cpu().numpy(), o, 128, 13, "char")
    assert (as_u32(base[:n]) == exp).all()


@pytest.mark.parametrize("unit,width", [("word", 5), ("char", 13)])
def test_batched_work_claims_change_nothing(engine, unit, width):
    """300 k short documents: the work counter is claimed in batches of up to 8 items here (sig_kernel: guided
    self-scheduling; a call on <= 16 k documents only ever takes single items).  Size-independent property: the rows
    equal those of small calls on slices of the same corpus; both ends are checked against the oracle as well"""
    from map_neo_amd.minhash import padded_size
    import torch

    spec = corpus.CorpusSpec(300000, 128, lang="en", dup_frac=0.1, seed=77)
    text, offs = corpus.generate(spec, device=engine.device)
    sig = engine.signatures(text, offs, 128, width, unit)
    n = offs.numel() - 1
    for lo in (0, 123457, n - 9000):
        hi = min(n, lo + 9000)
        b0, b1 = int(offs[lo].item()), int(offs[hi].item())
        t = torch.zeros(padded_size(b1 - b0), dtype=torch.uint8, device=engine.device)
        t[: b1 - b0] = text[b0:b1]
        part = engine.signatures(t, (offs[lo : hi + 1] - b0).contiguous(), 128, width, unit)
        assert bool((part == sig[lo:hi]).all()), (unit, lo)
    for lo in (0, n - 300):
        o = offs[lo : lo + 301].cpu().numpy().astype(np.uint64)
        exp = RC.signatures(text[int(o[0]) : int(o[-1])].cpu().numpy(), o - o[0], 128, width, unit)
        assert (as_u32(sig[lo : lo + 300]) == exp).all(), (unit, lo)
