"""GPU suite (-m gpu): the HIP path, called through the C ABI, against the CPU oracle.
Bit-exact is the bar everywhere (integer / byte / index work)."""
import numpy as np
import pytest
import torch

from map_neo_amd import corpus
from map_neo_amd.dedup import dedup_local
from map_neo_amd.minhash import as_u32, as_u64, pack_texts
from oracle import ref_c as RC
from oracle import ref_py as R

pytestmark = pytest.mark.gpu


def _dev(engine, docs):
    buf, offs = pack_texts(docs)
    return engine.to_device(buf), engine.to_device(offs), buf, offs


def test_shingle_hashes_stage(engine, mini_docs):
    """a2+a3: every shingle's sha1_hash32, in order, for ASCII / CJK / emoji / mixed docs."""
    text, offs_d, buf, offs = _dev(engine, mini_docs)
    nbytes = int(offs[-1])
    idx = list(range(0, 20)) + list(range(55, 125, 7)) + [198, 199]
    for unit, width in (("char", 13), ("char", 1), ("char", 20), ("char", 40), ("word", 5), ("word", 1), ("word", 12)):
        for i in idx:
            b, e = int(offs[i]), int(offs[i + 1])
            got = engine.shingle_hashes(text, nbytes, b, e, width, unit)
            exp = RC.shingle_hashes(bytes(buf[b:e]), width, unit)
            assert got.shape == exp.shape, (unit, width, i)
            assert (got == exp).all(), (unit, width, i)


@pytest.mark.parametrize("name,K,width,unit", [("sig_char13_K256", 256, 13, "char"), ("sig_word5_K128", 128, 5, "word"),
                                               ("sig_char5_K128", 128, 5, "char"), ("sig_char20_K64", 64, 20, "char")])
def test_signatures_golden(engine, mini_docs, mini_expected, name, K, width, unit):
    text, offs_d, _, _ = _dev(engine, mini_docs)
    sig = as_u32(engine.signatures(text, offs_d, K, width, unit))
    assert (sig == mini_expected[name]).all()


def test_signatures_reference_config_and_prefix(engine, mini_docs, mini_expected):
    """(char, 13, K=128) is the reference's configuration; K=128 == first 128 columns of K=256."""
    text, offs_d, _, _ = _dev(engine, mini_docs)
    s128 = as_u32(engine.signatures(text, offs_d, 128, 13, "char"))
    assert (s128 == mini_expected["sig_char13_K256"][:, :128]).all()
    for K in (1, 63, 65, 100, 192, 200):
        sk = as_u32(engine.signatures(text, offs_d, K, 13, "char"))
        assert (sk == mini_expected["sig_char13_K256"][:, :K]).all(), K


def test_signatures_random_utf8(engine):
    """random code points of 1..4 utf-8 bytes, ragged lengths incl. 0 and < width, unaligned starts"""
    rng = np.random.Generator(np.random.PCG64(11))
    pools = [(0x20, 0x7F), (0x80, 0x7FF), (0x800, 0xD7FF), (0x10000, 0x10FFFF)]
    docs = []
    for i in range(300):
        n = int(rng.integers(0, 400)) if i % 5 else int(rng.integers(0, 15))
        mix = rng.integers(0, 4, n) if i % 3 else np.full(n, i % 4)
        cps = [int(rng.integers(*pools[m])) for m in mix]
        docs.append("".join(chr(c) for c in cps))
    text, offs_d, buf, offs = _dev(engine, docs)
    for K, width, unit in ((128, 13, "char"), (256, 13, "char"), (128, 3, "word"), (64, 30, "char")):
        got = as_u32(engine.signatures(text, offs_d, K, width, unit))
        exp = RC.signatures(buf, offs, K, width, unit)
        assert (got == exp).all(), (K, width, unit)


def test_signatures_synthetic_vs_oracle(engine):
    """seeded bilingual corpus with near-duplicates, 3000 docs ~1 KB (C oracle finishes in seconds)"""
    spec = corpus.CorpusSpec(3000, 1024, lang="mixed", dup_frac=0.2)
    text, offs = corpus.generate(spec, device=engine.device)
    sig = as_u32(engine.signatures(text, offs, 128))
    exp = RC.signatures(text.cpu().numpy(), offs.cpu().numpy().astype(np.uint64), 128)
    assert (sig == exp).all()


def test_exact_pass_forced(engine):
    """Permutations chosen so that x = (t & p) + (t >> 61) >= p (the rare modular-reduction
    branch) and so that the fast form's guard (min <= 8) fires: results must still be exact."""
    docs = ["abcdefghijklmnopqrstuvwxyz", "the quick brown fox jumps over the lazy dog", "abcdefghijklm"]
    text, offs_d, buf, offs = _dev(engine, docs)
    p = (1 << 61) - 1
    hs = [int(h) for h in RC.shingle_hashes(docs[0].encode(), 13, "char")]
    a = np.ones(128, np.uint64)
    b = np.zeros(128, np.uint64)
    for i in range(128):
        h = hs[i % len(hs)]
        k = i % 9                       # t = a*h+b = p + k - ...: land x in [p-1, p+7] and around the 2^61 boundaries
        a[i] = 1
        b[i] = (p - h + k - 1) % p
    a[100:] = np.array([(1 << 61) - 2 - i for i in range(28)], np.uint64)      # large a: wrap mod 2^64 + top bits set
    b[100:] = np.array([(1 << 61) - 2 - 3 * i for i in range(28)], np.uint64)
    perms = (engine.to_device(a), engine.to_device(b))
    got = as_u32(engine.signatures(text, offs_d, 128, 13, "char", perms=perms))
    exp = RC.signatures(buf, offs, 128, 13, "char", perms=(a, b))
    assert (got == exp).all()
    assert (exp <= 8).any()             # the guard really was exercised


def _partition(labels):
    out = {}
    for i, l in enumerate(labels.tolist()):
        out.setdefault(l, []).append(i)
    return out


@pytest.mark.parametrize("b,r,K", [(9, 13, 128), (14, 9, 128), (20, 12, 256)])
def test_dedup_golden(engine, mini_docs, mini_expected, b, r, K):
    """whole path on the golden mini-corpus: identical buckets, labels, removed set"""
    text, offs_d, _, _ = _dev(engine, mini_docs)
    n = len(mini_docs)
    out = dedup_local(engine, text, offs_d, K, 13, "char", b, r, want_buckets=True)
    labels = as_u32(out["labels"])
    assert (labels == mini_expected[f"labels_b{b}_r{r}_K{K}"]).all()
    removed = out["removed"].cpu().numpy().astype(bool)
    assert (removed == (labels != np.arange(n))).all()
    assert int(out["n_removed"].item()) == int(removed.sum())
    # buckets: record i = doc*bands + band; its bucket representative's doc must be the bucket's min doc
    rb = as_u64(out["rec_bucket"]).reshape(n, b)
    bmin = (rb // b).T
    assert (bmin == mini_expected[f"bucketmin_b{b}_r{r}_K{K}"]).all()
    assert out["n_collisions"] == 0
    # edges only connect documents of one component
    eu, ev = as_u32(out["edge_u"]), as_u32(out["edge_v"])
    assert (labels[eu] == labels[ev]).all()


@pytest.mark.parametrize("key_bits", [1, 4, 10])
def test_surrogate_collisions_resolved_exactly(engine, mini_docs, mini_expected, key_bits):
    """truncate the surrogate key so that different tuples share slots: result must not change"""
    text, offs_d, _, _ = _dev(engine, mini_docs)
    out = dedup_local(engine, text, offs_d, 128, 13, "char", 9, 13, key_bits=key_bits)
    assert out["n_collisions"] > 0
    assert (as_u32(out["labels"]) == mini_expected["labels_b9_r13_K128"]).all()


def test_dedup_synthetic_vs_oracle(engine):
    spec = corpus.CorpusSpec(20000, 300, lang="mixed", dup_frac=0.25, edit_rate=0.01, short_frac=0.01)
    text, offs = corpus.generate(spec, device=engine.device)
    for (b, r, K) in [(9, 13, 128), (20, 12, 256)]:
        out = dedup_local(engine, text, offs, K, 13, "char", b, r)
        sig = as_u32(out["sig"])
        exp_sig = RC.signatures(text.cpu().numpy(), offs.cpu().numpy().astype(np.uint64), K)
        assert (sig == exp_sig).all()
        exp_labels, _ = RC.cluster(exp_sig, b, r)
        assert (as_u32(out["labels"]) == exp_labels).all()
        assert int(out["n_removed"].item()) == int((exp_labels != np.arange(spec.n_docs)).sum()) > 0


def test_components_adversarial(engine):
    """union-find on long chains, stars and random graphs vs the oracle's labels"""
    rng = np.random.Generator(np.random.PCG64(5))
    n = 200000
    chain = np.stack([np.arange(1, 50000), np.arange(0, 49999)], 1)
    star = np.stack([np.arange(60001, 90000), np.full(29999, 60000)], 1)
    rnd = rng.integers(100000, n, (80000, 2))
    edges = np.concatenate([chain[::-1], star, rnd]).astype(np.uint32)
    edges = edges[rng.permutation(len(edges))]
    labels = engine.components_init(n)
    eu, ev = engine.to_device(np.ascontiguousarray(edges[:, 0])), engine.to_device(np.ascontiguousarray(edges[:, 1]))
    half = len(edges) // 2
    engine.components_union(labels, eu[:half], ev[:half])
    engine.components_union(labels, eu[half:], ev[half:])
    engine.components_finish(labels)
    exp = R.components_min_label(n, [(int(u), int(v)) for u, v in edges if u != v])
    assert (as_u32(labels) == exp).all()


def test_empty_and_degenerate_inputs(engine):
    for docs in ([], [""], ["", "", ""], ["x"], ["short", "short"]):
        text, offs_d, buf, offs = _dev(engine, docs)
        out = dedup_local(engine, text, offs_d)
        n = len(docs)
        sig = as_u32(out["sig"]).reshape(n, 128)
        assert (sig == 0xFFFFFFFF).all()
        lab = as_u32(out["labels"])
        assert (lab == 0).all()          # all-max signatures collide in every band: one component
        assert int(out["n_removed"].item()) == max(0, n - 1)


def test_full_size_properties(engine):
    """C2 at full size (1M docs ~2 KB, BASELINE.json configs[1]): size-independent properties instead of a full oracle run:
    exact copies get identical signatures and one component; K=128 is a prefix of K=256;
    a sample of documents matches the oracle bit for bit; n_removed = sum(|component|-1)."""
    spec = corpus.CorpusSpec(1_000_000, 2048, lang="en", dup_frac=0.15, edit_rate=0.0)   # edit_rate 0: exact copies
    text, offs = corpus.generate_chunked(spec, engine.device)
    out = dedup_local(engine, text, offs)
    sig = out["sig"]
    src = torch.from_numpy(spec.src).to(engine.device)
    assert torch.equal(sig, sig[src])
    labels = out["labels"].to(torch.int64) & 0xFFFFFFFF
    assert torch.equal(labels, labels[src])
    assert bool((labels <= torch.arange(spec.n_docs, device=engine.device)).all())
    n_comp = int((labels == torch.arange(spec.n_docs, device=engine.device)).sum().item())
    assert int(out["n_removed"].item()) == spec.n_docs - n_comp
    assert int(out["n_removed"].item()) >= int(spec.is_dup.sum())
    s256 = engine.signatures(text, offs, 256)
    assert torch.equal(s256[:, :128], sig)
    sample = np.random.Generator(np.random.PCG64(1)).choice(spec.n_docs, 400, replace=False)
    o = offs.cpu().numpy()
    th = text.cpu().numpy()
    docs = [bytes(th[o[i]:o[i + 1]]) for i in sample]
    buf, so = pack_texts(docs)
    exp = RC.signatures(buf, so, 128)
    assert (as_u32(sig[torch.from_numpy(sample).to(engine.device)]) == exp).all()


def test_offsets_beyond_4_gib(engine):
    """2.2 M mixed EN/ZH documents = 4.6 GB of text: byte offsets pass 2^32.  Exact copies must get identical
    signatures and share a component, and a sample of documents from BEHIND the 4 GiB mark matches the oracle."""
    spec = corpus.CorpusSpec(2_200_000, 2048, lang="mixed", dup_frac=0.15, edit_rate=0.0, seed=77)
    text, offs = corpus.generate_chunked(spec, engine.device)
    assert int(offs[-1].item()) > (1 << 32) + (1 << 27)
    out = dedup_local(engine, text, offs)
    sig = out["sig"]
    src = torch.from_numpy(spec.src).to(engine.device)
    assert torch.equal(sig, sig[src])
    labels = out["labels"].to(torch.int64) & 0xFFFFFFFF
    assert torch.equal(labels, labels[src])
    n_comp = int((labels == torch.arange(spec.n_docs, device=engine.device)).sum().item())
    assert int(out["n_removed"].item()) == spec.n_docs - n_comp
    o = offs.cpu().numpy()
    first_high = int(np.searchsorted(o, 1 << 32))
    sample = np.random.Generator(np.random.PCG64(2)).choice(np.arange(first_high, spec.n_docs), 300, replace=False)
    docs = [bytes(text[int(o[i]):int(o[i + 1])].cpu().numpy()) for i in sample]
    buf, so = pack_texts(docs)
    exp = RC.signatures(buf, so, 128)
    assert (as_u32(sig[torch.from_numpy(sample).to(engine.device)]) == exp).all()


@pytest.mark.parametrize("world", [2, 8])
def test_routed_packed_path(engine, mini_docs, mini_expected, world):
    """multi-GPU data path on one GPU: route band records to `world` owners (mdd_route_count /
    mdd_route_pack), bucket each owner's PACKED records separately (mdd_packed_keys +
    mdd_bucket_edges), union all edges: labels must equal the one-GPU result."""
    text, offs_d, _, _ = _dev(engine, mini_docs)
    n = len(mini_docs)
    bands, rows, K = 9, 13, 128
    sig = engine.signatures(text, offs_d, K)
    key, meta = engine.band_keys(sig, bands, rows, 0)
    tuples, meta_s, counts = engine.route(sig, key, meta, bands, rows, world)
    counts = counts.tolist()
    assert sum(counts) == n * bands
    # every record lands in the segment of rank (key >> 32) % world, with its own tuple
    kh = as_u64(key)
    dest = ((kh >> np.uint64(32)) % np.uint64(world)).astype(np.int64)
    assert np.bincount(dest, minlength=world).tolist() == counts
    sig_h = as_u32(sig)
    tup_h = as_u32(tuples).reshape(-1, rows)
    meta_h = as_u64(meta_s)
    for i in range(0, n * bands, 7):
        d, j = int(meta_h[i] & np.uint64(0xFFFFFFFF)), int(meta_h[i] >> np.uint64(56))
        assert (tup_h[i] == sig_h[d, j * rows:(j + 1) * rows]).all()
    labels = engine.components_init(n)
    start = 0
    for r in range(world):
        c = counts[r]
        t_r = tuples[start * rows:(start + c) * rows]
        m_r = meta_s[start:start + c]
        k_r = engine.packed_keys(t_r, m_r, rows)
        eu, ev, n_coll, _ = engine.bucket_edges(k_r, m_r, t_r, 1, rows, 0, rows)
        assert n_coll == 0
        engine.components_union(labels, eu, ev)
        start += c
    engine.components_finish(labels)
    assert (as_u32(labels) == mini_expected["labels_b9_r13_K128"]).all()


@pytest.mark.parametrize("seg", [256, 512, 1024])
def test_long_document_segments_forced(engine, mini_docs, mini_expected, seg):
    """cut documents into tiny segments (several waves + atomicMin per document): results must
    not change -- ASCII, CJK, emoji and mixed documents, char and word units"""
    text, offs_d, _, _ = _dev(engine, mini_docs)
    engine.set_segment_bytes(seg)
    try:
        for name, K, width, unit in [("sig_char13_K256", 256, 13, "char"), ("sig_word5_K128", 128, 5, "word"),
                                     ("sig_char5_K128", 128, 5, "char"), ("sig_char20_K64", 64, 20, "char")]:
            sig = as_u32(engine.signatures(text, offs_d, K, width, unit))
            assert (sig == mini_expected[name]).all(), (seg, name)
    finally:
        engine.set_segment_bytes(0)


def test_long_documents_default_segments(engine):
    """documents of 20 KB .. 400 KB (C5-like and beyond) with the default segment length"""
    rng = np.random.Generator(np.random.PCG64(17))
    words = ["alpha", "beta", "gamma", "delta", "epsilon", "zeta", "eta", "theta", "iota", "kappa", "lambda", "mu"]
    cjk = [chr(0x4E00 + i) for i in range(3000)]
    docs = []
    for i in range(24):
        n = int(rng.integers(20_000, 400_000))
        if i % 3 == 0:
            docs.append(" ".join(rng.choice(words, n // 6)))
        elif i % 3 == 1:
            docs.append("".join(rng.choice(cjk, n // 3)))
        else:
            parts = []
            for _ in range(n // 40):
                parts.append(" ".join(rng.choice(words, 3)) if rng.random() < 0.5 else "".join(rng.choice(cjk, 8)) + "😀")
            docs.append(" ".join(parts))
    docs += ["short one", docs[0], docs[1][:50_000]]
    text, offs_d, buf, offs = _dev(engine, docs)
    for K, width, unit in ((128, 13, "char"), (128, 5, "word")):
        got = as_u32(engine.signatures(text, offs_d, K, width, unit))
        exp = RC.signatures(buf, offs, K, width, unit)
        assert (got == exp).all(), (K, width, unit)
    for (b, r, K) in [(9, 13, 128), (14, 9, 128), (20, 12, 256)]:       # C5 band sweep
        out = dedup_local(engine, text, offs_d, K, 13, "char", b, r)
        exp_sig = RC.signatures(buf, offs, K, 13, "char")
        exp_labels, _ = RC.cluster(exp_sig, b, r)
        assert (as_u32(out["labels"]) == exp_labels).all()


def test_c1_word5_config(engine):
    """BASELINE.json configs[0]: 10k synthetic English docs, 5-gram words, num_perm=128 -- GPU vs the CPU path"""
    spec = corpus.CorpusSpec(10000, 2048, seed=20240519 + 1, lang="en")
    text, offs = corpus.generate(spec, device=engine.device)
    out = dedup_local(engine, text, offs, 128, 5, "word", 9, 13)
    exp_sig = RC.signatures(text.cpu().numpy(), offs.cpu().numpy().astype(np.uint64), 128, 5, "word")
    assert (as_u32(out["sig"]) == exp_sig).all()
    exp_labels, _ = RC.cluster(exp_sig, 9, 13)
    assert (as_u32(out["labels"]) == exp_labels).all()


def test_host_pipeline_matches_resident(engine):
    """copy/compute overlap path: corpus in pinned host memory, batches over PCIe on a copy stream"""
    from map_neo_amd.minhash import HostPipeline
    spec = corpus.CorpusSpec(30000, 700, lang="mixed", dup_frac=0.1)
    text, offs = corpus.generate(spec, device=engine.device)
    ref = engine.signatures(text, offs, 128)
    host = text.cpu().pin_memory()
    pipe = HostPipeline(engine, batch_bytes=1 << 20)          # ~20 batches
    got = pipe.signatures(host, offs.cpu().numpy(), 128)
    torch.cuda.synchronize()
    assert torch.equal(got, ref)


def test_signatures_from_host_streamed(engine):
    """mdd_signatures_from_host: ONE launch hashes while the text arrives in chunks over PCIe (the kernel polls the byte
    count the copy stream publishes).  Mixed EN/ZH documents incl. long ones (segments), 1 MiB chunks (~20 of them),
    three calls in a row on the same context (the pinned value table alternates halves) -- always == resident."""
    spec = corpus.CorpusSpec(30000, 700, lang="mixed", dup_frac=0.1)
    text, offs = corpus.generate(spec, device=engine.device)
    long_docs = ["".join(chr(0x4E00 + (i * 7 + j) % 5000) for j in range(30000)) for i in range(3)] + ["word%d " % i * 9000 for i in range(3)]
    lbuf, loffs = pack_texts(long_docs)
    nb = int(offs[-1].item())
    full = torch.cat([text[:nb].cpu(), torch.from_numpy(lbuf)])
    offs_all = torch.cat([offs.cpu(), torch.from_numpy(loffs[1:].astype(np.int64)) + nb])
    tb = int(offs_all[-1].item())
    host = full[: tb].clone().pin_memory()
    dev_text = engine.to_device(np.concatenate([full[:tb].numpy(), np.zeros(((tb + 3) // 4) * 4 - tb + 128, np.uint8)]))
    offs_d = offs_all.to(engine.device)
    for K, width, unit in ((128, 13, "char"), (256, 13, "char"), (128, 5, "word")):
        ref = engine.signatures(dev_text, offs_d, K, width, unit, text_bytes=tb)
        got, tdev = engine.signatures_from_host(host, offs_d, K, width, unit, text_bytes=tb, chunk_bytes=1 << 20)  # ~25 chunks
        engine.sync()
        assert torch.equal(got, ref), (K, width, unit)
        assert torch.equal(tdev[:tb], dev_text[:tb])
    # empty corpus and a corpus smaller than the first chunk
    e, _ = engine.signatures_from_host(torch.zeros(16, dtype=torch.uint8).pin_memory(), torch.zeros(1, dtype=torch.int64, device=engine.device))
    assert e.shape == (0, 128)
    buf, o = pack_texts(["the quick brown fox jumps over the lazy dog", "", "abcdefghijklm"])
    got, _ = engine.signatures_from_host(torch.from_numpy(buf).pin_memory(), engine.to_device(o))
    engine.sync()
    assert torch.equal(got, engine.signatures(engine.to_device(buf), engine.to_device(o)))


def test_forest_edges(engine):
    """forest of flattened labels reproduces the same components with at most one edge per document"""
    rng = np.random.Generator(np.random.PCG64(9))
    n = 50000
    edges = rng.integers(0, n, (30000, 2)).astype(np.uint32)
    labels = engine.components_init(n)
    eu, ev = engine.to_device(np.ascontiguousarray(edges[:, 0])), engine.to_device(np.ascontiguousarray(edges[:, 1]))
    engine.components_union(labels, eu, ev)
    engine.components_finish(labels)
    fu, fv = engine.forest_edges(labels, len(edges))
    lab = as_u32(labels)
    assert fu.numel() == int((lab != np.arange(n)).sum()) <= len(edges)
    l2 = engine.components_init(n)
    engine.components_union(l2, fu, fv)
    engine.components_finish(l2)
    assert (as_u32(l2) == lab).all()
    assert (lab == R.components_min_label(n, [(int(u), int(v)) for u, v in edges if u != v])).all()


def test_distributed_steps_with_an_empty_shard(engine):
    """a rank without documents (more ranks than shards) walks the same calls with zero records"""
    sig = torch.empty((0, 128), dtype=torch.int32, device=engine.device)
    key, meta = engine.band_keys(sig, 9, 13, 5)
    assert key.numel() == 0 and meta.numel() == 0
    tuples, meta_s, counts = engine.route(sig, key, meta, 9, 13, 8)
    assert counts.tolist() == [0] * 8 and tuples.numel() == 0
    k = engine.packed_keys(tuples, meta_s, 13)
    eu, ev, n_coll, _ = engine.bucket_edges(k, meta_s, tuples, 1, 13, 0, 13)
    assert eu.numel() == 0 and n_coll == 0
    labels = engine.components_init(10)
    engine.components_union(labels, eu, ev)
    engine.components_finish(labels)
    fu, fv = engine.forest_edges(labels, 0)
    assert fu.numel() == 0 and as_u32(labels).tolist() == list(range(10))
    removed, n_removed = engine.removed_mask(labels, 10, 0)
    assert removed.numel() == 0 and int(n_removed.item()) == 0


def test_segmented_long_documents_equal_unsegmented_at_scale(engine):
    """20 k documents of ~20 KB (C5 shape): default 16 KiB segments (several waves + atomicMin per document) vs one
    wave per document -- identical signatures; K = 256 is checked as well"""
    spec = corpus.CorpusSpec(20_000, 20_480, lang="mixed", seed=9)
    text, offs = corpus.generate_chunked(spec, engine.device, docs_per_chunk=2048)
    for K in (128, 256):
        a = engine.signatures(text, offs, K)
        engine.set_segment_bytes(1 << 24)
        try:
            b = engine.signatures(text, offs, K)
        finally:
            engine.set_segment_bytes(0)
        assert torch.equal(a, b)
    assert int((offs[1:] - offs[:-1]).max().item()) > 16384


def test_routed_path_equals_local_path_at_scale(engine):
    """300 k documents: band records routed to 8 owners, bucketed per owner in PACKED layout and merged through
    spanning forests == the single-GPU bucket stage (labels, hence removed set)"""
    spec = corpus.CorpusSpec(300_000, 1024, lang="mixed", seed=13)
    text, offs = corpus.generate_chunked(spec, engine.device)
    out = dedup_local(engine, text, offs)
    sig, n = out["sig"], spec.n_docs
    key, meta = engine.band_keys(sig, 9, 13, 0)
    tuples, meta_s, counts = engine.route(sig, key, meta, 9, 13, 8)
    labels = engine.components_init(n)
    start = 0
    for c in counts.tolist():
        t_r, m_r = tuples[start * 13:(start + c) * 13], meta_s[start:start + c]
        eu, ev, _, _ = engine.bucket_edges(engine.packed_keys(t_r, m_r, 13), m_r, t_r, 1, 13, 0, 13)
        lab = engine.components_init(n)                   # per-owner forest, as cluster_distributed ships it
        engine.components_union(lab, eu, ev)
        engine.components_finish(lab)
        fu, fv = engine.forest_edges(lab, int(eu.numel()))
        engine.components_union(labels, fu, fv)
        start += c
    engine.components_finish(labels)
    assert torch.equal(labels, out["labels"])
    assert int(out["n_removed"].item()) > 0.3 * int(spec.is_dup.sum())     # LSH at ~0.8 Jaccard misses the more heavily edited copies


def test_single_huge_document(engine):
    """one 24 MB document (about 1 500 segments of 16 KiB) next to a 6 MB CJK one: segment size must not matter, and
    the exact substring stage must erase a verbatim second copy"""
    from map_neo_amd import exact
    rng = np.random.Generator(np.random.PCG64(91))
    words = np.array(["w%d" % i for i in range(5000)])
    big = " ".join(rng.choice(words, 4_000_000))                       # ~24 MB ASCII
    zh = "".join(chr(int(c)) for c in rng.integers(0x4E00, 0x9FFF, 2_000_000))     # 6 MB of 3-byte code points
    buf, offs = pack_texts([big, zh, "tiny"])
    text, o = engine.to_device(buf), engine.to_device(offs)
    ref = engine.signatures(text, o, 128)
    for seg in (4096, 1 << 20, 1 << 26):
        engine.set_segment_bytes(seg)
        try:
            assert torch.equal(engine.signatures(text, o, 128), ref), seg
        finally:
            engine.set_segment_bytes(0)
    assert torch.equal(engine.signatures(text, o, 256)[:, :128], ref)
    # substring dedup: [big, big] -> every window of the copy is a repeat; only its last code point is uncovered
    # (SUB:38 never hashes the last window), which is below min_length, so the copy goes (same on a small case by the oracle)
    from oracle import ref_exact
    small = big[:20_000]
    assert ref_exact.exact_substr_dedup([(small, "a"), (small, "b")], 50, 20) == [("a", small)]
    buf2, offs2 = pack_texts([big, big])
    t2, o2, docs = exact.exact_substr_dedup_packed(engine, engine.to_device(buf2), engine.to_device(offs2), 50, 20)
    assert docs.tolist() == [0]
    kept = bytes(t2[: int(o2[1].item())].cpu().numpy()).decode()
    assert len(big) - len(kept) < 5000 and kept[:1000] == big[:1000] and kept[-50:] == big[-50:]


def _skew_sig(dev, n, case, seed=7):
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import bucket_skew
    return bucket_skew.make_sig(dev, n, case, seed)


@pytest.mark.parametrize("case,n", [("one_bucket_1M", 1_100_000), ("100_buckets_10pct", 400_000), ("all_equal", 150_000)])
def test_bucket_skew_vs_oracle(engine, case, n):
    """BASELINE.json configs[4] (bucket skew): ONE bucket of 1 M records; 10 % of the records in 100 buckets; every document
    in one bucket per band -- band keys + buckets + union-find against the C oracle's clustering of the same signatures,
    and the bucket stage must not collapse on the hot entries (within 3x of the same number of singleton records + slack)"""
    from map_neo_amd.dedup import StageTimer, cluster_local
    sig = _skew_sig(engine.device, n, case)
    out = cluster_local(engine, sig, want_buckets=(n <= 400_000))
    res = RC.cluster(as_u32(sig), 9, 13, want_buckets=(n <= 400_000))
    exp_labels = res[0]
    assert (as_u32(out["labels"]) == exp_labels).all()
    assert int(out["n_removed"].item()) == int((exp_labels != np.arange(n)).sum()) > 0
    if n <= 400_000:                      # identical buckets: smallest document of every record's bucket
        rb = as_u64(out["rec_bucket"]).reshape(n, 9)
        exp_min = (res[2].reshape(9, n) % np.uint64(n)).T
        assert ((rb // np.uint64(9)) == exp_min).all()
    eu, ev = as_u32(out["edge_u"]), as_u32(out["edge_v"])
    assert (exp_labels[eu] == exp_labels[ev]).all()
    # timing: skewed vs the same number of records in singleton buckets
    uni = _skew_sig(engine.device, n, "uniform")
    ms = {}
    for name, s in (("uniform", uni), ("skew", sig)):
        cluster_local(engine, s)
        tm = StageTimer()
        tm.mark("start")
        cluster_local(engine, s, timer=tm)
        ms[name] = tm.result_ms()["bucket_edges"]
    assert ms["skew"] < 3 * ms["uniform"] + 1.0, ms


def test_c5_boilerplate_heavy_corpus_vs_oracle(engine):
    """BASELINE.json configs[4] in small: long documents, 8 % boilerplate-heavy ones (Zipf-popular templates -> skewed buckets),
    near-copies with an edit-rate sweep and token insertions / deletions, clusters of 2-50; all three (b, r, K) settings"""
    spec = corpus.CorpusSpec(6000, 5000, seed=20240519 + 5, lang="mixed", dup_frac=0.2, edit_rates=(0.01, 0.05, 0.1, 0.2),
                             indel_rate=0.003, clusters="zipf50", boiler_frac=0.08, boiler_share=0.92, n_templates=12)
    text, offs = corpus.generate_chunked(spec, engine.device, docs_per_chunk=2048)
    th, oh = text.cpu().numpy(), offs.cpu().numpy().astype(np.uint64)
    biggest = 0
    for (b, r, K) in [(9, 13, 128), (14, 9, 128), (20, 12, 256)]:
        out = dedup_local(engine, text, offs, K, 13, "char", b, r)
        exp_sig = RC.signatures(th, oh, K, 13, "char", threads=8)
        assert (as_u32(out["sig"]) == exp_sig).all()
        exp_labels, _ = RC.cluster(exp_sig, b, r)
        assert (as_u32(out["labels"]) == exp_labels).all()
        assert int(out["n_removed"].item()) == int((exp_labels != np.arange(spec.n_docs)).sum())
        biggest = max(biggest, int(np.bincount(exp_labels).max()))
    assert biggest >= 30          # the popular templates really formed big clusters


def test_signature_matches_and_min_jaccard_vs_oracle(engine):
    """mdd_signature_matches (K x MinHash.jaccard of candidate pairs; an extension) == the oracle's count,
    incl. K not a multiple of 64, self-pairs, all-max signatures and an out-of-range document; and dedup_local(min_jaccard=t)
    == the oracle's components over the edges whose count reaches ceil(t K)."""
    rng = np.random.Generator(np.random.PCG64(123))
    for K in (128, 256, 100, 1):
        n = 500
        base = rng.integers(0, 2 ** 32, (n, K), dtype=np.uint64).astype(np.uint32)
        for d in range(1, n, 3):                                  # every third row copies a share of its predecessor
            keep = rng.random(K) < rng.random()
            base[d, keep] = base[d - 1, keep]
        base[7] = 0xFFFFFFFF
        base[8] = 0xFFFFFFFF
        eu = rng.integers(0, n, 4000).astype(np.uint32)
        ev = np.where(rng.random(4000) < 0.5, np.maximum(eu, 1) - 1, rng.integers(0, n, 4000)).astype(np.uint32)
        eu[:3], ev[:3] = (7, 5, 9), (8, 5, 9)
        sig = engine.to_device(base)
        got = as_u32(engine.signature_matches(sig, engine.to_device(eu), engine.to_device(ev)))
        assert np.array_equal(got, R.signature_matches(base, eu, ev)), K
        assert got[0] == K and got[1] == K
    bad = engine.signature_matches(sig, engine.to_device(np.array([0, n], np.uint32)), engine.to_device(np.array([0, 0], np.uint32)))
    assert as_u32(bad).tolist() == [1, 0]
    assert engine.signature_matches(sig, engine.to_device(np.zeros(0, np.uint32)), engine.to_device(np.zeros(0, np.uint32))).numel() == 0
    # end to end: near-duplicates at several edit rates; the threshold sits above the weakest candidate pairs, so it does cut
    spec = corpus.CorpusSpec(3000, 600, lang="mixed", seed=77, dup_frac=0.3, edit_rates=(0.01, 0.05, 0.1, 0.2, 0.4))
    text, offs = corpus.generate(spec, device=engine.device)
    plain = dedup_local(engine, text, offs)
    sigs = as_u32(plain["sig"]).reshape(spec.n_docs, -1)
    eu, ev = as_u32(plain["edge_u"]), as_u32(plain["edge_v"])
    m = R.signature_matches(sigs, eu, ev)
    K = sigs.shape[1]
    assert int(m.min()) < K                                      # some candidate pair is not an exact signature match
    need = min(K, int(np.sort(m)[len(m) // 4]) + 1)               # above the lowest quarter, so the threshold does cut
    t = need / K
    res = dedup_local(engine, text, offs, min_jaccard=t)
    kept = [(int(u), int(v)) for u, v, c in zip(eu, ev, m) if c >= need]
    assert 0 < len(kept) < len(eu)
    assert np.array_equal(as_u32(res["labels"]), R.components_min_label(spec.n_docs, kept))
    assert int(res["n_removed"].item()) < int(plain["n_removed"].item())


@pytest.mark.parametrize("seg", [0, 256])
def test_word_scan_window_and_segment_boundaries(engine, seg):
    """The word-unit boundary scan works on 256-byte windows, one dword per lane, and carries three facts from window to
    window (the previous byte was a space, a token spans the window's first byte, a segment began inside a token):
    documents whose lengths sit around multiples of 256, with spaces sparse, dense and in runs, leading / trailing
    spaces, tokens longer than a window and 3-byte code points -- against the oracle, whole and cut into 256-byte segments"""
    rng = np.random.Generator(np.random.PCG64(4242))
    docs = []
    for ln in (0, 1, 3, 4, 5, 252, 255, 256, 257, 260, 508, 511, 512, 513, 516, 767, 768, 769, 1023, 1024, 1025, 2049):
        for p_space in (0.0, 0.02, 0.3, 0.7, 0.97):
            ch = np.where(rng.random(ln) < p_space, " ", np.where(rng.random(ln) < 0.1, "中", "a"))
            s = "".join(ch.tolist())
            docs.append(s)
            docs.append(" " + s + "  ")
    docs.append("a" * 700 + " b c d e f " + "g" * 300)
    text, offs_d, buf, offs = _dev(engine, docs)
    engine.set_segment_bytes(seg)
    try:
        for width in (1, 2, 5):
            got = as_u32(engine.signatures(text, offs_d, 64, width, "word"))
            exp = RC.signatures(buf, offs, 64, width, "word")
            bad = np.nonzero((got != exp).any(axis=1))[0]
            assert bad.size == 0, (seg, width, bad[:5], [len(docs[i]) for i in bad[:5]])
    finally:
        engine.set_segment_bytes(0)


@pytest.mark.parametrize("seg", [0, 256])
def test_char_scan_window_and_segment_boundaries(engine, seg):
    """The code-point scan of non-ASCII documents (SWAR start flags, ranks from three count planes): 1-, 2-, 3- and 4-byte
    code points in every mix, document lengths around multiples of the 256-byte window, whole and in 256-byte segments"""
    rng = np.random.Generator(np.random.PCG64(777))
    alphabet = np.array(["a", "é", "中", "😀"])
    docs = []
    for n_cp in (1, 12, 13, 14, 63, 64, 65, 85, 86, 127, 128, 129, 255, 256, 257, 300, 511, 512, 513, 1025):
        for probs in ((0.97, 0.01, 0.01, 0.01), (0.25, 0.25, 0.25, 0.25), (0.0, 0.0, 1.0, 0.0), (0.0, 0.0, 0.02, 0.98), (0.5, 0.0, 0.5, 0.0)):
            docs.append("".join(rng.choice(alphabet, n_cp, p=probs).tolist()))
    text, offs_d, buf, offs = _dev(engine, docs)
    engine.set_segment_bytes(seg)
    try:
        for width in (1, 13, 40):
            got = as_u32(engine.signatures(text, offs_d, 64, width, "char"))
            exp = RC.signatures(buf, offs, 64, width, "char")
            bad = np.nonzero((got != exp).any(axis=1))[0]
            assert bad.size == 0, (seg, width, bad[:5], [len(docs[i]) for i in bad[:5]])
    finally:
        engine.set_segment_bytes(0)
