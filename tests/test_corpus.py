"""CPU suite: the synthetic corpus generator -- determinism, chunk-independence and the optional features
(edit-rate sweep, token insert/delete, Zipf cluster sizes 2-50, boilerplate-heavy documents)."""
import hashlib

import numpy as np

from map_neo_amd import corpus


def _sha(t, o):
    return hashlib.sha256(t[: int(o[-1])].numpy().tobytes()).hexdigest()


def test_default_corpus_is_pinned():
    """the C2 bench corpus (defaults, seed of config 2) must never change silently: profiles/ and the issue model
    are tied to its byte and shingle counts"""
    spec = corpus.CorpusSpec(20000, 2048, seed=20240519 + 2, lang="en")
    t, o = corpus.generate(spec)
    assert int(o[-1]) == 43977347 and _sha(t, o).startswith("6f3db2316bdce8b1") and int(spec.is_dup.sum()) == 2996


def test_features_are_chunk_independent_and_shaped_as_stated():
    spec = corpus.CorpusSpec(30000, 600, seed=5, lang="mixed", edit_rates=(0.01, 0.05, 0.1, 0.2), indel_rate=0.01,
                             clusters="zipf50", boiler_frac=0.05, boiler_share=0.9, n_templates=50)
    t, o = corpus.generate(spec)
    for d0, d1 in ((0, 1000), (7000, 12345), (29000, 30000)):
        a, b = corpus.generate(spec, d0, d1)
        nb = int(o[d1] - o[d0])
        assert bytes(t[int(o[d0]):int(o[d1])].numpy()) == bytes(a[:nb].numpy())
        assert (b.numpy() == (o[d0:d1 + 1] - o[d0]).numpy()).all()
    # clusters: every source is an original, sizes (original + copies) within 2..50
    assert not spec.is_dup[spec.src].any()
    sizes = np.bincount(spec.src[spec.is_dup])
    sizes = sizes[sizes > 0] + 1
    assert 2 <= sizes.min() and sizes.max() <= 50 and (sizes == 2).sum() > (sizes == 10).sum()
    # near-copies keep the length of their source; edit rates come from the tuple
    assert (spec.ntok == spec.ntok[spec.src]).all()
    assert set(spec.rate_table.tolist()) == {int(r * 2 ** 32) for r in (0.01, 0.05, 0.1, 0.2)} and spec.rate_idx.max() == 3
    # boilerplate: ~5 % of the documents, popular templates more often than rare ones, copies share the template
    tpl = spec.template
    assert 0.03 < (tpl >= 0).mean() < 0.08 and (tpl == tpl[spec.src]).all()
    cnt = np.bincount(tpl[tpl >= 0], minlength=50)
    assert cnt[:5].sum() > cnt[-5:].sum()
    # two documents of one template share most of their text
    docs = np.nonzero((tpl == int(np.argmax(cnt))) & ~spec.is_dup & ~spec.is_zh)[0][:2]      # same template, same language
    x, y = (bytes(t[int(o[d]):int(o[d + 1])].numpy()) for d in docs)
    k = min(len(x), len(y)) // 2
    assert x[:k] == y[:k] and x != y


def test_indel_edits_shift_the_token_stream():
    """a near-copy with insert/delete edits still shares long runs of tokens with its source, but not position by position"""
    spec = corpus.CorpusSpec(4000, 900, seed=9, lang="en", dup_frac=0.3, edit_rate=0.0, indel_rate=0.02)
    t, o = corpus.generate(spec)
    d = int(np.nonzero(spec.is_dup & (spec.ntok > 60))[0][0])
    s = int(spec.src[d])
    a = bytes(t[int(o[d]):int(o[d + 1])].numpy()).split(b" ")
    b = bytes(t[int(o[s]):int(o[s + 1])].numpy()).split(b" ")
    assert len(a) == len(b) and a != b
    same_pos = sum(x == y for x, y in zip(a, b))
    shifted = sum(x == y for x, y in zip(a, b[1:]))
    assert same_pos + shifted >= len(a) - 8 and shifted > 0
