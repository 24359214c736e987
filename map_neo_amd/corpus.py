"""Seeded synthetic corpora -- bench/test input, not part of the hot path.

Counter-based: every token of every document is a pure function of (seed, doc, position),
computed with wrapping int64 arithmetic in torch, so the same corpus is produced on CPU and
on the GPU, in any chunking.  Documents are generated already normalised (EN:31-40 is the
identity on them): lowercase a-z pseudo-words joined by single spaces (EN), or runs of CJK
code points (ZH, 3 utf-8 bytes each, no separators).

Near-duplicates: a fraction `dup_frac` of documents copies the token stream of a "source"
document and substitutes each token with probability `edit_rate` (clusters of several
near-copies form around popular sources).  A fraction `short_frac` of documents is shorter
than the shingle width (they all share the all-0xFFFFFFFF signature).

SURVEY 8(d) features that are OFF by default (the C2 bench corpus is the default and must not
change): `edit_rates` (each near-copy draws its rate from the tuple, e.g. (0.01, 0.05, 0.1, 0.2)),
`indel_rate` (token deletions + insertions: per near-copy round(indel_rate * tokens) pairs of a
deleted and an inserted token, so lengths are preserved), `clusters="zipf50"` (C4: near-copies in
clusters of 2-50 documents with Zipf-distributed sizes around random originals), and
`boiler_frac` / `boiler_share` / `n_templates` (C5: boilerplate-heavy documents -- a fraction of
the documents takes `boiler_share` of its tokens from one of `n_templates` shared templates
chosen with Zipf popularity, which produces the skewed buckets of cookie banners, licences and
error pages).
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np
import torch

VOCAB_EN = 50_000
VOCAB_ZH = 6_000
_M64 = (1 << 64) - 1


def _i64(x: int) -> int:
    x &= _M64
    return x - (1 << 64) if x >= (1 << 63) else x


_C1 = _i64(0xBF58476D1CE4E5B9)
_C2 = _i64(0x94D049BB133111EB)
_GOLD = _i64(0x9E3779B97F4A7C15)


def _lsr(x: torch.Tensor, s: int) -> torch.Tensor:
    return (x >> s) & ((1 << (64 - s)) - 1)


def mix64(x: torch.Tensor) -> torch.Tensor:
    """splitmix64 finaliser on int64 tensors (wrapping)."""
    x = (x ^ _lsr(x, 30)) * _C1
    x = (x ^ _lsr(x, 27)) * _C2
    return x ^ _lsr(x, 31)


def _h(seed: int, a: torch.Tensor, b) -> torch.Tensor:
    return mix64(mix64(a * _GOLD + _i64(seed)) ^ (b * _C2 + 0x632BE5AB))


def _zipf_thresholds(vocab: int) -> np.ndarray:
    w = 1.0 / np.arange(1, vocab + 1, dtype=np.float64)
    cdf = np.cumsum(w) / w.sum()
    thr = np.minimum(np.floor(cdf * 2.0 ** 32), 2.0 ** 32 - 1).astype(np.int64)
    thr[-1] = (1 << 32) - 1
    return thr


class CorpusSpec:
    def __init__(self, n_docs: int, mean_bytes: int = 2048, seed: int = 20240519, lang: str = "en",
                 dup_frac: float = 0.15, edit_rate: float = 0.02, short_frac: float = 0.001,
                 sigma: float = 0.6, edit_rates: Optional[Tuple[float, ...]] = None, indel_rate: float = 0.0,
                 clusters: str = "popular", boiler_frac: float = 0.0, boiler_share: float = 0.9, n_templates: int = 1000):
        assert lang in ("en", "zh", "mixed") and clusters in ("popular", "zipf50")
        self.n_docs, self.mean_bytes, self.seed, self.lang = n_docs, mean_bytes, seed, lang
        self.dup_frac, self.edit_rate, self.short_frac, self.sigma = dup_frac, edit_rate, short_frac, sigma
        self.edit_rates, self.indel_rate, self.clusters = edit_rates, indel_rate, clusters
        self.boiler_frac, self.boiler_share, self.n_templates = boiler_frac, boiler_share, n_templates
        rng = np.random.Generator(np.random.PCG64(seed))
        n = n_docs
        # language per document
        if lang == "mixed":
            self.is_zh = rng.random(n) < 0.5
        else:
            self.is_zh = np.full(n, lang == "zh")
        # bytes per token: EN words 2..10 chars + space ~ 7 ; ZH 3
        tok_bytes = np.where(self.is_zh, 3.0, 7.0)
        mu = math.log(mean_bytes) - 0.5 * sigma * sigma
        nbytes = rng.lognormal(mu, sigma, n)
        ntok = np.maximum(1, np.round(nbytes / tok_bytes)).astype(np.int64)
        short = rng.random(n) < short_frac
        ntok[short] = rng.integers(0, 3, short.sum())           # 0..2 tokens: below 13 code points
        # duplicate structure: source document of every document
        src = np.arange(n, dtype=np.int64)
        is_dup = rng.random(n) < dup_frac
        is_dup[0] = False
        idx = np.nonzero(is_dup)[0]
        rng2 = np.random.Generator(np.random.PCG64(seed + 1000))      # draws of the optional features: a stream of their own
        if idx.size and clusters == "zipf50":
            # C4: clusters of 2..50 documents (1..49 near-copies of a random original), sizes ~ Zipf(s=1)
            orig = np.nonzero(~is_dup)[0]
            w = 1.0 / np.arange(1, 50)
            sizes = rng2.choice(np.arange(1, 50), size=idx.size, p=w / w.sum())       # more than enough clusters
            n_cl = int(np.searchsorted(np.cumsum(sizes), idx.size)) + 1
            owner = np.repeat(rng2.choice(orig, n_cl, replace=n_cl > orig.size), sizes[:n_cl])[: idx.size]
            src[rng2.permutation(idx)] = owner
        elif idx.size:
            # pick an earlier document; squaring the uniform favours few popular sources -> clusters
            u = rng.random(idx.size) ** 2
            s = np.minimum((u * idx).astype(np.int64), idx - 1)
            src[idx] = s
            while True:                     # pointer jumping: the source of a copy must be an original
                nxt = src[src]
                if np.array_equal(nxt, src):
                    break
                src = nxt
            is_dup = src != np.arange(n)
        ntok = ntok[src]                    # near-copies keep the token count of their source
        self.is_zh = self.is_zh[src]
        self.src, self.is_dup, self.ntok = src, is_dup, ntok
        # per-document substitution rate (as a 32-bit threshold) and boilerplate template (-1: none)
        rates = np.asarray(edit_rates if edit_rates else (edit_rate,), dtype=np.float64)
        self.rate_table = np.minimum(rates * 2.0 ** 32, 2.0 ** 32 - 1).astype(np.int64)
        self.rate_idx = rng2.integers(0, len(rates), n, dtype=np.uint8) if edit_rates else None    # None: one rate for all
        self.template = None
        if boiler_frac > 0:
            tpl = np.full(n, -1, dtype=np.int32)
            pick = rng2.random(n) < boiler_frac
            wt = 1.0 / np.arange(1, n_templates + 1)
            tpl[pick] = rng2.choice(n_templates, int(pick.sum()), p=wt / wt.sum())
            self.template = tpl[src]                    # a near-copy keeps its source's template
        self.thr_en = _zipf_thresholds(VOCAB_EN)
        self.thr_zh = _zipf_thresholds(VOCAB_ZH)


def _tokens(spec: CorpusSpec, d0: int, d1: int, device) -> Tuple[torch.Tensor, ...]:
    """token ids / doc / pos for documents [d0, d1)."""
    ntok = torch.from_numpy(spec.ntok[d0:d1]).to(device)
    nd = d1 - d0
    doc_local = torch.repeat_interleave(torch.arange(nd, device=device), ntok)
    first = torch.cumsum(ntok, 0) - ntok
    pos = torch.arange(doc_local.numel(), device=device) - first[doc_local]
    doc = doc_local + d0
    src = torch.from_numpy(spec.src[d0:d1]).to(device)[doc_local]
    is_dup = torch.from_numpy(spec.is_dup[d0:d1]).to(device)[doc_local]
    is_zh = torch.from_numpy(spec.is_zh[d0:d1]).to(device)[doc_local]
    spos = pos
    inserted = torch.zeros_like(is_dup)
    if spec.indel_rate > 0:
        # token deletions + insertions with the length preserved: pair j of a near-copy deletes the source token at
        # q_j and inserts an own token at q2_j >= q_j; in between, the copy reads the source one position ahead
        nt = ntok[doc_local]
        npair = torch.clamp(torch.round(nt.to(torch.float64) * spec.indel_rate).to(torch.int64), max=8)
        shift = torch.zeros_like(pos)
        for j in range(8):
            qa = _lsr(_h(spec.seed + 11 + j, doc, 0), 33) % torch.clamp(nt, min=1)
            qb = _lsr(_h(spec.seed + 31 + j, doc, 0), 33) % torch.clamp(nt, min=1)
            q, q2 = torch.minimum(qa, qb), torch.maximum(qa, qb)
            on = is_dup & (npair > j)
            if spec.rate_idx is not None and (spec.rate_table == 0).any():
                # a near-copy that drew the rate 0 is an EXACT copy: no insertions / deletions either
                zero = torch.from_numpy(spec.rate_table == 0).to(device)[torch.from_numpy(spec.rate_idx[d0:d1].astype(np.int64)).to(device)]
                on = on & ~zero[doc_local]
            shift = shift + (on & (pos >= q) & (pos < q2)).to(torch.int64)
            inserted = inserted | (on & (pos == q2) & (q2 > q))
        spos = torch.minimum(pos + shift, nt - 1)
    u_src = _lsr(_h(spec.seed, src, spos), 32)
    u_own = _lsr(_h(spec.seed + 1, doc, pos), 32)
    if spec.rate_idx is None:
        thr = int(spec.rate_table[0])
    else:
        thr = torch.from_numpy(spec.rate_table).to(device)[torch.from_numpy(spec.rate_idx[d0:d1].astype(np.int64)).to(device)][doc_local]
    edit = is_dup & ((_lsr(_h(spec.seed + 2, doc, pos), 32) < thr) | inserted)
    u = torch.where(edit, u_own, u_src)
    if spec.boiler_frac > 0:
        # boilerplate: the first boiler_share of the document's tokens come from its template's stream
        tpl = torch.from_numpy(spec.template[d0:d1].astype(np.int64)).to(device)[doc_local]
        nb = (ntok[doc_local].to(torch.float64) * spec.boiler_share).to(torch.int64)
        in_tpl = (tpl >= 0) & (pos < nb) & ~edit
        u = torch.where(in_tpl, _lsr(_h(spec.seed + 7, tpl, pos), 32), u)
    thr_en = torch.from_numpy(spec.thr_en).to(device)
    thr_zh = torch.from_numpy(spec.thr_zh).to(device)
    tok = torch.where(is_zh, torch.searchsorted(thr_zh, u).clamp_(max=VOCAB_ZH - 1),
                      torch.searchsorted(thr_en, u).clamp_(max=VOCAB_EN - 1))
    return tok, doc_local, pos, is_zh, ntok


def generate(spec: CorpusSpec, d0: int = 0, d1: Optional[int] = None, device="cpu"):
    """Packed utf-8 text (uint8, padded with MDD_TEXT_PAD = 128 zero bytes) and int64
    offsets[n+1] for documents [d0, d1) of the corpus."""
    d1 = spec.n_docs if d1 is None else d1
    device = torch.device(device)
    tok, doc_local, pos, is_zh, ntok = _tokens(spec, d0, d1, device)
    nd = d1 - d0
    # token byte length (without separator): EN 2..10 from the word id, ZH 3
    wl = 2 + (_lsr(mix64(tok * _GOLD + 12345), 33) % 9)
    tlen = torch.where(is_zh, torch.full_like(wl, 3), wl)
    last = pos == (ntok[doc_local] - 1)
    sep = (~is_zh) & (~last)                                 # EN tokens are followed by one space except the last
    tbytes = tlen + sep.to(torch.int64)
    tstart = torch.cumsum(tbytes, 0) - tbytes
    total = int(tbytes.sum().item()) if tbytes.numel() else 0
    doc_bytes = torch.zeros(nd, dtype=torch.int64, device=device)
    doc_bytes.index_add_(0, doc_local, tbytes)
    offsets = torch.zeros(nd + 1, dtype=torch.int64, device=device)
    offsets[1:] = torch.cumsum(doc_bytes, 0)
    cap = ((total + 3) // 4) * 4 + 128           # MDD_TEXT_PAD
    text = torch.zeros(cap, dtype=torch.uint8, device=device)
    if total:
        tid = torch.repeat_interleave(torch.arange(tok.numel(), device=device), tbytes)
        k = torch.arange(total, device=device) - tstart[tid]
        t = tok[tid]
        zh = is_zh[tid]
        en_char = 97 + (_lsr(mix64(t * 1315423911 + k * _C1 + 77), 35) % 26)
        en_byte = torch.where(k < tlen[tid], en_char, torch.full_like(en_char, 32))
        cp = 0x4E00 + t                                      # 3-byte utf-8: 1110xxxx 10xxxxxx 10xxxxxx
        zh_byte = torch.where(k == 0, 0xE0 | (cp >> 12), torch.where(k == 1, 0x80 | ((cp >> 6) & 0x3F), 0x80 | (cp & 0x3F)))
        text[:total] = torch.where(zh, zh_byte, en_byte).to(torch.uint8)
    return text, offsets


def generate_chunked(spec: CorpusSpec, device, docs_per_chunk: int = 32768, first: int = 0, last: Optional[int] = None):
    """Documents [first, last) of the corpus (default: all of it) on `device`, generated chunk-wise to bound temporaries
    (generate() holds ~100 bytes of temporaries per text byte)."""
    last = spec.n_docs if last is None else min(last, spec.n_docs)
    texts, offs, base = [], [torch.zeros(1, dtype=torch.int64, device=device)], 0
    for d0 in range(first, last, docs_per_chunk):
        d1 = min(last, d0 + docs_per_chunk)
        t, o = generate(spec, d0, d1, device)
        nb = int(o[-1].item())
        texts.append(t[:nb])
        offs.append(o[1:] + base)
        base += nb
    total = base
    cap = ((total + 3) // 4) * 4 + 128
    text = torch.zeros(cap, dtype=torch.uint8, device=device)
    if texts:
        torch.cat(texts, out=text[:total])
    return text, torch.cat(offs)
