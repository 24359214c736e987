"""CPU ORACLE (test infrastructure, NOT product code) -- numpy/hashlib restatement of
MAP-NEO Matrix MinHash-LSH near-dedup.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module.  The product path (`map_neo_amd/`) never imports it and has no CPU fallback.

Reference files restated (paths relative to /root/reference):
  EN   = Matrix/english_data_clean_pipeline/dedup/Minhash_Lsh_Deduplication/minhashLsh.py
  ZH-1 = Matrix/chinese_data_clean_pipeline/deduplication/minhash_lsh/generate_minhash.py
  ZH-2 = .../minhash_lsh/generate_dup_pairs.py
  ZH-3 = .../minhash_lsh/generate_connected_components.py
  ZH-4 = .../minhash_lsh/generate_dup_line_id_for_each_file.py

The per-shingle arithmetic lives in the third-party `datasketch` package (not vendored, not
installable here; the reference pins no version).  We restate the published algorithm of
datasketch.MinHash (1.5/1.6: seed=1, hashfunc=sha1_hash32, _mersenne_prime=2^61-1,
_max_hash=2^32-1) and nltk.ngrams (plain unpadded sliding window).

PARITY PIN STATUS: the reference ships no tests, fixtures or golden vectors for this path
 and its dependencies cannot be imported here, so this oracle is pinned
only by (a) SHA-1 FIPS vectors, (b) numpy's own RandomState stream, (c) the known-answer
vectors recorded in SURVEY.md section 8(c), which were derived from this same restatement.
=> "parity unpinned" with respect to a running datasketch; see DESIGN.md.
"""
from __future__ import annotations

import hashlib
import re
import string
import struct
from typing import Iterable, List, Sequence, Tuple

import numpy as np

MERSENNE_PRIME = np.uint64((1 << 61) - 1)   # datasketch _mersenne_prime
MAX_HASH = np.uint64((1 << 32) - 1)         # datasketch _max_hash
HASH_RANGE = 1 << 32

# Reference constants: EN:17-20, ZH-1:14-17
WIDTH = 13
BANDS = 9
ROWS = 13
NUM_PERM = 128
SEED = 1


# --------------------------------------------------------------------------- a4
def make_perms(seed: int = SEED, num_perm: int = NUM_PERM) -> Tuple[np.ndarray, np.ndarray]:
    """datasketch MinHash._init_permutations: interleaved (a_i, b_i) draws from the legacy
    numpy RandomState(seed); a in [1, p-1), b in [0, p-1).  Called per document by the
    reference through `MinHash(num_perm=128)` (EN:62, ZH-1:28)."""
    gen = np.random.RandomState(seed)
    p = int(MERSENNE_PRIME)
    ab = np.array(
        [(gen.randint(1, p, dtype=np.uint64), gen.randint(0, p, dtype=np.uint64))
         for _ in range(num_perm)], dtype=np.uint64).T
    return ab[0].copy(), ab[1].copy()


# --------------------------------------------------------------------------- a3
def sha1_hash32(data: bytes) -> int:
    """datasketch.hashfunc.sha1_hash32: little-endian u32 of the first 4 digest bytes."""
    return struct.unpack('<I', hashlib.sha1(data).digest()[:4])[0]


# --------------------------------------------------------------------------- a2
def shingles_char(text: str, width: int = WIDTH) -> List[bytes]:
    """EN:61,63 / ZH-1:27,29: nltk.ngrams over a str slides over code points, no padding;
    each window is joined and utf-8 encoded.  len(text) < width => no shingles."""
    n = len(text) - width + 1
    return [text[i:i + width].encode('utf8') for i in range(max(0, n))]


def shingles_word(text: str, n: int) -> List[bytes]:
    """Parameter extension (BASELINE.json config 1, 'n-gram words'; not in the reference):
    nltk.ngrams(tokens, n) over the space-separated tokens of the already-normalised text,
    joined with one space, utf-8 encoded.  Tokens are maximal runs of non-space characters
    (0x20 is the only separator) and the shingle is the text from the first token's start to
    the last token's end -- on normalised text (single spaces, EN:31-40) exactly
    ' '.join(tokens[i:i+n]); on text with runs of spaces the run stays inside the shingle
    (what the C restatement and the device kernel hash: no copy of the tokens is made)."""
    spans, i, L = [], 0, len(text)
    while i < L:
        if text[i] == ' ':
            i += 1
            continue
        j = i
        while j < L and text[j] != ' ':
            j += 1
        spans.append((i, j))
        i = j
    m = len(spans) - n + 1
    return [text[spans[k][0]:spans[k + n - 1][1]].encode('utf8') for k in range(max(0, m))]


def shingles(text: str, unit: str = 'char', width: int = WIDTH) -> List[bytes]:
    if unit == 'char':
        return shingles_char(text, width)
    if unit == 'word':
        return shingles_word(text, width)
    raise ValueError(f'unknown unit {unit!r}')


# --------------------------------------------------------------------------- a5
class MinHash:
    """Restatement of datasketch.MinHash as used by EN:62-63 / ZH-1:28-29: the constructor
    regenerates the permutation table (as the reference does per document), `update` does
    phv = ((a*hv + b) mod 2^64) % (2^61-1) & 0xFFFFFFFF ; hashvalues = min(hashvalues, phv)
    with numpy uint64 wrap-around multiply."""

    def __init__(self, num_perm: int = NUM_PERM, seed: int = SEED, perms=None):
        self.seed = seed
        self.hashvalues = np.ones(num_perm, dtype=np.uint64) * MAX_HASH
        if perms is None:
            perms = make_perms(seed, num_perm)
        self.permutations = perms

    def update(self, b: bytes) -> None:
        hv = np.uint64(sha1_hash32(b))
        a, bb = self.permutations
        with np.errstate(over='ignore'):
            phv = np.bitwise_and((a * hv + bb) % MERSENNE_PRIME, MAX_HASH)
        self.hashvalues = np.minimum(phv, self.hashvalues)


def permute_min(hashes: np.ndarray, a: np.ndarray, b: np.ndarray, chunk: int = 4096) -> np.ndarray:
    """Vectorised equivalent of repeated MinHash.update over a batch of u32 shingle hashes
    (same numpy uint64 wrap-around semantics; used so tests finish in seconds)."""
    K = a.shape[0]
    sig = np.full(K, MAX_HASH, dtype=np.uint64)
    hashes = np.asarray(hashes, dtype=np.uint64)
    for s in range(0, hashes.shape[0], chunk):
        h = hashes[s:s + chunk, None]
        with np.errstate(over='ignore'):
            phv = np.bitwise_and((a[None, :] * h + b[None, :]) % MERSENNE_PRIME, MAX_HASH)
        sig = np.minimum(sig, phv.min(axis=0))
    return sig


def signature(text: str, num_perm: int = NUM_PERM, width: int = WIDTH, unit: str = 'char',
              seed: int = SEED, perms=None, literal: bool = False) -> np.ndarray:
    """generate_minhash_value (EN:56-64) / first half of generate_hash_values (ZH-1:23-29)
    for one already-normalised text -> uint64[num_perm] `hashvalues`.
    literal=True runs the per-shingle MinHash.update loop exactly as the reference does."""
    if perms is None:
        perms = make_perms(seed, num_perm)
    feats = shingles(text, unit, width)
    if literal:
        m = MinHash(num_perm=num_perm, seed=seed, perms=perms)
        for f in feats:
            m.update(f)
        return m.hashvalues
    if not feats:
        return np.full(num_perm, MAX_HASH, dtype=np.uint64)
    hs = np.fromiter((sha1_hash32(f) for f in feats), dtype=np.uint64, count=len(feats))
    return permute_min(hs, perms[0], perms[1])


def signatures(texts: Sequence[str], num_perm: int = NUM_PERM, width: int = WIDTH,
               unit: str = 'char', seed: int = SEED) -> np.ndarray:
    """Batch of `signature` -> uint32[n, num_perm] (values always < 2^32)."""
    perms = make_perms(seed, num_perm)
    out = np.empty((len(texts), num_perm), dtype=np.uint32)
    for i, t in enumerate(texts):
        out[i] = signature(t, num_perm, width, unit, seed, perms).astype(np.uint32)
    return out


# --------------------------------------------------------------------------- a1
_PUNCT_TABLE = str.maketrans("", "", string.punctuation)


def normalize_text(text: str, fix_text=None) -> str:
    """EN:31-40.  ftfy.fix_text(normalization="NFC") is third-party and absent here; pass a
    callable as `fix_text` to apply it, default = unicodedata NFC only (identity on the
    synthetic corpora, which are generated already normalised)."""
    if fix_text is not None:
        text = fix_text(text)
    else:
        import unicodedata
        text = unicodedata.normalize('NFC', text)
    text = text.lower()
    text = text.translate(_PUNCT_TABLE)
    text = re.sub(r"\s+", " ", text.strip())
    return text


# --------------------------------------------------------------------------- a7
def _H(hs: np.ndarray) -> bytes:
    """EN:28-29 / ZH-1:20-21: big-endian bytes of a uint64 slice."""
    return bytes(hs.byteswap().data)


def band_keys(hashvalues: np.ndarray, bands: int = BANDS, rows: int = ROWS) -> List[bytes]:
    """generate_bands (EN:66-71) / tail of generate_hash_values (ZH-1:30-33): band j key =
    8-byte big-endian serialisation of hashvalues[j*r:(j+1)*r] (104 bytes for r=13)."""
    hv = np.ascontiguousarray(hashvalues, dtype=np.uint64)
    return [_H(hv[j * rows:(j + 1) * rows]) for j in range(bands)]


# --------------------------------------------------------------------------- a9-a11
def buckets(sigs: np.ndarray, bands: int = BANDS, rows: int = ROWS) -> dict:
    """EN:136 groupBy((band, key)) / ZH-2:19-37 dict per band: exact-equality grouping.
    Returns {(band, key_bytes): [doc indices in ascending order]} (all buckets, incl. singletons)."""
    out: dict = {}
    for d in range(sigs.shape[0]):
        for j, key in enumerate(band_keys(sigs[d].astype(np.uint64), bands, rows)):
            out.setdefault((j, key), []).append(d)
    return out


def generate_edges(doc_id_list: Sequence) -> List[tuple]:
    """EN:73-80 chain edges (max, min) between consecutive members."""
    edges = []
    for i in range(len(doc_id_list) - 1):
        x, y = doc_id_list[i], doc_id_list[i + 1]
        edges.append((x, y) if x > y else (y, x))
    return edges


def star_edges(doc_id_list: Sequence) -> List[tuple]:
    """ZH-2:30-37 star edges: each later doc paired with the first-seen doc of the bucket."""
    return [(d, doc_id_list[0]) for d in doc_id_list[1:]]


# --------------------------------------------------------------------------- a12
def large_star_small_star(edges: Iterable[tuple]) -> set:
    """EN:82-115,138-144: literal alternating large-star / small-star loop with the
    reference's own stopping rule (EN:142 compares the small-star output with the large-star
    output of the SAME iteration).  That rule can stop one large-star early, so the returned
    edge set is not always the pure star {(v, min(component))}; what IS guaranteed at a
    small-star fixpoint is that every non-minimum node of a component has a smaller neighbour,
    i.e. {x[0] for x in result} == all non-minimum members == remove_list of EN:145.
    Small inputs only (pure Python)."""
    a = set(edges)
    while True:
        nb: dict = {}
        for u, v in a:                       # large_star_map EN:82-83
            nb.setdefault(u, []).append(v)
            nb.setdefault(v, []).append(u)
        b = set()
        for u, pts in nb.items():            # large_star_reduce EN:85-95
            m = min([u] + pts)
            for p in pts:
                if p > u:
                    b.add((p, m))
        nb = {}
        for u, v in b:                       # small_star_map EN:97-102
            if v <= u:
                nb.setdefault(u, []).append(v)
            else:
                nb.setdefault(v, []).append(u)
        a2 = set()
        for u, N in nb.items():              # small_star_reduce EN:104-115
            m = min([u] + N)
            if m != u:
                a2.add((u, m))
            for p in N:
                if p != m:
                    a2.add((p, m))
        changes = len(a2 - b) + len(b - a2)  # EN:142
        a = a2
        if changes == 0:
            return a


def signature_matches(sigs: np.ndarray, eu: Sequence[int], ev: Sequence[int]) -> np.ndarray:
    """Number of equal signature components of the documents of every edge = K x datasketch's
    `MinHash.jaccard(other)` (published algorithm: `np.count_nonzero(self.hashvalues == other.hashvalues) / len(self)`).
    An extension of the build: the reference scripts never call it."""
    sigs = np.asarray(sigs)
    return np.array([int(np.count_nonzero(sigs[int(u)] == sigs[int(v)])) for u, v in zip(eu, ev)], dtype=np.uint32)


def components_min_label(n: int, edges: Iterable[tuple]) -> np.ndarray:
    """Connected components with label = minimum doc index of the component (what the EN
    star loop converges to, EN:145; a legal outcome of ZH-3/ZH-4)."""
    parent = list(range(n))

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    for u, v in edges:
        ru, rv = find(u), find(v)
        if ru != rv:
            if ru < rv:
                parent[rv] = ru
            else:
                parent[ru] = rv
    return np.array([find(i) for i in range(n)], dtype=np.uint32)


# --------------------------------------------------------------------------- whole path
def dedup(texts: Sequence[str], num_perm: int = NUM_PERM, width: int = WIDTH, unit: str = 'char',
          bands: int = BANDS, rows: int = ROWS, seed: int = SEED, edge_shape: str = 'chain'):
    """Whole path on dense doc indices 0..n-1 (doc_id order == index order):
    signatures -> buckets -> edges -> components -> removed mask (EN:145: every non-minimum
    member of a component is removed).  Returns dict of all intermediate artifacts."""
    sigs = signatures(texts, num_perm, width, unit, seed)
    bk = buckets(sigs, bands, rows)
    edges = set()
    for members in bk.values():
        if len(members) > 1:
            es = generate_edges(members) if edge_shape == 'chain' else star_edges(members)
            edges.update(es)
    labels = components_min_label(len(texts), edges)
    removed = labels != np.arange(len(texts), dtype=np.uint32)
    return {'signatures': sigs, 'buckets': bk, 'edges': edges, 'labels': labels, 'removed': removed}
