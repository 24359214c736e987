"""TEST-ONLY backend with the stage interface of map_neo_amd.minhash.Engine, computed by the
CPU oracle on torch CPU tensors.  It exists so that the sharding / all-to-all / all-gather
logic of map_neo_amd.dedup.dedup_distributed can run under gloo without a GPU.  Never
imported by the product package."""
import hashlib

import numpy as np
import torch

from oracle import ref_c as RC
from oracle import ref_py as R


def _key_of(band: int, tup: np.ndarray) -> int:
    h = hashlib.blake2b(bytes([band]) + tup.tobytes(), digest_size=8).digest()
    return int.from_bytes(h, "little") or 1


class OracleBackend:
    device = torch.device("cpu")

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        view = {np.dtype(np.uint8): np.uint8, np.dtype(np.uint32): np.int32, np.dtype(np.uint64): np.int64,
                np.dtype(np.int32): np.int32, np.dtype(np.int64): np.int64}[arr.dtype]
        return torch.from_numpy(arr.view(view).copy())

    def signatures(self, text, offsets, num_perm=128, width=13, unit="char", seed=1, perms=None, text_bytes=None,
                   out=None):
        sig = RC.signatures(text.numpy(), offsets.numpy().astype(np.uint64), num_perm, width, unit, seed)
        return self.to_device(sig)

    def band_keys(self, sig, bands=9, rows=13, doc_base=0, key_bits=64):
        s = sig.numpy().view(np.uint32)
        n = s.shape[0]
        key = np.zeros(n * bands, np.uint64)
        meta = np.zeros(n * bands, np.uint64)
        for d in range(n):
            for j in range(bands):
                key[d * bands + j] = _key_of(j, s[d, j * rows:(j + 1) * rows])
                meta[d * bands + j] = (j << 56) | (doc_base + d)
        return self.to_device(key), self.to_device(meta)

    def packed_keys(self, tuples, meta, rows, key_bits=64):
        t = tuples.numpy().view(np.uint32).reshape(-1, rows)
        m = meta.numpy().view(np.uint64)
        return self.to_device(np.array([_key_of(int(m[i] >> np.uint64(56)), t[i]) for i in range(len(m))], np.uint64))

    def route(self, sig, key, meta, bands, rows, world):
        s = sig.numpy().view(np.uint32)
        k = key.numpy().view(np.uint64)
        dest = ((k >> np.uint64(32)) % np.uint64(world)).astype(np.int64)
        order = np.argsort(dest, kind="stable")
        counts = np.bincount(dest, minlength=world).astype(np.int64)
        tup = np.stack([s[i // bands, (i % bands) * rows:(i % bands + 1) * rows] for i in order]) if len(order) else \
            np.zeros((0, rows), np.uint32)
        return self.to_device(np.ascontiguousarray(tup.reshape(-1))), meta[torch.from_numpy(order)], torch.from_numpy(counts)

    def bucket_edges(self, key, meta, tuple_base, rec_per_doc, doc_stride, band_stride, rows, want_buckets=False):
        m = meta.numpy().view(np.uint64)
        t = tuple_base.numpy().view(np.uint32)
        first = {}
        eu, ev = [], []
        for i in range(len(m)):
            off = (i // rec_per_doc) * doc_stride + (i % rec_per_doc) * band_stride
            k = (int(m[i]) >> 56, t[off:off + rows].tobytes())
            d = int(m[i]) & 0xFFFFFFFF
            if k in first:
                if first[k] != d:
                    eu.append(d)
                    ev.append(first[k])
            else:
                first[k] = d
        return (self.to_device(np.asarray(eu, np.uint32)), self.to_device(np.asarray(ev, np.uint32)),
                0, None)

    def components_init(self, n):
        return torch.arange(n, dtype=torch.int32)

    def components_union(self, labels, eu, ev):
        self._edges = getattr(self, "_edges", []) + list(zip(eu.numpy().view(np.uint32).tolist(),
                                                           ev.numpy().view(np.uint32).tolist()))

    def components_finish(self, labels):
        lab = R.components_min_label(labels.numel(), self._edges)
        labels.copy_(torch.from_numpy(lab.view(np.int32)))
        self._edges = []

    def signature_matches(self, sig, eu, ev):
        m = R.signature_matches(sig.numpy().view(np.uint32), eu.numpy().view(np.uint32), ev.numpy().view(np.uint32))
        return torch.from_numpy(m.astype(np.int32))

    def forest_edges(self, labels, max_edges):
        lab = labels.numpy().view(np.uint32)
        idx = np.flatnonzero(lab != np.arange(lab.shape[0], dtype=np.uint32)).astype(np.uint32)
        return self.to_device(idx), self.to_device(lab[idx])

    def removed_mask(self, labels, first, n):
        lab = labels.numpy().view(np.uint32)
        rm = lab[first:first + n] != np.arange(first, first + n, dtype=np.uint32)
        return torch.from_numpy(rm.astype(np.uint8)), torch.tensor([int(rm.sum())], dtype=torch.int64)
