"""ctypes loader for the plain-C oracle (oracle/mdd_ref.c).  TEST INFRASTRUCTURE ONLY:
importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libmdd_ref.so')


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, 'mdd_ref.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-s', '-C', _HERE, 'libmdd_ref.so'])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        u8p, u32p, u64p = C.c_void_p, C.c_void_p, C.c_void_p
        L.ref_sha1.argtypes = [u8p, C.c_uint64, u8p]
        L.ref_sha1_hash32.argtypes = [u8p, C.c_uint64]
        L.ref_sha1_hash32.restype = C.c_uint32
        L.ref_make_perms.argtypes = [C.c_uint32, C.c_int, u64p, u64p]
        L.ref_signatures.argtypes = [u8p, u64p, C.c_uint64, C.c_int, C.c_int, C.c_int, u64p, u64p, u32p]
        L.ref_signatures.restype = C.c_uint64
        L.ref_signatures_mt.argtypes = [u8p, u64p, C.c_uint64, C.c_int, C.c_int, C.c_int, u64p, u64p, u32p, C.c_int]
        L.ref_signatures_mt.restype = C.c_uint64
        L.ref_shingle_hashes.argtypes = [u8p, C.c_uint64, C.c_int, C.c_int, u32p, C.c_uint64]
        L.ref_shingle_hashes.restype = C.c_uint64
        L.ref_cluster.argtypes = [u32p, C.c_uint64, C.c_int, C.c_int, C.c_int, u32p, u64p]
        L.ref_cluster.restype = C.c_uint64
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def sha1(data: bytes) -> bytes:
    buf = np.frombuffer(data, dtype=np.uint8) if data else np.zeros(1, np.uint8)
    out = np.zeros(20, np.uint8)
    lib().ref_sha1(_p(buf), len(data), _p(out))
    return out.tobytes()


def make_perms(seed: int, K: int):
    a = np.zeros(K, np.uint64)
    b = np.zeros(K, np.uint64)
    lib().ref_make_perms(seed, K, _p(a), _p(b))
    return a, b


UNIT = {'char': 0, 'word': 1}


def signatures(text_u8: np.ndarray, offsets: np.ndarray, K: int = 128, width: int = 13,
               unit: str = 'char', seed: int = 1, perms=None, threads: int = 1) -> np.ndarray:
    text_u8 = np.ascontiguousarray(text_u8, np.uint8)
    offsets = np.ascontiguousarray(offsets, np.uint64)
    n = offsets.shape[0] - 1
    a, b = perms if perms is not None else make_perms(seed, K)
    a = np.ascontiguousarray(a, np.uint64)
    b = np.ascontiguousarray(b, np.uint64)
    sig = np.empty((n, K), np.uint32)
    tp = text_u8 if text_u8.size else np.zeros(1, np.uint8)
    if threads > 1:
        lib().ref_signatures_mt(_p(tp), _p(offsets), n, UNIT[unit], width, K, _p(a), _p(b), _p(sig), threads)
    else:
        lib().ref_signatures(_p(tp), _p(offsets), n, UNIT[unit], width, K, _p(a), _p(b), _p(sig))
    return sig


def shingle_hashes(doc: bytes, width: int = 13, unit: str = 'char') -> np.ndarray:
    buf = np.frombuffer(doc, dtype=np.uint8) if doc else np.zeros(1, np.uint8)
    cap = max(1, len(doc))
    out = np.zeros(cap, np.uint32)
    n = lib().ref_shingle_hashes(_p(buf), len(doc), UNIT[unit], width, _p(out), cap)
    return out[:n]


def cluster(sig: np.ndarray, bands: int, rows: int, want_buckets: bool = False):
    sig = np.ascontiguousarray(sig, np.uint32)
    n, K = sig.shape
    assert bands * rows <= K
    labels = np.empty(n, np.uint32)
    bucket_of = np.empty(n * bands, np.uint64) if want_buckets else None
    ne = lib().ref_cluster(_p(sig), n, K, bands, rows, _p(labels), _p(bucket_of) if want_buckets else None)
    return (labels, ne, bucket_of) if want_buckets else (labels, ne)
