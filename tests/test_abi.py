"""CPU suite: the C-ABI library loads and exports every symbol include/mdd.h declares; the
host-only entry point (mdd_make_perms) matches the golden table; compute calls are NOT made
here (no GPU).  Also checks the product package never routes through the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from map_neo_amd import lib as mlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "mdd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mdd_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = mlib.load()
    syms = _header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(L, s), f"libmdd_hip.so lacks {s}"
    assert set(syms) == set(mlib.SIGNATURES), "ctypes table and header disagree"
    assert L.mdd_version() >= 100


def test_io_library_exports_every_declared_symbol():
    """include/mdd_io.h (native ingest / egress) against libmdd_io.so"""
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "mdd_io.h")).read(), flags=re.S)
    syms = sorted(set(re.findall(r"\b(mdd_io_[a-z0-9_]+)\s*\(", src)))
    assert len(syms) >= 17 and "mdd_io_write_jsonl" in syms
    L = C.CDLL(os.path.join(ROOT, "map_neo_amd", "libmdd_io.so"))
    for s in syms:
        assert hasattr(L, s), f"libmdd_io.so lacks {s}"


def test_make_perms_matches_golden(golden_perms):
    from map_neo_amd.minhash import make_perms
    a, b = make_perms(1, 256)
    assert (np.stack([a, b]) == golden_perms).all()
    a128, _ = make_perms(1, 128)
    assert (a128 == golden_perms[0, :128]).all()
    L = mlib.load()
    assert L.mdd_make_perms(1, 257, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)) == -1
    assert b"outside" in L.mdd_last_error()


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from map_neo_amd.minhash import Engine
    with pytest.raises(mlib.MddError):
        Engine(0)
    h = C.c_void_p()
    rc = mlib.load().mdd_ctx_create(0, None, C.byref(h))
    assert rc != 0 and b"no CPU fallback" in mlib.load().mdd_last_error()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "map_neo_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f), encoding="utf8").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "ref_py" not in txt and "libmdd_ref" not in txt, f


def test_integration_doc_names_every_entry_point():
    """INTEGRATION.md's appendix maps every exported entry point to what it replaces in the reference"""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [s for s in _header_symbols() if s not in doc]
    assert not missing, missing
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "mdd_io.h")).read(), flags=re.S)
    for s in sorted(set(re.findall(r"\b(mdd_io_[a-z0-9_]+)\s*\(", src))):
        assert s in doc or ("`" + s[len("mdd_io"):] + "`") in doc, s
