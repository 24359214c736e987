"""oracle/libmdd_cpu.so: the CPU oracle behind the product's C ABI ("same symbols built twice").  The hot path is
driven through raw ctypes calls and pinned on the committed golden vectors (tests/golden): signatures, bucket minima,
component labels, removed mask; plus the error contract.  tests/test_gpu_twin.py makes the same calls on libmdd_hip.so."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tests import twin


@pytest.fixture(scope="module")
def cpu_lib():
    subprocess.check_call(["make", "-s", "-C", os.path.join(twin.ROOT, "oracle"), "libmdd_cpu.so"])
    return twin.load_cpu()


@pytest.fixture()
def cpu_ctx(cpu_lib):
    ctx = C.c_void_p()
    assert cpu_lib.mdd_ctx_create(0, None, C.byref(ctx)) == 0
    yield ctx
    assert cpu_lib.mdd_ctx_destroy(ctx) == 0


def _pack(docs):
    enc = [d.encode("utf8") for d in docs]
    offs = np.zeros(len(enc) + 1, np.uint64)
    np.cumsum([len(e) for e in enc], out=offs[1:])
    return np.frombuffer(b"".join(enc), dtype=np.uint8), offs


def test_twin_exports_the_hot_path_subset_with_the_headers_signatures(cpu_lib):
    assert cpu_lib.mdd_version() == 100
    for name in twin.TWIN_SYMBOLS:
        assert hasattr(cpu_lib, name)


def test_twin_perms_golden(cpu_lib, golden_perms):
    a, b = np.zeros(256, np.uint64), np.zeros(256, np.uint64)
    assert cpu_lib.mdd_make_perms(1, 256, a.ctypes.data, b.ctypes.data) == 0
    assert np.array_equal(np.stack([a, b]), golden_perms)


@pytest.mark.parametrize("b,r,K", [(9, 13, 128), (14, 9, 128), (20, 12, 256)])
def test_twin_path_golden(cpu_lib, cpu_ctx, mini_docs, mini_expected, b, r, K):
    text, offs = _pack(mini_docs)
    n = len(mini_docs)
    out = twin.run_path(cpu_lib, cpu_ctx, twin.HostBuffers(), text, offs, K=K, bands=b, rows=r)
    assert np.array_equal(out["sig"], mini_expected["sig_char13_K256"][:, :K])
    assert np.array_equal(out["labels"], mini_expected[f"labels_b{b}_r{r}_K{K}"])
    assert np.array_equal(out["removed"].astype(bool), out["labels"] != np.arange(n))
    assert out["n_removed"] == int(out["removed"].sum())
    bmin = (out["rec_bucket"].reshape(n, b) // b).T
    assert np.array_equal(bmin, mini_expected[f"bucketmin_b{b}_r{r}_K{K}"])
    assert all(m <= K for _, _, m in out["edges"]) and all(m >= r for _, _, m in out["edges"])   # a shared band = r equal components


def test_twin_other_units_and_widths_golden(cpu_lib, cpu_ctx, mini_docs, mini_expected):
    text, offs = _pack(mini_docs)
    for name, unit, width, K in (("sig_word5_K128", 1, 5, 128), ("sig_char5_K128", 0, 5, 128), ("sig_char20_K64", 0, 20, 64)):
        out = twin.run_path(cpu_lib, cpu_ctx, twin.HostBuffers(), text, offs, K=K, width=width, unit=unit, bands=4, rows=8)
        assert np.array_equal(out["sig"], mini_expected[name]), name


def test_twin_error_contract(cpu_lib, cpu_ctx):
    a = np.zeros(8, np.uint64)
    assert cpu_lib.mdd_make_perms(1, 1000, a.ctypes.data, a.ctypes.data) == -1
    assert b"K=1000" in cpu_lib.mdd_last_error()
    sig = np.zeros(128, np.uint32)
    k = np.zeros(16, np.uint64)
    assert cpu_lib.mdd_band_keys(cpu_ctx, sig.ctypes.data, 1, 128, 20, 13, 0, 64, k.ctypes.data, k.ctypes.data) == -1
    assert b"do not fit" in cpu_lib.mdd_last_error()
    assert cpu_lib.mdd_signatures(None, None, 0, None, 0, 0, 13, 128, None, None, None) == -1
    # empty batch: nothing to do, no error
    assert cpu_lib.mdd_signatures(cpu_ctx, None, 0, None, 0, 0, 13, 128, None, None, None) == 0
