"""The same raw C-ABI call sequence on libmdd_hip.so (device buffers) and on oracle/libmdd_cpu.so (host buffers) must give the
same signatures, (key, meta) records, bucket heads, component labels, removed mask and per-edge signature matches."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from map_neo_amd import corpus
from tests import twin

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lang,b,r,K", [("en", 9, 13, 128), ("mixed", 20, 12, 256)])
def test_hip_library_equals_cpu_twin_through_the_c_abi(engine, lang, b, r, K):
    subprocess.check_call(["make", "-s", "-C", os.path.join(twin.ROOT, "oracle"), "libmdd_cpu.so"])
    cpu = twin.load_cpu()
    cctx = C.c_void_p()
    assert cpu.mdd_ctx_create(0, None, C.byref(cctx)) == 0
    spec = corpus.CorpusSpec(4000, 400, lang=lang, seed=31, dup_frac=0.25, short_frac=0.01)
    text, offs = corpus.generate(spec)
    text, offs = text.numpy(), offs.numpy().astype(np.uint64)
    ref = twin.run_path(cpu, cctx, twin.HostBuffers(), text, offs, K=K, bands=b, rows=r)
    got = twin.run_path(engine.L, engine.ctx, twin.DeviceBuffers(engine.device), text, offs, K=K, bands=b, rows=r)
    assert cpu.mdd_ctx_destroy(cctx) == 0
    for name in ("sig", "key", "meta", "labels", "removed", "rec_bucket"):
        assert np.array_equal(got[name], ref[name]), name
    assert got["n_removed"] == ref["n_removed"] > 0
    # which lower record an edge names depends on scheduling (include/mdd.h); the documents that emit one do not
    assert sorted(u for u, _, _ in got["edges"]) == sorted(u for u, _, _ in ref["edges"])
    assert all(ref["labels"][u] == ref["labels"][v] for u, v, _ in got["edges"])
