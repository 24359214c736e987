"""GPU suite: error behaviour of the C ABI -- bad arguments come back as MDD_ERR_INVALID with a message
(never a crash), and malformed device-side offsets are clamped instead of read out of bounds."""
import ctypes as C

import numpy as np
import pytest
import torch

from map_neo_amd import lib as mlib
from map_neo_amd.minhash import TEXT_PAD, as_u32, pack_texts

pytestmark = pytest.mark.gpu


def _p(t):
    return C.c_void_p(t.data_ptr())


def _fails(L, rc, needle):
    assert rc != 0
    msg = L.mdd_last_error().decode()
    assert needle in msg, msg


def test_invalid_arguments_are_reported(engine):
    L, ctx = engine.L, engine.ctx
    buf, offs = pack_texts(["hello world, this is a document"] * 3)
    text, o = engine.to_device(buf), engine.to_device(offs)
    a, b = engine.perms(1, 128)
    sig = torch.empty((3, 128), dtype=torch.int32, device=engine.device)
    tb = int(offs[-1])
    ok = lambda K=128, width=13, unit=0, t=text: L.mdd_signatures(ctx, _p(t), tb, _p(o), 3, unit, width, K, _p(a), _p(b), _p(sig))
    assert ok() == 0
    _fails(L, ok(K=0), "K=0")
    _fails(L, ok(K=257), "K=257")
    _fails(L, ok(width=0), "width=0")
    _fails(L, ok(width=129), "width=129")
    _fails(L, ok(unit=7), "unknown unit")
    _fails(L, L.mdd_signatures(ctx, C.c_void_p(text.data_ptr() + 1), tb - 1, _p(o), 3, 0, 13, 128, _p(a), _p(b), _p(sig)), "aligned")
    _fails(L, L.mdd_signatures(None, _p(text), tb, _p(o), 3, 0, 13, 128, _p(a), _p(b), _p(sig)), "ctx is NULL")
    _fails(L, L.mdd_signatures(ctx, _p(text), tb, None, 3, 0, 13, 128, _p(a), _p(b), _p(sig)), "NULL")
    key = torch.empty(27, dtype=torch.int64, device=engine.device)
    meta = torch.empty(27, dtype=torch.int64, device=engine.device)
    _fails(L, L.mdd_band_keys(ctx, _p(sig), 3, 128, 10, 13, 0, 64, _p(key), _p(meta)), "bands")
    _fails(L, L.mdd_band_keys(ctx, _p(sig), 3, 128, 9, 13, 0, 0, _p(key), _p(meta)), "key_bits")
    _fails(L, L.mdd_band_keys(ctx, _p(sig), 3, 128, 9, 13, 2 ** 32 - 2, 64, _p(key), _p(meta)), "2^32")
    assert L.mdd_band_keys(ctx, _p(sig), 3, 128, 9, 13, 0, 64, _p(key), _p(meta)) == 0
    tab = torch.empty(48, dtype=torch.int64, device=engine.device)            # not a power of two
    eu = torch.empty(27, dtype=torch.int32, device=engine.device)
    cnt = torch.zeros(2, dtype=torch.int64, device=engine.device)
    coll = torch.empty(64, dtype=torch.int64, device=engine.device)
    be = lambda size: L.mdd_bucket_edges(ctx, _p(key), _p(meta), 27, _p(sig), 9, 128, 13, 13, _p(tab), size, _p(eu), _p(eu), 27,
                                         _p(cnt), C.c_void_p(cnt.data_ptr() + 8), None)
    _fails(L, be(48), "power of two")
    _fails(L, be(32), "power of two")                                          # < 2 * n_rec
    _fails(L, L.mdd_ctx_set_segment_bytes(ctx, 100), "multiple of 256")
    _fails(L, L.mdd_exact_group(ctx, _p(text), tb, _p(o), _p(o), 3, 10, 64, _p(key), _p(tab), 4, _p(eu), _p(cnt)), "power of two")
    with pytest.raises(mlib.MddError):
        engine.signatures(text[: tb + 8], o, text_bytes=tb)                   # missing MDD_TEXT_PAD
    engine.sync()


def test_malformed_offsets_do_not_leave_the_allocation(engine):
    """offsets are device data the host cannot check: the kernels clamp them to [0, text_bytes]"""
    docs = ["some perfectly normal document number %d with enough characters" % i for i in range(50)]
    buf, offs = pack_texts(docs)
    good = as_u32(engine.signatures(engine.to_device(buf), engine.to_device(offs)))
    bad = offs.copy()
    bad[10] = 2 ** 40                  # far outside
    bad[20] = bad[19] - 5 if bad[19] > 5 else 0   # descending
    bad[-1] = 2 ** 63
    text = engine.to_device(buf)
    sig = as_u32(engine.signatures(text, engine.to_device(bad), text_bytes=int(offs[-1])))
    engine.sync()
    assert sig.shape == good.shape
    untouched = [d for d in range(50) if d not in (9, 10, 18, 19, 20, 49)]
    assert (sig[untouched] == good[untouched]).all()
    # exact grouping and normalisation with ranges past the text
    ub = engine.to_device(np.array([0, 5, 2 ** 40, 30], dtype=np.uint64))
    ue = engine.to_device(np.array([20, 2 ** 41, 2 ** 40 + 5, 10], dtype=np.uint64))
    first = as_u32(engine.exact_group(text, ub, ue))
    assert first.tolist()[0] == 0 and len(first) == 4
    t, o, f = engine.normalize_en(text, engine.to_device(bad))
    engine.sync()
    assert o.numel() == len(bad)
