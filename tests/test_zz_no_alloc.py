"""mdd_signatures_noalloc (include/mdd.h): the allocation-free variant of mdd_signatures -- segment queue and
counters in the library's file-scope `static __device__` scratch -- must give the signatures of mdd_signatures,
bit for bit, on two calls in a row (the second smaller), called through the C ABI with preallocated buffers.
Collected last (test_zz_*): a failure here must not hide the other GPU tests under `pytest -x`."""
import ctypes as C

import numpy as np
import pytest
import torch

from map_neo_amd import corpus
from map_neo_amd import lib as _lib
from map_neo_amd.minhash import TEXT_PAD, as_u32, pack_texts

pytestmark = pytest.mark.gpu


def _p(t):
    return C.c_void_p(t.data_ptr())


def _call(engine, fn, text, text_bytes, offs, K, width, unit, a, b, out):
    n = offs.numel() - 1
    rc = fn(engine.ctx, _p(text), int(text_bytes), _p(offs), n, _lib.MDD_UNIT[unit], width, K, _p(a), _p(b), _p(out))
    _lib.check(rc, "signatures")


def test_noalloc_signatures_equal_mdd_signatures_two_calls(engine):
    L = engine.L
    # call 1: mixed EN/ZH documents incl. long ones (the segment queue in the static scratch is really used:
    # 256-byte segments put ~all documents through it); call 2: a smaller corpus, different K / unit
    spec1 = corpus.CorpusSpec(6000, 900, lang="mixed", dup_frac=0.2, seed=5)
    t1, o1 = corpus.generate(spec1, device=engine.device)
    spec2 = corpus.CorpusSpec(700, 400, lang="en", dup_frac=0.1, seed=6)
    t2, o2 = corpus.generate(spec2, device=engine.device)
    a, b = engine.perms(1, 256)
    cases = [(t1, o1, 128, 13, "char"), (t2, o2, 256, 5, "word")]
    # every buffer exists before the calls under test: they allocate nothing (the library never does; torch does not either)
    outs = [torch.full((o.numel() - 1, K), 0x5A5A5A5A, dtype=torch.int32, device=engine.device) for (_, o, K, _, _) in cases]
    refs = [torch.empty_like(x) for x in outs]
    tbytes = [int(o[-1].item()) for (_, o, _, _, _) in cases]
    engine.set_segment_bytes(256)
    try:
        # the HIP runtime loads the code object and sizes its kernel-private memory on the FIRST launch of a kernel
        # shape: have that behind us (through the ordinary entry point) before free memory is compared
        for (t, o, K, w, u), ref, tb in zip(cases, refs, tbytes):
            _call(engine, L.mdd_signatures, t, tb, o, K, w, u, a, b, ref)
        torch.cuda.synchronize()
        for ref in refs:
            ref.fill_(0)
        torch.cuda.synchronize()
        mem0 = torch.cuda.memory_allocated(engine.device)
        free0 = torch.cuda.mem_get_info(engine.device)[0]
        for (t, o, K, w, u), out, tb in zip(cases, outs, tbytes):           # two calls in a row, second one smaller
            _call(engine, L.mdd_signatures_noalloc, t, tb, o, K, w, u, a, b, out)
        engine.sync()
        assert torch.cuda.memory_allocated(engine.device) == mem0
        assert torch.cuda.mem_get_info(engine.device)[0] == free0          # no hipMalloc behind torch's back either
        for (t, o, K, w, u), ref, tb in zip(cases, refs, tbytes):
            _call(engine, L.mdd_signatures, t, tb, o, K, w, u, a, b, ref)
        engine.sync()
    finally:
        engine.set_segment_bytes(0)
    for out, ref in zip(outs, refs):
        assert torch.equal(out, ref)
    # ... and both equal the oracle (first case), so "equal" is not "equally wrong"
    from oracle import ref_c as RC
    exp = RC.signatures(t1.cpu().numpy(), o1.cpu().numpy().astype(np.uint64), 128, 13, "char")
    assert (as_u32(outs[0]) == exp).all()
    # default segments, same entry point again (third call on the same scratch)
    out3 = torch.empty_like(outs[0])
    _call(engine, L.mdd_signatures_noalloc, t1, tbytes[0], o1, 128, 13, "char", a, b, out3)
    engine.sync()
    assert torch.equal(out3, refs[0])


def test_noalloc_rejects_more_segments_than_the_scratch_holds(engine):
    """text_bytes / segment_bytes > 2^18 -> MDD_ERR_INVALID, nothing enqueued, output untouched"""
    L = engine.L
    buf, offs = pack_texts(["hello world, this is a document"] * 4)
    text, o = engine.to_device(buf), engine.to_device(offs)
    a, b = engine.perms(1, 128)
    out = torch.full((4, 128), 7, dtype=torch.int32, device=engine.device)
    engine.set_segment_bytes(256)
    try:
        fake_bytes = 256 * ((1 << 18) + 1)          # validated on the host before anything is launched
        big = torch.zeros(fake_bytes + TEXT_PAD, dtype=torch.uint8, device=engine.device)
        rc = L.mdd_signatures_noalloc(engine.ctx, _p(big), fake_bytes, _p(o), 4, 0, 13, 128, _p(a), _p(b), _p(out))
        assert rc == -1 and b"static scratch" in L.mdd_last_error()
        engine.sync()
        assert bool((out == 7).all())
        _call(engine, L.mdd_signatures_noalloc, text, int(offs[-1]), o, 128, 13, "char", a, b, out)
        ref = engine.signatures(text, o, 128)
        assert torch.equal(out, ref)
    finally:
        engine.set_segment_bytes(0)
