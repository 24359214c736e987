"""Self-test of the loop-back RCCL stand-in (tests/native/fake_rccl.cpp, TEST INFRASTRUCTURE) on the CPU
(FAKE_RCCL_HOSTCOPY: payloads move with memcpy): the GPU test of the library's exchange code
(tests/test_gpu_comm_loopback.py) is only as good as this transport, so its matching, its strictness (size mismatch,
unmatched operations time out instead of hanging) and its group-abort rule are pinned here."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "native", "libfake_rccl.so")
U8, U64 = 1, 5      # ncclUint8, ncclUint64


class UniqueId(C.Structure):        # ncclUniqueId travels BY VALUE into ncclCommInitRank
    _fields_ = [("internal", C.c_char * 128)]


@pytest.fixture(scope="module")
def fake():
    if not os.path.exists(FAKE):
        pytest.skip("tests/native/libfake_rccl.so not built (make -C map_neo_amd/csrc fake_rccl)")
    L = C.CDLL(FAKE)
    L.ncclGetUniqueId.argtypes = [C.c_void_p]
    L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    L.ncclCommDestroy.argtypes = [C.c_void_p]
    for f in (L.ncclSend, L.ncclRecv):
        f.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    L.fake_rccl_fail_send.argtypes = [C.c_uint64]
    L.fake_rccl_set_hostcopy(1)
    yield L
    L.fake_rccl_set_hostcopy(-1)
    L.fake_rccl_fail_send(0)


def _run(fake, world, body):
    uid = UniqueId()
    assert fake.ncclGetUniqueId(C.byref(uid)) == 0
    out, errs = [None] * world, []

    def run(r):
        try:
            comm = C.c_void_p()
            assert fake.ncclCommInitRank(C.byref(comm), world, uid, r) == 0
            out[r] = body(r, comm)
            fake.ncclCommDestroy(comm)
        except BaseException as e:      # noqa: BLE001
            errs.append((r, repr(e)))

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    return out


def test_grouped_variable_all_to_all_and_allgather(fake):
    world = 5
    rng = np.random.default_rng(3)
    counts = rng.integers(0, 40, (world, world))
    counts[2, :] = 0            # a rank that sends nothing
    counts[:, 4] = 0            # a rank that receives nothing
    send = [rng.integers(0, 255, int(counts[r].sum()), dtype=np.uint8) for r in range(world)]

    def body(r, comm):
        recv = np.zeros(int(counts[:, r].sum()), np.uint8)
        soff = np.concatenate([[0], np.cumsum(counts[r])])
        roff = np.concatenate([[0], np.cumsum(counts[:, r])])
        assert fake.ncclGroupStart() == 0
        for p in range(world):
            if counts[r, p]:
                assert fake.ncclSend(send[r].ctypes.data + int(soff[p]), int(counts[r, p]), U8, p, comm, None) == 0
            if counts[p, r]:
                assert fake.ncclRecv(recv.ctypes.data + int(roff[p]), int(counts[p, r]), U8, p, comm, None) == 0
        assert fake.ncclGroupEnd() == 0
        mine = np.array([r * 1000 + 7], np.uint64)
        allv = np.zeros(world, np.uint64)
        assert fake.ncclAllGather(mine.ctypes.data, allv.ctypes.data, 1, U64, comm, None) == 0
        return recv, allv

    out = _run(fake, world, body)
    for r in range(world):
        exp = np.concatenate([send[p][int(counts[p, :r].sum()): int(counts[p, :r + 1].sum())] for p in range(world)])
        assert (out[r][0] == exp).all()
        assert (out[r][1] == np.arange(world, dtype=np.uint64) * 1000 + 7).all()


def test_size_mismatch_and_unmatched_operations_are_errors(fake, monkeypatch):
    monkeypatch.setenv("FAKE_RCCL_TIMEOUT_S", "1")
    buf = [np.zeros(64, np.uint8) for _ in range(2)]

    def mismatch(r, comm):
        if r == 0:
            return fake.ncclSend(buf[0].ctypes.data, 16, U8, 1, comm, None)
        return fake.ncclRecv(buf[1].ctypes.data, 24, U8, 0, comm, None)

    out = _run(fake, 2, mismatch)
    assert out[1] == 4 and out[0] == 0        # ncclInvalidArgument on the receiving side; the sender is released

    def unmatched(r, comm):
        if r == 0:
            return fake.ncclRecv(buf[0].ctypes.data, 8, U8, 1, comm, None)     # rank 1 never sends
        return 0

    assert _run(fake, 2, unmatched)[0] == 2    # ncclSystemError after the time-out instead of a hang


def test_failed_call_inside_a_group_drops_the_group(fake):
    buf = [np.arange(32, dtype=np.uint8), np.zeros(32, np.uint8)]

    def body(r, comm):
        fake.fake_rccl_reset_thread()
        peer = 1 - r
        assert fake.ncclGroupStart() == 0
        rc_send = fake.ncclSend(buf[0].ctypes.data, 8, U8, peer, comm, None)
        fake.ncclRecv(buf[1].ctypes.data + 8 * r, 8, U8, peer, comm, None)
        rc_end = fake.ncclGroupEnd()
        return rc_send, rc_end, fake.fake_rccl_group_depth()

    fake.fake_rccl_fail_send(1)
    try:
        out = _run(fake, 2, body)
    finally:
        fake.fake_rccl_fail_send(0)
    assert out[0] == (3, 3, 0) and out[1] == (3, 3, 0)      # ncclInternalError from the call and again from GroupEnd
    out = _run(fake, 2, body)                               # and nothing of the dropped group is left behind
    assert out[0] == (0, 0, 0) and out[1] == (0, 0, 0)
