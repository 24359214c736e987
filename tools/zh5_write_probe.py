"""ZH-5's output side alone (remove_dup.py:7-13: copy a shard without its dropped lines): which system calls move the
kept byte ranges fastest on the file system at hand.  64 files of 32 MB, lines of ~2 KB, 1.2 % of the lines dropped.

    python tools/zh5_write_probe.py [dir]
"""
import mmap
import os
import shutil
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

base = sys.argv[1] if len(sys.argv) > 1 else None
root = tempfile.mkdtemp(prefix="mdd_zh5_", dir=base)
inp = os.path.join(root, "in")
os.makedirs(inp)
rng = np.random.default_rng(1)
N_FILES, LINES = 64, 16000
starts, drops = {}, {}
for k in range(N_FILES):
    lens = rng.integers(1000, 3100, LINES)
    body = rng.integers(97, 123, int(lens.sum()), dtype=np.uint8)
    st = np.concatenate(([0], np.cumsum(lens)))
    body[st[1:] - 1] = 10
    name = "part-%04d.jsonl" % k
    body.tofile(os.path.join(inp, name))
    starts[name] = st
    drops[name] = np.sort(rng.choice(LINES, LINES // 85, replace=False))
files = sorted(starts)
total = sum(os.path.getsize(os.path.join(inp, f)) for f in files)


def runs_of(f):
    st, d = starts[f], drops[f]
    a = np.concatenate(([0], st[d + 1]))
    b = np.concatenate((st[d], [st[-1]]))
    keep = b > a
    return a[keep].tolist(), b[keep].tolist()


def m_copy_file_range(f, out):
    a, b = runs_of(f)
    with open(os.path.join(inp, f), "rb") as fi, open(os.path.join(out, f), "wb") as fo:
        for lo, hi in zip(a, b):
            while lo < hi:
                lo += os.copy_file_range(fi.fileno(), fo.fileno(), hi - lo, lo)


def m_sendfile(f, out):
    a, b = runs_of(f)
    with open(os.path.join(inp, f), "rb") as fi, open(os.path.join(out, f), "wb") as fo:
        for lo, hi in zip(a, b):
            while lo < hi:
                lo += os.sendfile(fo.fileno(), fi.fileno(), lo, hi - lo)


def m_mmap_writev(f, out):
    a, b = runs_of(f)
    with open(os.path.join(inp, f), "rb") as fi:
        m = mmap.mmap(fi.fileno(), 0, prot=mmap.PROT_READ)
        mv = memoryview(m)
        fd = os.open(os.path.join(out, f), os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        try:
            iov = [mv[lo:hi] for lo, hi in zip(a, b)]
            for i in range(0, len(iov), 1024):
                chunk = iov[i:i + 1024]
                want = sum(len(x) for x in chunk)
                got = os.writev(fd, chunk)
                while got < want:                      # short write: finish the remainder piecewise
                    for x in chunk:
                        if got >= len(x):
                            got -= len(x)
                            continue
                        os.write(fd, x[got:])
                        got = 0
                    break
        finally:
            os.close(fd)
            del iov, chunk
            mv.release()
            m.close()


def m_read_writev(f, out):
    a, b = runs_of(f)
    with open(os.path.join(inp, f), "rb") as fi:
        data = fi.read()
    mv = memoryview(data)
    fd = os.open(os.path.join(out, f), os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    iov = [mv[lo:hi] for lo, hi in zip(a, b)]
    for i in range(0, len(iov), 1024):
        os.writev(fd, iov[i:i + 1024])
    os.close(fd)


def m_whole_file(f, out):      # upper bound: no lines dropped, one call per file
    with open(os.path.join(inp, f), "rb") as fi, open(os.path.join(out, f), "wb") as fo:
        sz, pos = os.fstat(fi.fileno()).st_size, 0
        while pos < sz:
            pos += os.copy_file_range(fi.fileno(), fo.fileno(), sz - pos, pos)


print("cpus allowed:", len(os.sched_getaffinity(0)), " bytes:", total, flush=True)
for rep in range(2):
    for name, fn in (("copy_file_range per run", m_copy_file_range), ("sendfile per run", m_sendfile), ("mmap + writev", m_mmap_writev),
                     ("read + writev", m_read_writev), ("copy_file_range whole file", m_whole_file)):
        for threads in (16, 32, 64):
            out = os.path.join(root, "out")
            shutil.rmtree(out, ignore_errors=True)
            os.makedirs(out)
            t0 = time.time()
            with ThreadPoolExecutor(threads) as pool:
                list(pool.map(lambda f: fn(f, out), files))
            dt = time.time() - t0
            print("%-28s %2d threads: %.3f s = %5.1f GB/s" % (name, threads, dt, total / dt / 1e9), flush=True)
shutil.rmtree(root)
