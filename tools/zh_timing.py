"""Where the fused ZH command's wall time goes on this box: per-batch ingest timing (MDD_TIMING=1),
and a micro-benchmark of the ways ZH-5's output can be produced on the file system that holds the shards:
kernel-side copies into new files, into existing files, and an in-place shift through a shared mapping.

    python tools/zh_timing.py [docs] [shards] [dir]
"""
import ctypes
import json
import mmap
import os
import shutil
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.getcwd())
from map_neo_amd import corpus
from map_neo_amd.cli import zh_dedup

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
shards = int(sys.argv[2]) if len(sys.argv) > 2 else 64
base = sys.argv[3] if len(sys.argv) > 3 else None
spec = corpus.CorpusSpec(n, 2048, seed=9, lang="zh")
text, offs = corpus.generate_chunked(spec, "cpu")
tb = text.numpy().tobytes()
o = offs.numpy()
root = tempfile.mkdtemp(prefix="mdd_zh_t_", dir=base)
inp = os.path.join(root, "in")
os.makedirs(inp)
per = (n + shards - 1) // shards
for s in range(shards):
    with open(os.path.join(inp, f"part-{s:04d}.jsonl"), "w", encoding="utf8") as f:
        for i in range(s * per, min(n, (s + 1) * per)):
            f.write(json.dumps({"text": tb[o[i]:o[i + 1]].decode(), "url": "x"}, ensure_ascii=False) + "\n")
print(subprocess.run(["df", "-T", root], capture_output=True, text=True).stdout, flush=True)
print("cpus allowed:", len(os.sched_getaffinity(0)), flush=True)
O2 = os.path.join(root, "out")
import torch
if torch.cuda.is_available():
    from map_neo_amd.minhash import Engine
    for _ in range(3):
        t0 = time.time()
        e = Engine(0)
        print("Engine(0): %.1f ms" % ((time.time() - t0) * 1e3), flush=True)
        del e
for rep in range(4 if torch.cuda.is_available() else 0):
    shutil.rmtree(O2, ignore_errors=True)
    if rep == 3:
        os.environ["MDD_TIMING"] = "1"
    t0 = time.time()
    zh_dedup.main(["--input_dir", inp, "--output_dir", O2, "--ingest", "device", "--workers", "16"])
    print("wall %.3f" % (time.time() - t0), zh_dedup.last_stats, flush=True)

files = sorted(os.listdir(inp))
total = sum(os.path.getsize(os.path.join(inp, f)) for f in files)


def copy_new(f, out):
    with open(os.path.join(inp, f), "rb") as fi, open(os.path.join(out, f), "wb") as fo:
        sz, pos = os.fstat(fi.fileno()).st_size, 0
        while pos < sz:
            pos += os.copy_file_range(fi.fileno(), fo.fileno(), sz - pos, pos)


def copy_over(f, out):
    with open(os.path.join(inp, f), "rb") as fi, open(os.path.join(out, f), "r+b") as fo:
        sz, pos = os.fstat(fi.fileno()).st_size, 0
        while pos < sz:
            pos += os.copy_file_range(fi.fileno(), fo.fileno(), sz - pos, pos, pos)


def shift_mmap(f, out):      # in-place: drop 200 bytes every 160 KB (what ZH-5 does to a speculative copy)
    with open(os.path.join(out, f), "r+b") as fo:
        sz = os.fstat(fo.fileno()).st_size
        m = mmap.mmap(fo.fileno(), sz)
        a = ctypes.addressof(ctypes.c_char.from_buffer(m))
        src = dst = 0
        while src < sz:
            run = min(160 << 10, sz - src)
            ctypes.memmove(a + dst, a + src, run)
            dst += run
            src += run + 200
        del a
        m.close()
        os.ftruncate(fo.fileno(), dst)


def shift_rw(f, out):        # the same through pread / pwrite (no mapping)
    fd = os.open(os.path.join(out, f), os.O_RDWR)
    sz = os.fstat(fd).st_size
    src = dst = 0
    while src < sz:
        run = min(160 << 10, sz - src)
        os.pwrite(fd, os.pread(fd, run, src), dst)
        dst += run
        src += run + 200
    os.ftruncate(fd, dst)
    os.close(fd)


for name, fn, fresh in (("copy_file_range into new files", copy_new, True), ("copy_file_range over existing files", copy_over, False),
                        ("in-place shift through mmap", shift_mmap, False), ("copy_file_range over existing files", copy_over, False),
                        ("in-place shift through pread/pwrite", shift_rw, False)):
    for threads in (16, 32):
        out = os.path.join(root, "mb")
        if fresh:
            shutil.rmtree(out, ignore_errors=True)
            os.makedirs(out)
        t0 = time.time()
        with ThreadPoolExecutor(threads) as pool:
            list(pool.map(lambda f: fn(f, out), files))
        dt = time.time() - t0
        print("%-40s %2d threads: %.3f s = %.1f GB/s" % (name, threads, dt, total / dt / 1e9), flush=True)
shutil.rmtree(root)
