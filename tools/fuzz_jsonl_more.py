#!/usr/bin/env python3
"""Extended run of tests/test_gpu_jsonl.py::test_wave_scanner_equals_thread_scanner_on_mutated_lines: per seed ~8 000 lines
(valid rows and the same rows with bytes flipped, dropped and inserted), the wave-per-line scan kernel against the
thread-per-line one (status, content range, decoded text, id token of every line) and against the pure-Python reader
(json.loads per line) on the lines that decode as utf-8.  python tools/fuzz_jsonl_more.py [n_seeds] [first_seed]"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from map_neo_amd import hostio
from map_neo_amd.minhash import Engine, padded_size
from test_hostio import TRICKY

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
first_seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # a later range of seeds: [first_seed, first_seed + n_seeds)
eng = Engine(0)
pools = [(0x20, 0x7F), (0x80, 0x7FF), (0x800, 0xD7FF), (0xE000, 0xFFFF), (0x10000, 0x10FFFF), (0x00, 0x20)]
inserts = [b"\\", b'"', b"\\u", b"\\ud83d", b"\\ude00", b"\\udc00\\ud800", b"\xc3", b"\xa9", b"\xe4\xb8", b"\xf0\x9f", b"\xff", b"\xed\xa0\x80", b"{", b"}",
           b"[", b"]", b",", b":", b"\t", b"\x01", b"\x1c", b" ", b"0", b"-", b"e", b".", b"tru", b"null", b"\\x", b"\\\\", b'\\"']
bad = total = 0
tmp = tempfile.mkdtemp(prefix="mdd_fj_")
for seed in range(first_seed, first_seed + n_seeds):
    rng = np.random.Generator(np.random.PCG64(90_000 + seed))
    base_rows = [t for t in TRICKY if "\n" not in t and "\r" not in t]
    for i in range(600):
        n = int(rng.integers(0, [300, 300, 1200, 40][i % 4]))
        pool_set = [pools, pools[:1], pools[:1] + pools[2:3], pools][i % 4]
        t = "".join(chr(int(rng.integers(*pool_set[int(rng.integers(0, len(pool_set)))]))) for _ in range(n))
        base_rows.append(json.dumps({"k": [i, {"text": "in"}, [[], {}], -1.5e-3, True], "doc_id": [f"d{i}", i, None, {"a": i}][i % 4], "text": t,
                                     "z": None, "pad": "x" * int(rng.integers(0, 400))}, ensure_ascii=bool(i % 2)))
    lines = []
    for r in base_rows:
        raw = r.encode("utf8")
        lines.append(raw)
        for _ in range(12):
            m = bytearray(raw)
            for _ in range(int(rng.integers(1, 4))):
                pos = int(rng.integers(0, len(m) + 1))
                kind = int(rng.integers(0, 3))
                if kind == 0 and pos < len(m):
                    del m[pos]
                elif kind == 1 and pos < len(m):
                    m[pos] = int(rng.integers(0, 256))
                else:
                    m[pos:pos] = inserts[int(rng.integers(0, len(inserts)))]
            lines.append(bytes(m).replace(b"\n", b" ").replace(b"\r", b" "))
    data = b"\n".join(lines) + b"\n"
    buf = np.zeros(padded_size(len(data)), np.uint8)
    buf[: len(data)] = np.frombuffer(data, np.uint8)
    raw = eng.to_device(buf)
    got = {}
    for how in ("thread", "wave"):
        os.environ["MDD_JSONL_SCAN"] = how
        got[how] = eng.jsonl_extract(raw, len(data), "text", "doc_id")
        torch.cuda.synchronize()
    os.environ.pop("MDD_JSONL_SCAN")
    a, b = got["thread"], got["wave"]
    same = (a["n_lines"] == b["n_lines"] == len(lines) and (a["status"] == b["status"]).all() and (a["ok_lines"] == b["ok_lines"]).all()
            and (a["id_kind"] == b["id_kind"]).all() and (a["id_beg"] == b["id_beg"]).all() and (a["id_end"] == b["id_end"]).all()
            and a["text_bytes"] == b["text_bytes"] and torch.equal(a["offsets"], b["offsets"]) and torch.equal(a["text"], b["text"]))
    if not same:
        bad += 1
        d = np.flatnonzero(a["status"] != b["status"])
        print(f"seed {seed}: scanners DIFFER", [(int(i), int(a['status'][i]), int(b['status'][i]), lines[int(i)][:160]) for i in d[:3]], flush=True)
    valid = []
    for ln in lines:
        try:
            ln.decode("utf8")
            valid.append(ln)
        except UnicodeDecodeError:
            pass
    p = os.path.join(tmp, "m.jsonl")
    open(p, "wb").write(b"\n".join(valid) + b"\n")
    dev = hostio.read_jsonl_shard_device(eng, p, "text", "doc_id")
    py = hostio.read_jsonl_shard(p, "text", "doc_id", False, "skip", use_native=False)
    nb = int(py["offsets"][-1])
    ok = (dev["ids"] == py["ids"] and dev["line_no"].tolist() == py["line_no"].tolist() and dev["cand"].tolist() == py["cand"].tolist()
          and dev["offsets"].cpu().numpy().tolist() == py["offsets"].tolist() and dev["text"][:nb].cpu().numpy().tobytes() == py["text"].tobytes()
          and dev["n_bad"] == py["n_bad"])
    if not ok:
        bad += 1
        print(f"seed {seed}: device reader != Python reader", flush=True)
    total += len(lines)
    if seed % 5 == 0:
        print(f"seed {seed}: ok ({len(lines)} lines, status counts {np.bincount(a['status'], minlength=4).tolist()}, device lines {dev['device_lines']})", flush=True)
print("mismatching seeds:", bad, "of", n_seeds, "--", total, "lines")
sys.exit(1 if bad else 0)
