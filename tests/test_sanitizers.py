"""CPU suite (sanitizers): the two pieces of plain C / C++ that parse untrusted input -- the native
JSON-lines scanner + writer (map_neo_amd/csrc/mdd_io.cpp) and the C oracle (oracle/mdd_ref.c) -- built with
-fsanitize=address,undefined (`make asan`) and run as standalone drivers (tests/native/) on adversarial, mutated and
random input.  ASan / UBSan abort on the first finding, so exit code 0 means a clean run."""
import gzip
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(shutil.which("gcc") is None or shutil.which("g++") is None, reason="needs gcc / g++ (CPU container)")
# the harness preloads a library of its own; the sanitizer runtime only insists on coming first in the link order of the
# EXECUTABLE, which it does (static link order) -- verify_asan_link_order=0 silences the check about preloaded libraries
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:verify_asan_link_order=0", UBSAN_OPTIONS="print_stacktrace=1")


def _make(directory):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, directory), "asan"])


def test_jsonl_scanner_and_writer_under_asan_ubsan(tmp_path):
    _make("map_neo_amd/csrc")
    from test_hostio import TRICKY
    rng = np.random.Generator(np.random.PCG64(123))
    base = ("\n".join(TRICKY) + "\n").encode("utf8")
    shards = []

    def put(name, data, gz=False):
        p = tmp_path / name
        p.write_bytes(gzip.compress(data) if gz else data)
        shards.append(str(p))

    put("tricky.jsonl", base)
    put("tricky.jsonl.gz", base, gz=True)
    put("empty.jsonl", b"")
    put("blank.jsonl", b"\n\n  \n\t\n")
    put("nonl.jsonl", b'{"doc_id": 1, "text": "no newline at the end"}')
    put("truncated.gz", gzip.compress(base)[:-9])
    put("notgzip.gz", b"\x1f\x8b" + bytes(rng.integers(0, 256, 200, dtype=np.uint8)))
    put("huge_line.jsonl", b'{"text": "' + b"A b,c " * 200000 + b'"}\n')
    put("deep.jsonl", (b'{"x": ' + b"[" * 5000 + b"]" * 5000 + b', "text": "deep"}\n'))
    put("escapes.jsonl", ('{"text": "' + "\\ud83d\\ude00\\u0000\\ud800 \\udc00\\n" * 500 + '"}\n').encode())
    arr = np.frombuffer(base, dtype=np.uint8)
    for k in range(60):                       # mutations: flipped bytes, deleted / duplicated spans, cut tails
        m = arr.copy()
        kind = k % 4
        if kind == 0:
            idx = rng.integers(0, m.size, 40)
            m[idx] = rng.integers(0, 256, 40, dtype=np.uint8)
        elif kind == 1:
            a, b = sorted(rng.integers(0, m.size, 2).tolist())
            m = np.concatenate([m[:a], m[b:]])
        elif kind == 2:
            a, b = sorted(rng.integers(0, m.size, 2).tolist())
            m = np.concatenate([m[:b], m[a:]])
        else:
            m = m[: int(rng.integers(1, m.size))]
        put(f"mut{k}.jsonl", m.tobytes(), gz=(k % 8 == 7))
    for k in range(10):                       # pure noise, with newlines
        put(f"noise{k}.jsonl", bytes(rng.integers(0, 256, 5000, dtype=np.uint8)))
    r = subprocess.run([os.path.join(ROOT, "build", "san_io")] + shards, capture_output=True, text=True, timeout=300, env=ENV)
    assert r.returncode == 0 and "san_io ok" in r.stdout, r.stdout[-500:] + r.stderr[-4000:]
    # sanity: the unmutated shard really went through the scanner and came back out as JSON lines
    out = [json.loads(l) for l in open(str(tmp_path / "tricky.jsonl.out"), encoding="utf8")]
    assert len(out) > 5 and all(set(o) == {"doc_id", "text"} for o in out)


def _corpus_file(path, docs):
    offs = np.zeros(len(docs) + 1, np.uint64)
    offs[1:] = np.cumsum([len(d) for d in docs])
    with open(path, "wb") as f:
        f.write(np.array([len(docs), int(offs[-1])], np.uint64).tobytes())
        f.write(offs.tobytes())
        f.write(b"".join(docs))


def test_c_oracle_under_asan_ubsan(tmp_path, mini_docs):
    _make("oracle")
    rng = np.random.Generator(np.random.PCG64(7))
    cases = {
        "mini": [d.encode("utf8") for d in mini_docs],
        "empty": [],
        "blanks": [b"", b" ", b"   ", b"", b"a"],
        "noise": [bytes(rng.integers(0, 256, int(rng.integers(0, 300)), dtype=np.uint8)) for _ in range(200)],     # invalid utf-8
        "cut_utf8": ["数据清洗去重😀".encode("utf8")[:k] for k in range(1, 22)],                                      # truncated sequences at the end
        "continuation": [b"\x80" * 40, b"\xf0\x9f\x98", b"\xe4" * 30, b"ab \xc3"],
        "long": [b"word " * 30000, ("长" * 20000).encode("utf8")],
    }
    for name, docs in cases.items():
        p = tmp_path / f"{name}.bin"
        _corpus_file(str(p), docs)
        r = subprocess.run([os.path.join(ROOT, "build", "san_ref"), str(p)], capture_output=True, text=True, timeout=300, env=ENV)
        assert r.returncode == 0 and "san_ref ok" in r.stdout, name + "\n" + r.stdout[-300:] + r.stderr[-4000:]
