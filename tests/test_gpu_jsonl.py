"""GPU suite: JSON-lines field extraction ON THE DEVICE (csrc/mdd_jsonl.hip, Engine.jsonl_extract,
hostio.read_jsonl_shard_device) against the pure-Python reader -- the reference's own
`for idx, line in enumerate(f): json.loads(line)[field]` (ZH-1:40-46) -- on the adversarial lines of tests/test_hostio.py,
random documents, escapes at every chunk boundary, hypothesis-generated rows, CRLF files and a big shard."""
import json
import os
import sys

import numpy as np
import pytest

from map_neo_amd import hostio

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_hostio import TRICKY  # noqa: E402

pytestmark = pytest.mark.gpu


def _same(dev, py):
    assert dev["ids"] == py["ids"]
    assert dev["line_no"].tolist() == py["line_no"].tolist()
    assert dev["cand"].tolist() == py["cand"].tolist()
    assert dev["offsets"].cpu().numpy().tolist() == py["offsets"].tolist()
    nb = int(py["offsets"][-1])
    assert dev["text_bytes"] == nb
    assert dev["text"][:nb].cpu().numpy().tobytes() == py["text"].tobytes()
    assert not dev["text"][nb:].cpu().numpy().any()             # the padding stays zero
    assert dev["n_bad"] == py["n_bad"]


def _both(engine, path, content="text", idf="doc_id"):
    dev = hostio.read_jsonl_shard_device(engine, path, content, idf)
    py = hostio.read_jsonl_shard(path, content, idf, False, "skip", use_native=False)
    _same(dev, py)
    return dev, py


def test_adversarial_lines(engine, tmp_path):
    p = str(tmp_path / "t.jsonl")
    with open(p, "w", encoding="utf8", newline="\n") as f:
        f.write("\n".join(TRICKY))
    dev, py = _both(engine, p)
    assert py["n_bad"] >= 10 and len(py["cand"]) >= 15
    assert dev["device_lines"] >= 12 and dev["fallback_lines"] >= 10      # both routes are exercised
    # without an id field, and with the fields swapped
    _same(hostio.read_jsonl_shard_device(engine, p, "text", None), hostio.read_jsonl_shard(p, "text", None, False, "skip", use_native=False))
    _both(engine, p, "doc_id", "text")
    # invalid utf-8 in one line: that line is skipped, the others stay
    raw = open(p, "rb").read().replace(b'{"doc_id": "empty"', b'{"doc_id": "badutf8", "text": "\xff\xfe"}\n{"doc_id": "empty"')
    open(p, "wb").write(raw)
    dev = hostio.read_jsonl_shard_device(engine, p, "text", "doc_id")
    assert "badutf8" not in dev["ids"] and "empty" in dev["ids"] and "last" in dev["ids"]


def test_escapes_at_every_chunk_position(engine, tmp_path):
    """the decode kernel works in 64-byte chunks with carries: every escape form at every offset, long backslash runs"""
    esc = ['\\"', "\\\\", "\\/", "\\b", "\\f", "\\n", "\\r", "\\t", "\\u0041", "\\u00e9", "\\u4e2d", "\\ud83d\\ude00", "\\\\\\\\\\\\", '\\\\\\"',
           "\\u005c", "\\\\u0041", "\\\\\\u0041"]
    lines = []
    for e in esc:
        for k in range(0, 70):
            lines.append('{"doc_id": %d, "text": "%s%s|%s%s tail"}' % (len(lines), "a" * k, e, e, "é" if k % 3 == 0 else ""))
    # the 256-byte steps of the scan and decode kernels (printable ASCII, a dword per lane): whatever ends a run, at every
    # alignment around the first and second step
    for e in ['\\"', "\\n", "\\u00e9", "é", "中", "\x7f", "😀"]:
        for k in list(range(240, 340)) + list(range(560, 600, 3)):
            lines.append('{"doc_id": %d, "text": "%s%s%s"}' % (len(lines), "a" * k, e, "b" * 300))
    lines.append('{"text": "%s"}' % ("\\\\" * 200))
    lines.append('{"text": "%s\\""}' % ("\\\\" * 131))
    lines.append('{"text": "%s"}' % ("\\ud83d\\ude00" * 50))
    p = str(tmp_path / "e.jsonl")
    open(p, "w", encoding="utf8", newline="\n").write("\n".join(lines) + "\n")
    dev, py = _both(engine, p)
    assert dev["fallback_lines"] == 0 and dev["n_bad"] == 0 and len(py["cand"]) == len(lines)
    # ... and a raw control character there (not JSON: the line is handed back and is a bad line)
    bad = ['{"text": "%s\x01%s"}' % ("a" * k, "b" * 300) for k in range(250, 330, 5)]
    open(p, "w", encoding="utf8", newline="\n").write("\n".join(lines[:50] + bad) + "\n")
    dev, py = _both(engine, p)
    assert dev["n_bad"] == len(bad) and len(py["cand"]) == 50


def test_random_rows_and_line_endings(engine, tmp_path):
    rng = np.random.Generator(np.random.PCG64(5))
    pools = [(0x20, 0x7F), (0x80, 0x7FF), (0x800, 0xD7FF), (0xE000, 0xFFFF), (0x10000, 0x10FFFF), (0x00, 0x20)]
    rows = []
    for i in range(1500):
        n = int(rng.integers(0, 400))
        cps = [int(rng.integers(*pools[int(rng.integers(0, len(pools)))])) for _ in range(n)]
        t = "".join(chr(c) for c in cps)
        rows.append(json.dumps({"k": [i, {"text": "inner"}, [[], {}]], "doc_id": f"d{i}" if i % 7 else i, "text": t, "z": None},
                               ensure_ascii=bool(i % 2)))
    for name, sep, tail in (("lf", "\n", "\n"), ("crlf", "\r\n", "\r\n"), ("open", "\n", ""), ("blank", "\n\n  \n", "\n\n")):
        p = str(tmp_path / f"r_{name}.jsonl")
        open(p, "wb").write((sep.join(rows) + tail).encode("utf8"))
        dev, _ = _both(engine, p)
        assert dev["fallback_lines"] == 0
    # a lone "\r" is a line end for a text-mode reader: the whole shard takes the host route, same result
    p = str(tmp_path / "cr.jsonl")
    open(p, "wb").write(("\r".join(rows[:50]) + "\n").encode("utf8"))
    dev, py = _both(engine, p)
    assert dev["device_lines"] == 0 and len(py["cand"]) == 50
    # empty file, one blank line, gzip
    for name, data in (("empty", b""), ("nl", b"\n"), ("sp", b"  ")):
        p = str(tmp_path / f"{name}.jsonl")
        open(p, "wb").write(data)
        _both(engine, p)
    import gzip
    p = str(tmp_path / "g.jsonl.gz")
    with gzip.open(p, "wb") as f:
        f.write("\n".join(rows[:100]).encode("utf8"))
    _both(engine, p)


def test_batch_of_shards_equals_per_shard_host_reader(engine, tmp_path):
    """hostio.read_jsonl_batch_device: several shards in one pinned buffer and one extraction -- a shard without a final
    newline, an empty one, one made of blank lines, a gzip one (handed to the host reader), bad lines -- per-file rows,
    line numbers and documents equal the host reader's"""
    import gzip
    rng = np.random.Generator(np.random.PCG64(9))
    contents = []
    for k in range(7):
        rows = [json.dumps({"text": "shard %d row %d %s" % (k, i, "\u4e2d\u6587\n" * int(rng.integers(0, 4))), "i": i}, ensure_ascii=bool(i % 2))
                for i in range(int(rng.integers(1, 60)))]
        if k == 2:
            rows.insert(1, "not json")
            rows.insert(3, '{"text": "nan", "v": NaN}')
        contents.append("\n".join(rows) + ("" if k % 2 else "\n"))
    contents += ["", "\n  \n\n", '{"text": "single line, no newline"}']
    paths = []
    for k, c in enumerate(contents):
        p = str(tmp_path / f"s{k:02d}.jsonl")
        open(p, "wb").write(c.encode("utf8"))
        paths.append(p)
    gz = str(tmp_path / "s99.jsonl.gz")
    with gzip.open(gz, "wb") as f:
        f.write(contents[0].encode("utf8"))
    paths.insert(4, gz)
    r = hostio.read_jsonl_batch_device(engine, paths, "text", None)
    assert r["host_files"] == [4] and r["fallback_lines"] >= 2
    offs = r["offsets"].cpu().numpy()
    text = r["text"].cpu().numpy()
    docs_by_file = {k: [] for k in range(len(paths))}
    for d, row in enumerate(r["cand"].tolist()):
        docs_by_file[int(r["file_of"][row])].append((int(r["line_no"][row]), text[offs[d]:offs[d + 1]].tobytes()))
    for k, p in enumerate(paths):
        if k == 4:
            assert docs_by_file[k] == []
            continue
        h = hostio.read_jsonl_shard(p, "text", None, False, "skip", use_native=False)
        exp = [(int(h["line_no"][row]), h["text"][int(h["offsets"][d]):int(h["offsets"][d + 1])].tobytes()) for d, row in enumerate(h["cand"].tolist())]
        assert docs_by_file[k] == exp, p
        assert int(r["n_bad"][k]) == h["n_bad"]
        assert sorted(r["line_no"][r["file_of"] == k].tolist()) == h["line_no"].tolist()


def test_device_reader_property(engine, tmp_path):
    """hypothesis (the strategy of tests/test_hostio.py): arbitrary JSON rows, nested values, both ensure_ascii settings"""
    from hypothesis import HealthCheck, given, settings, strategies as st
    text = st.text(alphabet=st.characters(blacklist_categories=("Cs",)), max_size=80)
    scalar = st.one_of(st.none(), st.booleans(), st.integers(-10**12, 10**12), st.floats(allow_nan=False, allow_infinity=False), text)
    value = st.recursive(scalar, lambda c: st.one_of(st.lists(c, max_size=3), st.dictionaries(text, c, max_size=3)), max_leaves=6)
    row = st.fixed_dictionaries({}, optional={"doc_id": st.one_of(text, st.integers(0, 10**9), st.none()),
                                              "raw_content": st.one_of(text, st.none(), st.integers()),
                                              "extra": value, "raw_content2": text})
    counter = [0]

    @settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
    @given(rows=st.lists(row, max_size=12), ascii_=st.booleans(), blank=st.booleans(), nan=st.booleans())
    def run(rows, ascii_, blank, nan):
        counter[0] += 1
        path = str(tmp_path / f"h{counter[0]}.jsonl")
        lines = [json.dumps(r, ensure_ascii=ascii_) for r in rows]
        if nan:
            lines.append('{"raw_content": "python accepts this", "v": NaN}')
        if blank:
            lines.insert(len(lines) // 2, "   ")
        open(path, "wb").write(("\n".join(lines) + ("\n" if lines else "")).encode("utf8"))
        _both(engine, path, "raw_content", "doc_id")

    run()


def test_deep_nesting_and_long_lines(engine, tmp_path):
    deep_ok = '{"d": ' + "[" * 60 + "]" * 60 + ', "text": "sixty levels"}'
    deep_fb = '{"d": ' + "[" * 80 + "]" * 80 + ', "text": "eighty levels: handed back"}'
    objs = '{"d": ' + '{"a": ' * 30 + "1" + "}" * 30 + ', "text": "nested objects"}'
    bad_close = '{"d": [{"a": 1]}, "text": "mismatched brackets"}'
    long_doc = "long " * 300000                                   # one 1.5 MB line
    lines = [deep_ok, deep_fb, objs, bad_close, json.dumps({"text": long_doc, "doc_id": "L"}), '{"text": "after"}']
    p = str(tmp_path / "d.jsonl")
    open(p, "w", encoding="utf8", newline="\n").write("\n".join(lines))
    dev, py = _both(engine, p)
    assert dev["fallback_lines"] == 2 and dev["n_bad"] == 1 and len(py["cand"]) == 5


def test_signatures_from_device_reader_equal_host_reader(engine, tmp_path):
    """the point of it: the extracted text goes straight into mdd_signatures"""
    from map_neo_amd import corpus
    from map_neo_amd.minhash import padded_size
    spec = corpus.CorpusSpec(3000, 700, seed=3, lang="mixed")
    text, offs = corpus.generate(spec, device="cpu")
    t, o = text.numpy(), offs.numpy()
    p = str(tmp_path / "c.jsonl")
    with open(p, "w", encoding="utf8", newline="\n") as f:
        for d in range(spec.n_docs):
            doc = bytes(t[o[d]:o[d + 1]]).decode("utf8")
            f.write(json.dumps({"text": doc + ('\n"quoted"\t' if d % 5 == 0 else ""), "n": d}, ensure_ascii=bool(d % 2)) + "\n")
    dev, py = _both(engine, p, "text", None)
    sig = engine.signatures(dev["text"], dev["offsets"], text_bytes=dev["text_bytes"])
    buf = np.zeros(padded_size(py["text"].size), np.uint8)
    buf[: py["text"].size] = py["text"]
    exp = engine.signatures(engine.to_device(buf), engine.to_device(py["offsets"]))
    import torch
    assert torch.equal(sig, exp)


def test_wave_scanner_equals_thread_scanner_on_mutated_lines(engine, tmp_path, monkeypatch):
    """the scan kernel (a wave per line, an automaton over 64-byte chunks) against the first implementation (a thread per line
    walking the bytes: MDD_JSONL_SCAN=thread) on valid rows and on the same rows with bytes flipped, dropped and inserted --
    broken escapes, invalid utf-8, unbalanced brackets, stray quotes at every chunk offset: identical status, content range,
    decoded length and id token for every line; and what the device vouches for equals the pure-Python reader"""
    import torch
    from map_neo_amd.minhash import padded_size
    rng = np.random.Generator(np.random.PCG64(11))
    pools = [(0x20, 0x7F), (0x80, 0x7FF), (0x800, 0xD7FF), (0xE000, 0xFFFF), (0x10000, 0x10FFFF), (0x00, 0x20)]
    base_rows = [t for t in TRICKY if "\n" not in t and "\r" not in t]
    for i in range(600):
        n = int(rng.integers(0, 300))
        t = "".join(chr(int(rng.integers(*pools[int(rng.integers(0, len(pools)))]))) for _ in range(n))
        base_rows.append(json.dumps({"k": [i, {"text": "in"}, [[], {}], -1.5e-3, True], "doc_id": [f"d{i}", i, None, {"a": i}][i % 4], "text": t,
                                     "z": None, "pad": "x" * int(rng.integers(0, 70))}, ensure_ascii=bool(i % 2)))
    inserts = [b"\\", b'"', b"\\u", b"\\ud83d", b"\\ude00", b"\xc3", b"\xa9", b"\xe4\xb8", b"\xf0\x9f", b"\xff", b"{", b"}", b"[", b"]", b",", b":",
               b"\t", b"\x01", b"\x1c", b" ", b"0", b"-", b"e", b".", b"tru", b"null", b"\\x"]
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
    raw = engine.to_device(buf)
    got = {}
    for how in ("thread", "wave"):
        monkeypatch.setenv("MDD_JSONL_SCAN", how)
        x = engine.jsonl_extract(raw, len(data), "text", "doc_id")
        torch.cuda.synchronize()
        got[how] = x
    a, b = got["thread"], got["wave"]
    assert a["n_lines"] == b["n_lines"] == len(lines)
    diff = np.flatnonzero(a["status"] != b["status"])
    assert diff.size == 0, [(int(i), int(a["status"][i]), int(b["status"][i]), lines[int(i)][:200]) for i in diff[:5]]
    assert (a["status"] == 0).sum() > 800 and (a["status"] == 2).sum() > 800 and (a["status"] == 1).sum() > 20, np.bincount(a["status"])
    for key in ("ok_lines", "id_kind", "id_beg", "id_end"):
        assert (a[key] == b[key]).all(), key
    assert a["text_bytes"] == b["text_bytes"] and torch.equal(a["offsets"], b["offsets"]) and torch.equal(a["text"], b["text"])
    monkeypatch.delenv("MDD_JSONL_SCAN")
    # (the pure-Python reader opens the shard as utf-8 text: it gets the lines that decode, plus two that are blank to
    # str.strip but not to the device)
    valid = []
    for ln in lines:
        try:
            ln.decode("utf8")
            valid.append(ln)
        except UnicodeDecodeError:
            pass
    assert len(valid) > len(lines) // 2
    valid[5:5] = ["\u0085\u00a0".encode("utf8"), b"\x1c \x0b"]
    p = str(tmp_path / "m.jsonl")
    open(p, "wb").write(b"\n".join(valid) + b"\n")
    _both(engine, p)
