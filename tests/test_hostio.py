"""CPU suite: native JSON-lines ingest (libmdd_io.so) against the pure-Python path on tricky
input.  The native scanner must either agree with json.loads or hand the line back."""
import gzip
import json
import os

import numpy as np
import pytest

from map_neo_amd import hostio
from oracle import ref_py as R

TRICKY = [
    '{"doc_id": "a/1", "text": "plain ascii text, with Punctuation!  and   spaces\\tTabs"}',
    '{"text": "escapes: \\" \\\\ \\/ \\b \\f \\n \\r \\t \\u0041 \\u00e9 \\u4e2d \\ud83d\\ude00 end", "doc_id": 7}',
    '{"doc_id": "dup", "text": "first", "text": "second wins like a python dict"}',
    '{"meta": {"text": "nested must not match", "doc_id": "nested"}, "doc_id": "outer", "text": "top level value"}',
    '{"doc_id": "arr", "arr": [1, 2.5, -3e10, true, false, null, "x", {"text": "no"}, []], "text": "after array"}',
    '{"doc_id": "nul", "text": null}',
    '{"doc_id": "num", "text": 12345}',
    '{"doc_id": "missing"}',
    '{}',
    '{"doc_id": null, "text": "null id"}',
    '{"doc_id": "lone", "text": "lone surrogate \\ud83d here"}',
    '{"doc_id": "lone2", "text": "low first \\udc00\\ud800"}',
    '{"doc_id": "pairsplit", "text": "\\ud83d\\u0041"}',
    '{"doc_id": "uni", "text": "Ünïcödé ＦＵＬＬ　ｗｉｄｔｈ İstanbul ǅ ß ΣΑΣ  中文，标点。"}',
    '{"doc_id": "ws", "text": " \\u001c lead \\u001f\\u000b\\u000c trail \\n"}',
    '{"doc_id": "nbsp", "text": "a\\u00a0b\\u2003c"}',
    '  {"doc_id" : "spaces" ,\t"text" :\t"ws between tokens" }  \r',
    '{"doc_id": "esc\\u0041id", "text": "escaped id"}',
    '{"te\\u0078t": "escaped key equals text", "doc_id": "esckey"}',
    '{"doc_id": "nan", "v": NaN, "text": "python accepts NaN"}',
    '{"doc_id": "inf", "v": -Infinity, "text": "and -Infinity"}',
    '{"doc_id": "badnum", "v": 01, "text": "leading zero is invalid"}',
    '{"doc_id": "trail", "text": "trailing garbage"} xyz',
    '{"doc_id": "ctrl", "text": "raw control \x01 char"}',
    '{"doc_id": "tab", "text": "raw\ttab is invalid in strict json"}',
    '{"doc_id": "trunc", "text": "unterminated',
    '["not", "an", "object"]',
    '"just a string"',
    '42',
    '{"doc_id": "badesc", "text": "bad \\x escape"}',
    '{"doc_id": "badu", "text": "bad \\u12g4 escape"}',
    '{"doc_id": "comma", "text": "trailing comma",}',
    '{"doc_id": "single", \'text\': "single quotes"}',
    '{"doc_id": "deep", "d": ' + '[' * 50 + ']' * 50 + ', "text": "deep nesting"}',
    '{"doc_id": "empty", "text": ""}',
    '{"doc_id": "del", "text": "del \x7f char and high ascii punct ~`^"}',
    '',
    '   ',
    '\x1c\x1d',
    '{"doc_id": "last", "text": "no newline at end"}',
]


def _write(tmp_path, gz):
    data = ("\n".join(TRICKY)).encode("utf8")
    # one line with invalid utf-8 bytes in the middle of the file
    data = data.replace(b'{"doc_id": "empty"', b'{"doc_id": "badutf8", "text": "\xff\xfe"}\n{"doc_id": "empty"')
    p = tmp_path / ("t.jsonl.gz" if gz else "t.jsonl")
    if gz:
        with gzip.open(p, "wb") as f:
            f.write(data)
    else:
        p.write_bytes(data)
    return str(p)


def _same(a, b):
    assert a["ids"] == b["ids"]
    assert (a["line_no"] == b["line_no"]).all()
    assert (a["cand"] == b["cand"]).all()
    assert (a["offsets"] == b["offsets"]).all()
    assert a["text"].tobytes() == b["text"].tobytes()
    assert a["n_bad"] == b["n_bad"]


@pytest.mark.parametrize("gz", [False, True])
@pytest.mark.parametrize("normalize_en", [False, True])
def test_native_equals_python(tmp_path, gz, normalize_en):
    assert hostio._io() is not None, "libmdd_io.so missing: make -C map_neo_amd/csrc"
    path = _write(tmp_path, gz)
    if not gz:
        # python's text reader refuses the file with invalid utf-8 as a whole; compare on a clean copy
        clean = [l for l in TRICKY]
        open(path, "w", encoding="utf8", newline="\n").write("\n".join(clean))
    else:
        with gzip.open(path, "wt", encoding="utf8", newline="\n") as f:
            f.write("\n".join(TRICKY))
    nat = hostio.read_jsonl_shard(path, "text", "doc_id", normalize_en, "skip", use_native=True)
    py = hostio.read_jsonl_shard(path, "text", "doc_id", normalize_en, "skip", use_native=False)
    _same(nat, py)
    assert py["n_bad"] >= 10 and len(py["cand"]) >= 15
    # spot checks against json.loads directly
    rows = {}
    for l in TRICKY:
        try:
            r = json.loads(l)
            if isinstance(r, dict) and isinstance(r.get("doc_id"), str):
                rows[r["doc_id"]] = r
        except ValueError:
            pass
    o = nat["offsets"]
    for k, ridx in enumerate(nat["cand"].tolist()):
        rid = nat["ids"][ridx]
        if isinstance(rid, str) and rid in rows:
            want = rows[rid]["text"]
            want = R.normalize_text(want) if normalize_en else want
            assert nat["text"][int(o[k]):int(o[k + 1])].tobytes().decode("utf8") == want, rid


def test_native_invalid_utf8_line_is_skipped(tmp_path):
    path = _write(tmp_path, gz=False)
    nat = hostio.read_jsonl_shard(path, "text", "doc_id", False, "skip", use_native=True)
    assert "badutf8" not in nat["ids"] and "empty" in nat["ids"] and "last" in nat["ids"]


def test_native_ascii_normalisation_exhaustive():
    """every ASCII byte, in context, through the native EN normaliser == the oracle's normalize_text (EN:31-40)"""
    import os
    import tempfile
    texts = ["".join(chr(c) for c in range(1, 128)), "".join(f" x{chr(c)}Y {chr(c)}" for c in range(1, 128)),
             "A  B\t\tC\n\nD\r\nE\x0bF\x0cG\x1cH\x1dI\x1eJ\x1fK", "  ...  ", "!!!", "MiXeD--CaSe__123"]
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "n.jsonl")
        with open(p, "w", encoding="utf8", newline="\n") as f:
            for i, t in enumerate(texts):
                f.write(json.dumps({"doc_id": i, "text": t}) + "\n")
        nat = hostio.read_jsonl_shard(p, "text", "doc_id", True, "raise", use_native=True)
    o = nat["offsets"]
    got = [nat["text"][int(o[k]):int(o[k + 1])].tobytes().decode() for k in range(len(texts))]
    assert got == [R.normalize_text(t) for t in texts] == [hostio.normalize_text(t) for t in texts]


def test_random_roundtrip_native_vs_python(tmp_path):
    rng = np.random.Generator(np.random.PCG64(5))
    pools = [(0x20, 0x7F), (0x80, 0x7FF), (0x800, 0xD7FF), (0xE000, 0xFFFF), (0x10000, 0x10FFFF), (0x00, 0x20)]
    p = tmp_path / "r.jsonl"
    with open(p, "w", encoding="utf8", newline="\n") as f:
        for i in range(400):
            n = int(rng.integers(0, 80))
            cps = [int(rng.integers(*pools[int(rng.integers(0, len(pools)))])) for _ in range(n)]
            t = "".join(chr(c) for c in cps)
            f.write(json.dumps({"k": [i, {"text": "inner"}], "doc_id": f"d{i}", "text": t},
                               ensure_ascii=bool(i % 2)) + "\n")
    for norm in (False, True):
        _same(hostio.read_jsonl_shard(str(p), "text", "doc_id", norm, "raise", True),
              hostio.read_jsonl_shard(str(p), "text", "doc_id", norm, "raise", False))


def test_parquet_shard_reader_matches_row_path(tmp_path):
    """Arrow-buffer ingest == the per-row Python path (nulls, empty, non-ASCII, several row groups)"""
    import pyarrow as pa
    import pyarrow.parquet as pq
    from map_neo_amd import hostio
    rng = np.random.Generator(np.random.PCG64(8))
    pool = ["Hello,  World!", None, "naïve Café  x", "", "  ALL caps\tHERE ", "中文文档", "a" * 300, "x\x1cy", "😀 ok"]
    contents = [pool[int(i)] for i in rng.integers(0, len(pool), 500)]
    ids = [None if i % 50 == 7 else f"d{i}" for i in range(500)]
    path = str(tmp_path / "s.parquet")
    pq.write_table(pa.table({"doc_id": ids, "raw_content": contents, "n": list(range(500))}), path, row_group_size=64)
    for norm in (False, True):
        r = hostio.read_parquet_shard(path, "raw_content", "doc_id", normalize_en=norm)
        assert r["ids"] == ids
        exp_cand = [i for i, c in enumerate(contents) if c is not None]
        assert r["cand"].tolist() == exp_cand
        docs = {}
        for pos, k in enumerate(r["ready"].tolist()):
            docs[k] = bytes(r["text"][int(r["offsets"][pos]):int(r["offsets"][pos + 1])]).decode("utf8")
        raw = {}
        for pos, k in enumerate(r["raw"].tolist()):
            raw[k] = bytes(r["raw_text"][int(r["raw_offsets"][pos]):int(r["raw_offsets"][pos + 1])]).decode("utf8")
        assert sorted(list(docs) + list(raw)) == list(range(len(exp_cand)))
        for k, i in enumerate(exp_cand):
            if k in docs:
                assert docs[k] == (R.normalize_text(contents[i]) if norm else contents[i])
                assert not norm or hostio._ftfy is not None      # normalised here only when ftfy is installed
            else:
                # round 3: non-ASCII documents too are left raw for the device (Engine.signatures_en)
                assert norm and raw[k] == contents[i] and hostio._ftfy is None
    # absent column / non-string column: nothing to hash
    assert len(hostio.read_parquet_shard(path, "missing", "doc_id")["cand"]) == 0
    assert len(hostio.read_parquet_shard(path, "n", "doc_id")["cand"]) == 0


def test_parquet_shard_reader_leaves_invalid_utf8_rows_unhashed(tmp_path):
    """rows whose bytes are not utf-8 (a lone surrogate's CESU bytes, stray 0xFF) cannot be a Python str in the
    reference (ZH-1:52-54 skips such lines): they are never hashed, hence always kept"""
    import pyarrow as pa
    import pyarrow.parquet as pq
    from map_neo_amd import hostio
    vals = [b"good one", b"bad \xff\xfe bytes", b"", b"caf\xc3\xa9", b"lone \xed\xa0\x80 surrogate"]
    offs = np.zeros(len(vals) + 1, np.int32)
    offs[1:] = np.cumsum([len(v) for v in vals])
    arr = pa.Array.from_buffers(pa.string(), len(vals), [None, pa.py_buffer(offs.tobytes()), pa.py_buffer(b"".join(vals))])
    path = str(tmp_path / "x.parquet")
    pq.write_table(pa.table({"doc_id": [f"d{i}" for i in range(5)], "raw_content": arr}), path)
    r = hostio.read_parquet_shard(path, "raw_content", "doc_id")
    assert r["cand"].tolist() == [0, 2, 3] and r["ids"] == [f"d{i}" for i in range(5)]
    assert bytes(r["text"]) == b"good onecaf\xc3\xa9" and r["offsets"].tolist() == [0, 8, 8, 13]


def test_native_reader_equals_python_reader_property(tmp_path):
    """hypothesis: arbitrary JSON rows (unicode incl. astral and control characters, nested values, duplicate-looking
    keys, both ensure_ascii settings, optional gzip) -- the native scanner + fallback must return exactly what the
    pure-Python reader returns, with and without EN normalisation"""
    import gzip
    import json
    from hypothesis import given, settings, strategies as st, HealthCheck
    from map_neo_amd import hostio
    if hostio._io() is None:
        pytest.skip("libmdd_io.so not built")
    text = st.text(alphabet=st.characters(blacklist_categories=("Cs",)), max_size=80)
    scalar = st.one_of(st.none(), st.booleans(), st.integers(-10**12, 10**12), st.floats(allow_nan=False, allow_infinity=False), text)
    value = st.recursive(scalar, lambda c: st.one_of(st.lists(c, max_size=3), st.dictionaries(text, c, max_size=3)), max_leaves=6)
    row = st.fixed_dictionaries({}, optional={"doc_id": st.one_of(text, st.integers(0, 10**9), st.none()),
                                              "raw_content": st.one_of(text, st.none(), st.integers()),
                                              "extra": value, "raw_content2": text})
    counter = [0]

    @settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
    @given(rows=st.lists(row, max_size=12), ascii_=st.booleans(), gz=st.booleans(), blank=st.booleans())
    def run(rows, ascii_, gz, blank):
        counter[0] += 1
        path = str(tmp_path / f"h{counter[0]}.jsonl{'.gz' if gz else ''}")
        lines = [json.dumps(r, ensure_ascii=ascii_) for r in rows]
        if blank:
            lines.insert(len(lines) // 2, "   ")
        data = ("\n".join(lines) + ("\n" if lines else "")).encode("utf8")
        with (gzip.open(path, "wb") if gz else open(path, "wb")) as f:
            f.write(data)
        for norm in (False, True):
            a = hostio.read_jsonl_shard(path, "raw_content", "doc_id", normalize_en=norm, use_native=True)
            b = hostio.read_jsonl_shard(path, "raw_content", "doc_id", normalize_en=norm, use_native=False)
            assert a["ids"] == b["ids"] and a["n_bad"] == b["n_bad"]
            assert a["line_no"].tolist() == b["line_no"].tolist() and a["cand"].tolist() == b["cand"].tolist()
            assert a["offsets"].tolist() == b["offsets"].tolist() and bytes(a["text"]) == bytes(b["text"])

    run()


def test_native_writer_equals_python_writer(tmp_path):
    """mdd_io_write_jsonl (egress of the exact-match stages) against json.dumps(row, ensure_ascii=False) line by line:
    escapes, control characters, non-BMP, empty documents, scalar and nested ids, keys that need escaping,
    part sub-ranges of a shared text buffer, plain and gzip output, invalid utf-8"""
    import gzip
    from map_neo_amd import hostio
    if hostio._io() is None:
        pytest.skip("libmdd_io.so not built")
    rng = np.random.default_rng(3)
    docs = ["", "plain", 'q"uote\\back', "ctl\x00\x01\x1f\x7f\n\r\t\b\f", "é漢字😀  ", "a" * 70000, "퟿"]
    for _ in range(500):
        cps = rng.integers(0, 0x3000, int(rng.integers(0, 200)))
        docs.append("".join(chr(int(c)) for c in cps))
    ids = ["id%d" % i for i in range(len(docs))]
    ids[3], ids[4], ids[5], ids[6], ids[7] = 'we"ird\nid', 17, None, 1.5, True
    enc = [d.encode() for d in docs]
    raw = np.frombuffer(b"".join(enc), dtype=np.uint8)
    offs = np.zeros(len(docs) + 1, np.uint64)
    offs[1:] = np.cumsum([len(e) for e in enc])
    a, b = str(tmp_path / "a"), str(tmp_path / "b")

    def both(idf, cf, ids_, offs_, level):
        hostio.write_jsonl_part(a, idf, cf, ids_, raw, offs_, level, native=True)
        hostio.write_jsonl_part(b, idf, cf, ids_, raw, offs_, level, native=False)
        op = gzip.open if level else open
        with op(a, "rb") as fa, op(b, "rb") as fb:
            A, B = fa.read(), fb.read()
        assert A == B
        return A

    for level in (6, 1, 0):
        A = both("doc_id", "raw_content", ids, offs, level)
        rows = [json.loads(l) for l in A.decode().split("\n") if l]
        assert [r["raw_content"] for r in rows] == docs and [r["doc_id"] for r in rows] == ids
    both('i"d', "c\\té", [[1, "a\nb"], {"k": "v"}] + ids[2:], offs, 6)          # nested ids, keys needing escapes
    both("doc_id", "raw_content", ids[5:9], offs[5:10], 6)                           # a part in the middle of the buffer
    assert both("doc_id", "raw_content", [], offs[:1], 6) == b""
    bad = np.frombuffer(b"ok\xff\xfe" + "퟿".encode() + b"\xed\xa0\x80", dtype=np.uint8)
    for o in ([0, 4], [4, 10]):                                                      # stray bytes; utf-8 encoded surrogate
        for nat in (False, True):
            with pytest.raises(UnicodeDecodeError):
                hostio.write_jsonl_part(a, "doc_id", "raw_content", ["x"], bad, np.array(o, np.uint64), 6, native=nat)
    assert not os.path.exists(a)                                                     # the native writer leaves no partial file
    with pytest.raises(OSError):
        hostio.write_jsonl_part(str(tmp_path / "no" / "dir"), "doc_id", "raw_content", ["x"], raw, offs[:2], 6)
