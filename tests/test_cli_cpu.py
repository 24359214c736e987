"""CPU suite: host logic of the drop-in commands -- flags match the reference's scripts,
text normalisation matches the oracle, ZH-4 / ZH-5 bookkeeping (no GPU involved)."""
import json
import os
import pickle

import numpy as np
import pytest

from map_neo_amd import hostio
from map_neo_amd.cli import (generate_connected_components, generate_dup_line_id_for_each_file, generate_dup_pairs,
                             generate_minhash, minhashLsh, remove_dup)
from oracle import ref_py as R


def _flags(parser):
    return {a.option_strings[0]: a.default for a in parser._actions if a.option_strings and a.option_strings[0] != "-h"}


def test_en_flags_match_reference():
    f = _flags(minhashLsh.build_parser())
    # EN:11-16
    for name in ("--input_dir", "--inter_dir", "--output_dir", "--content_field_name"):
        assert name in f
    assert f["--content_field_name"] == "raw_content"
    # reference constants EN:17-20 are the defaults of the extension flags
    assert (f["--width"], f["--bands"], f["--rows"], f["--num_perm"], f["--seed"]) == (13, 9, 13, 128, 1)


def test_zh_flags_match_reference():
    f = _flags(generate_minhash.build_parser())
    assert f["--workers"] == 42 and f["--content_field_name"] == "text"          # ZH-1:82-85
    assert {"--input_dir", "--output_dir"} <= set(f)
    assert {"--input_dir", "--output_dir"} <= set(_flags(generate_dup_pairs.build_parser()))               # ZH-2:53-54
    f3 = _flags(generate_connected_components.build_parser())
    assert {"--input_dir", "--output_file", "--num_workers"} <= set(f3) and f3["--num_workers"] == 90      # ZH-3:65-67
    assert {"--input_file", "--output_dir"} <= set(_flags(generate_dup_line_id_for_each_file.build_parser()))  # ZH-4:36-37
    assert {"--input_dir", "--output_dir", "--dup_line_id_dir", "--workers"} <= set(_flags(remove_dup.build_parser()))


@pytest.mark.parametrize("text", ["  Hello,   World!\n\tFoo-bar  ", "ＡＢＣ　ｄｅｆ", "école d'été — «test»", "", " \n ",
                                  "MiXeD CaSe: 123,456.78 (ok)? [yes]"])
def test_normalize_matches_oracle(text):
    assert hostio.normalize_text(text) == R.normalize_text(text)


def test_zh4_zh5_host_logic(tmp_path):
    """components -> per-file line-id sets -> filtered files; files without duplicates pass through"""
    inp, dup, out = tmp_path / "in", tmp_path / "dup", tmp_path / "out"
    inp.mkdir()
    (inp / "a.jsonl").write_bytes(b'{"text": "l0"}\n{"text": "l1"}\n{"text": "l2"}\n')
    (inp / "b.jsonl").write_bytes(b'{"text": "m0"}\n{"text": "m1"}\n')
    (inp / "c.jsonl").write_bytes(b'{"text": "n0"}\n')
    comps = [[0, 1, 3], [2, 4]]
    mapper = {0: "a.jsonl@0", 1: "a.jsonl@2", 2: "a.jsonl@1", 3: "b.jsonl@1", 4: "b.jsonl@0"}
    cc = tmp_path / "cc.pkl"
    with open(cc, "wb") as f:
        pickle.dump((comps, 2, mapper), f)
    n = generate_dup_line_id_for_each_file.generate_duplicates(
        generate_dup_line_id_for_each_file.build_parser().parse_args(["--input_file", str(cc), "--output_dir", str(dup)]))
    assert n == 3
    assert pickle.load(open(dup / "a.jsonl.pkl", "rb")) == {2}
    assert pickle.load(open(dup / "b.jsonl.pkl", "rb")) == {0, 1}
    assert not os.path.exists(dup / "c.jsonl.pkl")
    remove_dup.main(["--input_dir", str(inp), "--output_dir", str(out), "--dup_line_id_dir", str(dup), "--workers", "2"])
    assert (out / "a.jsonl").read_bytes() == b'{"text": "l0"}\n{"text": "l1"}\n'
    assert (out / "b.jsonl").read_bytes() == b""
    assert (out / "c.jsonl").read_bytes() == b'{"text": "n0"}\n'          # no .pkl: copied unchanged


def test_band_key_roundtrip():
    import numpy as np
    from map_neo_amd.cli import zh_common as Z
    sig = R.signatures(["the quick brown fox jumps over the lazy dog", "too short"], 128)
    for band in range(9):
        keys = Z.band_key_bytes(sig, band, 13)
        assert keys[0] == R.band_keys(sig[0].astype(np.uint64), 9, 13)[band]      # == _H(...) of ZH-1:32
        assert (Z.key_bytes_to_tuples(keys, 13) == sig[:, band * 13:(band + 1) * 13]).all()


def test_copy_without_lines_equals_line_loop(tmp_path):
    """ZH-5 line copy as runs between dropped lines == the reference's per-line loop, incl. empty files,
    a last line without newline, empty lines, indices past the end"""
    import io
    import random
    from map_neo_amd.cli import zh_common as Z
    rnd = random.Random(3)
    cases = [b"", b"\n", b"a", b"a\n", b"a\nb", b"a\nb\n", b"\n\n\n", b"a\n\nb\n\n", b"x" * 5 + b"\n" + b"y\n" * 7 + b"z",
             "".join(f'{{"text": "行{i}"}}\n' for i in range(40)).encode()]
    for data in cases:
        src = tmp_path / "src.jsonl"
        src.write_bytes(data)
        n = data.count(b"\n") + 3
        for _ in range(30):
            drop = {rnd.randrange(n) for _ in range(rnd.randrange(5))}
            exp = b"".join(line for i, line in enumerate(io.BytesIO(data)) if i not in drop)
            dst = tmp_path / "dst.jsonl"
            Z.copy_without_lines(str(src), str(dst), drop)
            assert dst.read_bytes() == exp, (data, drop)
            # the byte-range route of the fused command: line starts as the device-side index gives them (+ the file size)
            starts = [0] + [i + 1 for i, c in enumerate(data) if c == 10 and i + 1 < len(data)] + [len(data)] if data else [0]
            Z.copy_without_lines(str(src), str(dst), drop, np.asarray(starts, dtype=np.int64))
            assert dst.read_bytes() == exp, (data, drop, "line_starts")
    # more kept runs than one gathered write takes (IOV_MAX = 1024)
    data = b"".join(b"line %d\n" % i for i in range(5000))
    src.write_bytes(data)
    drop = set(range(0, 5000, 2))
    starts = np.concatenate(([0], np.flatnonzero(np.frombuffer(data, dtype=np.uint8) == 10) + 1))
    Z.copy_without_lines(str(src), str(dst), drop, starts)
    assert dst.read_bytes() == b"".join(b"line %d\n" % i for i in range(1, 5000, 2))


def test_shard_output_names_are_unique_and_keep_the_dotted_part():
    """`basename.split(".")[0]` mapped c4-train.00000-of-01024.json.gz and ...00001... to one output file"""
    names = hostio.shard_output_names(["/d/c4-train.00000-of-01024.json.gz", "/d/c4-train.00001-of-01024.json.gz", "/d/data.0.jsonl",
                                       "/d/data.1.jsonl", "/d/a.jsonl", "/d/a.jsonl.gz", "/e/a.parquet", "/d/plain", "/d/x.PQ"])
    assert names[:4] == ["c4-train.00000-of-01024.jsonl.gz", "c4-train.00001-of-01024.jsonl.gz", "data.0.jsonl.gz", "data.1.jsonl.gz"]
    assert names[4:7] == ["00004_a.jsonl.gz", "00005_a.jsonl.gz", "00006_a.jsonl.gz"] and names[7:] == ["plain.jsonl.gz", "x.jsonl.gz"]
    assert len(set(names)) == len(names)


def test_id_order_key_is_total():
    ids = ["b", 3, None, "a", 2.5, None, True, {"k": 1}, "10", 10]
    got = sorted(ids, key=hostio.id_order_key)
    assert got[:2] == [None, None] and got[2:6] == [True, 2.5, 3, 10] and got[6:9] == ["10", "a", "b"] and got[9] == {"k": 1}


def test_copy_without_lines_handles_gzip_shards(tmp_path):
    """line indices refer to the decompressed text; a gzip'd shard is inflated, filtered and written as gzip"""
    import gzip
    from map_neo_amd.cli import zh_common as Z
    lines = [("line %d" % i).encode() + b"\n" for i in range(50)]
    lines[-1] = lines[-1][:-1]                      # last line without a newline
    src_gz, src_plain = tmp_path / "s.jsonl.gz", tmp_path / "s.jsonl"
    src_gz.write_bytes(gzip.compress(b"".join(lines)))
    src_plain.write_bytes(b"".join(lines))
    drop = {0, 7, 8, 49, 100}
    Z.copy_without_lines(str(src_gz), str(tmp_path / "o.jsonl.gz"), drop)
    Z.copy_without_lines(str(src_plain), str(tmp_path / "o.jsonl"), drop)
    want = b"".join(l for i, l in enumerate(lines) if i not in drop)
    assert (tmp_path / "o.jsonl").read_bytes() == want
    assert gzip.decompress((tmp_path / "o.jsonl.gz").read_bytes()) == want
    Z.copy_without_lines(str(src_gz), str(tmp_path / "all.gz"), set())
    assert gzip.decompress((tmp_path / "all.gz").read_bytes()) == b"".join(lines)


def test_checkpoint_manifest_roundtrip_and_refusals(tmp_path):
    from map_neo_amd import checkpoint as C
    d = str(tmp_path / "ck")
    p = {"width": 13, "ngram_unit": "char", "num_perm": 4, "bands": 2, "rows": 2, "seed": 1}
    ids = ["a", "b", "c"]
    sig = np.arange(12, dtype=np.uint32).reshape(3, 4)
    C.save(d, p, ids, signatures=sig, extra={"n_rows": 5})
    man = C.save(d, p, ids, edges=np.array([[1, 0]], np.uint32), labels=np.array([0, 0, 2], np.uint32))
    assert sorted(man["artifacts"]) == ["doc_ids", "edges", "labels", "signatures"]            # earlier artifacts stay listed
    assert (C.load(d, "signatures", p)["signatures"] == sig).all() and C.load(d, "edges", p)["doc_ids"] == ids
    assert C.load(d, "labels", p)["labels"].tolist() == [0, 0, 2]
    with pytest.raises(C.CheckpointError, match="other parameters"):
        C.load(d, "labels", dict(p, bands=3))                   # labels depend on (bands, rows) ...
    assert C.load(d, "signatures", dict(p, bands=3))["doc_ids"] == ids          # ... signatures do not
    with pytest.raises(C.CheckpointError, match="other parameters"):
        C.load(d, "signatures", dict(p, width=5))
    with pytest.raises(C.CheckpointError, match="nothing to resume"):
        C.load(str(tmp_path / "none"), "signatures", p)
    os.remove(os.path.join(d, "edges.npy"))
    man = C.save(d, p, ids)
    assert "edges" not in man["artifacts"]                      # a listed artifact whose file is gone is dropped
    # a NEW run in a reused directory must never list stale edges / labels next to fresh signatures ...
    C.save(d, p, ids, edges=np.array([[1, 0]], np.uint32), labels=np.array([0, 0, 2], np.uint32))
    man = C.save(d, p, ids, signatures=sig + 1)
    assert sorted(man["artifacts"]) == ["doc_ids", "signatures"]
    with pytest.raises(C.CheckpointError, match="no 'labels'"):
        C.load(d, "labels", p)
    # ... other documents (same count) carry nothing over either ...
    C.save(d, p, ids, edges=np.array([[1, 0]], np.uint32))
    man = C.save(d, p, ["a", "b", "x"], labels=np.array([0, 1, 2], np.uint32))
    assert sorted(man["artifacts"]) == ["doc_ids", "labels"]
    # ... what was read (field, normalisation) is part of the parameters, and the input directory is checked
    class A:
        width, ngram_unit, num_perm, bands, rows, seed, content_field_name = 13, "char", 4, 2, 2, 1, "raw_content"
    pa = C.params_of(A)
    assert pa["normalize"] == "en" and pa["content_field"] == "raw_content" and pa["id_field"] == "doc_id"
    C.save(d, pa, ids, signatures=sig, extra={"input_dir": "/data/in"})
    assert C.load(d, "signatures", pa, "/data/in")["doc_ids"] == ids
    with pytest.raises(C.CheckpointError, match="written for input"):
        C.load(d, "signatures", pa, "/data/other")
    with pytest.raises(C.CheckpointError, match="other parameters"):
        C.load(d, "signatures", dict(pa, content_field="text"))
    with open(os.path.join(d, "doc_ids.json"), "w") as f:
        json.dump(["a", "b", "tampered"], f)
    with pytest.raises(C.CheckpointError, match="digest"):
        C.load(d, "signatures", pa)


def test_stage_cli_checkpoint_and_resume(tmp_path, mini_docs):
    """cli/dedup with the oracle-backed CPU backend: --checkpoint_dir, then --resume_from signatures | edges | labels --
    identical outputs every time (checkpoint / resume)"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from cpu_backend import OracleBackend
    from map_neo_amd.cli import dedup as stage
    n = len(mini_docs)
    inp = tmp_path / "in"
    inp.mkdir()
    with open(inp / "a.jsonl", "w", encoding="utf8") as fh:
        for i in range(n):
            fh.write(json.dumps({"doc_id": f"id{(i * 37) % n:04d}", "raw_content": mini_docs[i]}, ensure_ascii=False) + "\n")
    base = ["--input_dir", str(inp), "--normalize", "none", "--workers", "1", "--checkpoint_dir", str(tmp_path / "ck")]

    def run(out, extra):
        stage.run(stage.build_parser().parse_args(base + ["--output_dir", str(tmp_path / out)] + extra), engine_factory=lambda d: OracleBackend())
        return {f: open(tmp_path / out / f, encoding="utf8").read() for f in ("clusters.jsonl", "kept_ids.txt", "removed_ids.txt")}

    ref = run("o0", [])
    assert ref["removed_ids.txt"].strip()
    man = json.load(open(tmp_path / "ck" / "manifest.json"))
    assert sorted(man["artifacts"]) == ["doc_ids", "edges", "labels", "signatures"]
    for st in ("signatures", "edges", "labels"):
        assert run("o_" + st, ["--resume_from", st]) == ref, st
    with pytest.raises(Exception, match="other parameters"):
        run("o_bad", ["--resume_from", "labels", "--bands", "14", "--rows", "9"])
    # signatures can be reused for another (bands, rows): equals a fresh run with those parameters
    swept = run("o_sweep", ["--resume_from", "signatures", "--bands", "14", "--rows", "9"])
    fresh = stage.run(stage.build_parser().parse_args(["--input_dir", str(inp), "--normalize", "none", "--workers", "1", "--bands", "14", "--rows", "9",
                                                       "--output_dir", str(tmp_path / "o_fresh")]), engine_factory=lambda d: OracleBackend())
    assert swept["removed_ids.txt"] == open(tmp_path / "o_fresh" / "removed_ids.txt").read() and fresh["n_removed_ids"] > 0


def test_artifact_unpickler_refuses_code(tmp_path):
    """cli/zh_common.read_own_pickle loads plain containers only: a pickle that names any other
    global -- the way a pickle executes code -- is refused; the real artifact shapes load unchanged."""
    import pickle

    from map_neo_amd.cli import zh_common as Z

    class Evil:
        def __reduce__(self):
            import os
            return (os.system, ("echo pwned > %s" % (tmp_path / "pwned"),))

    p = tmp_path / "evil.pkl"
    p.write_bytes(pickle.dumps(Evil()))
    with pytest.raises(pickle.UnpicklingError):
        Z.read_own_pickle(str(p))
    assert not (tmp_path / "pwned").exists()
    arts = [[{"doc_id": "f.jsonl@3", "hash": b"\x00" * 104}], ([[0, 5], [1]], 2, {0: "f@0", 1: "f@1", 5: "g@2"}), {3, 4, 9}, set()]
    for proto in (2, 3, 4, 5):
        for a in arts:
            p.write_bytes(pickle.dumps(a, protocol=proto))
            assert Z.read_own_pickle(str(p)) == a


def test_text_mode_newlines_like_the_reference(tmp_path):
    """ZH-1:40-42 / ZH-5:9-13 open their files in text mode: "\\r\\n" and a lone "\\r" end a line, and ZH-5 writes every kept
    line back with "\\n".  Line numbering (ingest) and the filtered copy (egress) follow that on a mixed fixture."""
    from map_neo_amd import hostio
    from map_neo_amd.cli import zh_common as Z
    rows = ['{"text": "row %d %s"}' % (k, "x" * (k % 7)) for k in range(12)]
    ends = ["\n", "\r\n", "\r", "\r\n", "\n", "\r", "\r", "\n", "\r\n", "\n", "\r", ""]
    src = tmp_path / "mixed.jsonl"
    src.write_bytes("".join(r + e for r, e in zip(rows, ends)).encode())
    # the reference's enumeration
    with open(src, "r") as f:
        ref_lines = list(f)
    assert len(ref_lines) == 12 and all(ln.endswith("\n") for ln in ref_lines[:-1])
    line_no, text, offs = Z.read_jsonl_texts(str(src), "text")
    assert line_no.tolist() == list(range(12))
    got = [bytes(text[int(offs[k]):int(offs[k + 1])]).decode() for k in range(12)]
    assert got == [json.loads(r)["text"] for r in rows]
    # "\r\n"-only files keep the native reader and the same numbering
    crlf = tmp_path / "crlf.jsonl"
    crlf.write_bytes("\r\n".join(rows).encode())
    assert Z.read_jsonl_texts(str(crlf), "text")[0].tolist() == list(range(12))
    for path in (src, crlf):
        drop = {1, 2, 6, 11}
        with open(path, "r") as f, open(tmp_path / "ref.out", "w") as fw:       # ZH-5:9-13 restated
            for idx, line in enumerate(f):
                if idx not in drop:
                    fw.write(line)
        Z.copy_without_lines(str(path), str(tmp_path / "got.out"), drop)
        assert (tmp_path / "got.out").read_bytes() == (tmp_path / "ref.out").read_bytes()
        Z.copy_without_lines(str(path), str(tmp_path / "got2.out"), set())
        with open(path, "r") as f:
            assert (tmp_path / "got2.out").read_bytes() == f.read().encode()
