"""GPU suite: the command-line drop-ins end to end against the oracle."""
import gzip
import json
import os
import pickle

import numpy as np
import pytest

from map_neo_amd.cli import (generate_connected_components, generate_dup_line_id_for_each_file, generate_dup_pairs,
                             generate_minhash, minhashLsh, remove_dup, zh_dedup)
from oracle import ref_py as R

pytestmark = pytest.mark.gpu


def _en_rows(mini_docs):
    rng = np.random.Generator(np.random.PCG64(21))
    rows = []
    for i, d in enumerate(mini_docs):
        # RedPajama-style string ids whose order differs from file order
        rows.append({"doc_id": f"shard/{int(rng.integers(0, 5))}/{(i * 7919) % 1000:05d}", "raw_content": "  " + d.upper() + "!! ",
                     "meta": {"i": i}})
    rows.append({"doc_id": "shard/9/null", "raw_content": None, "meta": {"i": -1}})
    return rows


def test_en_cli_end_to_end(tmp_path, mini_docs, capsys):
    rows = _en_rows(mini_docs)
    inp = tmp_path / "in"
    inp.mkdir()
    with open(inp / "part-0.json", "w", encoding="utf8") as f:
        for r in rows[:120]:
            f.write(json.dumps(r, ensure_ascii=False) + "\n")
    with gzip.open(inp / "part-1.json.gz", "wt", encoding="utf8") as f:
        for r in rows[120:]:
            f.write(json.dumps(r, ensure_ascii=False) + "\n")
    minhashLsh.main(["--input_dir", str(inp), "--inter_dir", str(tmp_path / "inter"), "--output_dir", str(tmp_path / "out"),
                     "--workers", "2"])
    # oracle: EN:119-150 on the same rows
    cand = [r for r in rows if r["raw_content"] is not None]
    cand.sort(key=lambda r: r["doc_id"])
    texts = [R.normalize_text(r["raw_content"]) for r in cand]
    exp = R.dedup(texts)
    ids = [r["doc_id"] for r in cand]
    exp_removed = {ids[k] for k in np.flatnonzero(exp["removed"])}
    # literal star loop (EN:138-145) gives the same remove_list
    star = R.large_star_small_star(R.generate_edges(m)[i] for m in exp["buckets"].values() for i in range(len(m) - 1))
    assert {ids[u] for u, _ in star} == exp_removed
    out_rows = [json.loads(l) for part in sorted(os.listdir(tmp_path / "out"))
                for l in gzip.open(tmp_path / "out" / part, "rt", encoding="utf8")]
    assert {r["doc_id"] for r in out_rows} == {r["doc_id"] for r in rows} - exp_removed
    assert all("meta" in r for r in out_rows)                              # all original columns survive
    assert any(r["raw_content"] is None for r in out_rows)                 # null content is always kept
    assert f"the number of doc need to be remove is  {len(exp_removed)}" in capsys.readouterr().out
    sig = np.load(tmp_path / "inter" / "signatures.npy")
    inter_ids = json.load(open(tmp_path / "inter" / "doc_ids.json"))
    assert inter_ids == [r["doc_id"] for r in rows if r["raw_content"] is not None]     # input order
    row_of = {d: k for k, d in enumerate(ids)}
    assert (sig == exp["signatures"][[row_of[d] for d in inter_ids]]).all()
    # checkpoint manifest + resume: every artifact is named, and restarting from each gives identical outputs
    man = json.load(open(tmp_path / "inter" / "manifest.json"))
    assert sorted(man["artifacts"]) == ["doc_ids", "edges", "labels", "signatures"] and man["params"]["bands"] == 9

    def parts(d):
        return [gzip.open(tmp_path / d / p, "rb").read() for p in sorted(os.listdir(tmp_path / d))]

    ref_parts = parts("out")
    for st in ("signatures", "edges", "labels"):
        minhashLsh.main(["--input_dir", str(inp), "--inter_dir", str(tmp_path / "inter"), "--output_dir", str(tmp_path / ("out_" + st)),
                         "--workers", "2", "--resume_from", st])
        assert parts("out_" + st) == ref_parts, st
        assert f"the number of doc need to be remove is  {len(exp_removed)}" in capsys.readouterr().out
    from map_neo_amd.checkpoint import CheckpointError
    with pytest.raises(CheckpointError, match="other parameters"):
        minhashLsh.main(["--input_dir", str(inp), "--inter_dir", str(tmp_path / "inter"), "--output_dir", str(tmp_path / "bad"),
                         "--resume_from", "edges", "--bands", "14", "--rows", "9"])


def _zh_files(tmp_path, mini_docs):
    inp = tmp_path / "zh_in"
    inp.mkdir()
    files = {"f_a.jsonl": mini_docs[:70], "f_b.jsonl": mini_docs[70:150], "f_c.jsonl": mini_docs[150:]}
    for name, docs in files.items():
        with open(inp / name, "w", encoding="utf8") as f:
            for i, d in enumerate(docs):
                if name == "f_b.jsonl" and i == 5:
                    f.write("{broken json\n")                              # skipped line: never removed (ZH-1:52-54)
                f.write(json.dumps({"text": d, "k": i}, ensure_ascii=False) + "\n")
    return inp, files


def test_zh_five_steps_and_fused(tmp_path, mini_docs):
    inp, files = _zh_files(tmp_path, mini_docs)
    M, P, C, D, O = (tmp_path / x for x in ("mh", "pairs", "cc.pkl", "dupids", "out"))
    generate_minhash.main(["--input_dir", str(inp), "--output_dir", str(M), "--workers", "2"])
    # artifact format: <M>/<band>/<file>.gz = pickle list of {'doc_id','hash'} with 104-byte keys == oracle _H
    recs = pickle.load(open(M / "3" / "f_a.jsonl.gz", "rb"))
    assert recs[13]["doc_id"] == "f_a.jsonl@13" and len(recs[13]["hash"]) == 104
    assert recs[13]["hash"] == R.band_keys(R.signature(mini_docs[13]), 9, 13)[3]
    generate_dup_pairs.main(["--input_dir", str(M), "--output_dir", str(P)])
    assert sorted(os.listdir(P)) == [f"dup_paris_{b}.txt" for b in range(9)]
    generate_connected_components.main(["--input_dir", str(P), "--output_file", str(C)])
    generate_dup_line_id_for_each_file.main(["--input_file", str(C), "--output_dir", str(D)])
    remove_dup.main(["--input_dir", str(inp), "--output_dir", str(O), "--dup_line_id_dir", str(D), "--workers", "2"])
    # oracle: ids in (file, line) order; keep the smallest member of every component
    ids, texts = [], []
    for name in sorted(files):
        for idx, line in enumerate(open(inp / name, encoding="utf8")):
            try:
                texts.append(json.loads(line)["text"])
                ids.append((name, idx))
            except Exception:
                pass
    exp = R.dedup(texts, edge_shape="star")
    drop = {}
    for k in np.flatnonzero(exp["removed"]):
        drop.setdefault(ids[k][0], set()).add(ids[k][1])
    for name in files:
        want = b"".join(l for i, l in enumerate(open(inp / name, "rb")) if i not in drop.get(name, set()))
        assert (O / name).read_bytes() == want, name
        if name in drop:
            assert pickle.load(open(D / (name + ".pkl"), "rb")) == drop[name]
    comps, n_comp, mapper = pickle.load(open(C, "rb"))
    assert n_comp == len(comps) == len({int(l) for l in exp["labels"][exp["removed"]]})
    # fused command gives byte-identical outputs
    O2 = tmp_path / "out_fused"
    zh_dedup.main(["--input_dir", str(inp), "--output_dir", str(O2), "--report", str(tmp_path / "rep.json")])
    for name in files:
        assert (O2 / name).read_bytes() == (O / name).read_bytes()
    rep = json.load(open(tmp_path / "rep.json"))
    assert len(rep["removed"]) == int(exp["removed"].sum())


def test_en_cli_duplicate_doc_ids(tmp_path, mini_docs, capsys):
    """rows sharing a doc_id are one node of the reference's graph (edges carry doc_ids, EN:136-145;
    the anti-join EN:149 drops every row of a removed id)"""
    base = mini_docs[60]
    rows = [{"doc_id": "b", "raw_content": base},
            {"doc_id": "a", "raw_content": mini_docs[61]},      # unrelated text, but shares its id with a copy of base
            {"doc_id": "a", "raw_content": base},
            {"doc_id": "c", "raw_content": base},
            {"doc_id": "d", "raw_content": mini_docs[62]}]
    inp = tmp_path / "in"
    inp.mkdir()
    with open(inp / "x.json", "w", encoding="utf8") as f:
        for r in rows:
            f.write(json.dumps(r) + "\n")
    minhashLsh.main(["--input_dir", str(inp), "--inter_dir", str(tmp_path / "m"), "--output_dir", str(tmp_path / "o"),
                     "--workers", "1"])
    out_rows = [json.loads(l) for part in sorted(os.listdir(tmp_path / "o"))
                for l in gzip.open(tmp_path / "o" / part, "rt", encoding="utf8")]
    # component {a, b, c} keeps the smallest id "a" (both of its rows); "d" is untouched
    assert sorted(r["doc_id"] for r in out_rows) == ["a", "a", "d"]
    assert "the number of doc need to be remove is  2" in capsys.readouterr().out


def test_stage_entry_point_jsonl_and_parquet(tmp_path, mini_docs):
    """north_star I/O contract: jsonl / parquet shards in, kept ids + cluster assignments out"""
    import pyarrow as pa
    import pyarrow.parquet as pq
    from map_neo_amd.cli import dedup as stage
    n = len(mini_docs)
    ids = [f"id{(i * 37) % n:04d}" for i in range(n)]
    inp = tmp_path / "in"
    inp.mkdir()
    with open(inp / "a.jsonl", "w", encoding="utf8") as f:
        for i in range(0, 60):
            f.write(json.dumps({"doc_id": ids[i], "raw_content": mini_docs[i]}, ensure_ascii=False) + "\n")
    with gzip.open(inp / "b.jsonl.gz", "wt", encoding="utf8") as f:
        for i in range(60, 120):
            f.write(json.dumps({"doc_id": ids[i], "raw_content": mini_docs[i]}, ensure_ascii=False) + "\n")
    pq.write_table(pa.table({"doc_id": ids[120:], "raw_content": mini_docs[120:], "extra": list(range(120, n))}),
                   inp / "c.parquet")
    out = tmp_path / "out"
    s = stage.run(stage.build_parser().parse_args(["--input_dir", str(inp), "--output_dir", str(out), "--normalize", "none",
                                                   "--write_shards", "--workers", "2"]))
    exp = R.dedup(mini_docs)
    lab = exp["labels"]
    comp = {}
    for k in range(n):
        comp.setdefault(int(lab[k]), []).append(k)
    exp_removed, exp_cluster = set(), {}
    for members in comp.values():
        if len(members) > 1:
            keep = min(ids[k] for k in members)
            for k in members:
                exp_cluster[ids[k]] = keep
                if ids[k] != keep:
                    exp_removed.add(ids[k])
    assert set(open(out / "removed_ids.txt").read().split()) == exp_removed
    assert set(open(out / "kept_ids.txt").read().split()) == set(ids) - exp_removed
    got_cluster = {r["doc_id"]: r["cluster"] for r in map(json.loads, open(out / "clusters.jsonl", encoding="utf8"))}
    assert got_cluster == exp_cluster
    assert s["n_removed_ids"] == len(exp_removed) and s["kept_rows_written"] == n - len(exp_removed)
    rows = [json.loads(l) for p in sorted(os.listdir(out / "shards")) for l in gzip.open(out / "shards" / p, "rt", encoding="utf8")]
    assert {r["doc_id"] for r in rows} == set(ids) - exp_removed
    assert any("extra" in r for r in rows)


def test_stage_entry_point_parquet_device_normalisation(tmp_path, mini_docs):
    """Parquet + --normalize en: ASCII documents are normalised on the device, the others by the
    Python steps in the pool; result == oracle dedup of the Python-normalised texts"""
    import pyarrow as pa
    import pyarrow.parquet as pq
    from map_neo_amd import hostio
    from map_neo_amd.cli import dedup as stage
    rng = np.random.Generator(np.random.PCG64(21))
    docs = []
    for i, t in enumerate(mini_docs):
        if t.isascii() and i % 2:
            t = "  " + t.upper().replace(" ", " ,\t ", 3) + "!! "          # same document after EN:31-40
        docs.append(t)
    docs[5] = None
    n = len(docs)
    ids = [f"id{(i * 41) % n:04d}" for i in range(n)]
    inp = tmp_path / "in"
    inp.mkdir()
    pq.write_table(pa.table({"doc_id": ids[:100], "raw_content": docs[:100]}), inp / "a.parquet", row_group_size=17)
    pq.write_table(pa.table({"doc_id": ids[100:], "raw_content": docs[100:]}), inp / "b.parquet")
    out = tmp_path / "out"
    s = stage.run(stage.build_parser().parse_args(["--input_dir", str(inp), "--output_dir", str(out), "--normalize", "en",
                                                   "--workers", "2"]))
    hashed = [k for k in range(n) if docs[k] is not None]
    lab = R.dedup([R.normalize_text(docs[k]) for k in hashed])["labels"]
    comp = {}
    for j, k in enumerate(hashed):
        comp.setdefault(int(lab[j]), []).append(k)
    exp_removed = set()
    for members in comp.values():
        keep = min(ids[k] for k in members)
        exp_removed |= {ids[k] for k in members if ids[k] != keep}
    assert set(open(out / "removed_ids.txt").read().split()) == exp_removed
    assert ids[5] in set(open(out / "kept_ids.txt").read().split())
    assert s["n_hashed"] == n - 1 and len(exp_removed) > 0


def _write_jsonl(path, rows, gz=False):
    op = gzip.open if gz else open
    with op(path, "wt", encoding="utf8") as f:
        for r in rows:
            f.write(json.dumps(r, ensure_ascii=False) + "\n")


def _read_parts(d):
    return [json.loads(l) for p in sorted(os.listdir(d)) for l in gzip.open(os.path.join(d, p), "rt", encoding="utf8")]


def _en_stage_corpus():
    rng = np.random.Generator(np.random.PCG64(31))
    words = ["alpha", "beta", "gamma", "delta", "epsilon", "zeta", "naïve", "数据", "x"]
    paras = [" ".join(rng.choice(words, int(rng.integers(3, 30)))) for _ in range(40)] + [""]
    docs = ["\n".join(rng.choice(paras, int(rng.integers(1, 8)))) for _ in range(150)]
    docs += [docs[int(i)] for i in rng.integers(0, 150, 30)]
    ids = [f"d{int(i):04d}" for i in rng.permutation(len(docs))]
    return [{"doc_id": i, "raw_content": t, "meta": k} for k, (i, t) in enumerate(zip(ids, docs))]


def test_exact_document_deduplication_cli(tmp_path):
    from oracle import ref_exact as RX
    from map_neo_amd.cli import exact_document_deduplication as cli
    rows = _en_stage_corpus()
    rows[3]["raw_content"] = None
    rows[100].pop("raw_content")
    inp = tmp_path / "in"
    inp.mkdir()
    _write_jsonl(inp / "a.json", rows[:90])
    _write_jsonl(inp / "b.json.gz", rows[90:], gz=True)
    cli.main(["--input_dir", str(inp), "--output_dir", str(tmp_path / "o"), "--workers", "2"])
    keep = RX.exact_document_dedup([r.get("raw_content") for r in rows])
    assert _read_parts(tmp_path / "o") == [r for r, k in zip(rows, keep) if k]


def test_paragraph_cli(tmp_path, capsys):
    from oracle import ref_exact as RX
    from map_neo_amd.cli import paragraph as cli
    rows = _en_stage_corpus()
    rows[5]["raw_content"] = None
    inp = tmp_path / "in"
    inp.mkdir()
    _write_jsonl(inp / "a.json", rows[:70])
    _write_jsonl(inp / "b.json", rows[70:])
    cli.main(["--input_dir", str(inp), "--output_dir", str(tmp_path / "o"), "--workers", "2"])
    exp = RX.paragraph_dedup([(r["doc_id"], r["raw_content"]) for r in rows], min_len=20)
    assert _read_parts(tmp_path / "o") == [{"doc_id": i, "raw_content": t} for i, t in exp]
    assert "after paragraph dedup: %d" % len(exp) in capsys.readouterr().out


def test_exact_substr_dedup_cli(tmp_path, capsys):
    from oracle import ref_exact as RX
    from map_neo_amd.cli import exact_substr_dedup as cli
    rows = _en_stage_corpus()
    inp = tmp_path / "in"
    inp.mkdir()
    _write_jsonl(inp / "a.json", rows[:70])
    _write_jsonl(inp / "b.json", rows[70:])
    cli.main(["--input_dir", str(inp), "--output_dir", str(tmp_path / "o"), "--workers", "1"])
    exp = RX.exact_substr_dedup([(r["raw_content"], r["doc_id"]) for r in rows], 50, 20)
    assert _read_parts(tmp_path / "o") == [{"doc_id": i, "raw_content": t} for i, t in exp]
    out = capsys.readouterr().out
    assert "the number of docs is  %d" % len(rows) in out and "the number of deduped docs is  %d" % len(exp) in out


def test_stage_entry_point_distributed_one_rank(tmp_path, mini_docs):
    """the multi-GPU code path of the stage CLI (RCCL all-to-all / all-gather) with a single rank == 1-GPU result"""
    import subprocess
    import sys
    n = len(mini_docs)
    ids = [f"id{(i * 37) % n:04d}" for i in range(n)]
    inp = tmp_path / "in"
    inp.mkdir()
    for f, (a, b) in enumerate([(0, 70), (70, 140), (140, n)]):
        _write_jsonl(inp / f"s{f}.jsonl", [{"doc_id": ids[i], "raw_content": mini_docs[i]} for i in range(a, b)])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for mode, extra in (("single", []), ("dist", ["--distributed"])):
        out = tmp_path / mode
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 300), PYTHONPATH=root)
        r = subprocess.run([sys.executable, "-m", "map_neo_amd.cli.dedup", "--input_dir", str(inp), "--output_dir", str(out),
                            "--normalize", "none", "--workers", "2"] + extra, capture_output=True, text=True, timeout=120,
                           env=env, cwd=root)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        outs[mode] = {name: open(out / name, encoding="utf8").read() for name in ("kept_ids.txt", "removed_ids.txt", "clusters.jsonl")}
        assert sorted(os.listdir(out)) == ["clusters.jsonl", "kept_ids.txt", "removed_ids.txt", "summary.json"]
    assert outs["single"] == outs["dist"]
    lab = R.dedup(mini_docs)["labels"]
    assert len(outs["dist"]["removed_ids.txt"].split()) == int((lab != np.arange(n)).sum())


def test_en_pipeline_equals_the_four_commands_chained(tmp_path):
    """one process with the corpus kept in HBM == exact_document_deduplication -> minhashLsh -> paragraph ->
    exact_substr_dedup chained through gzip JSON files"""
    from map_neo_amd.cli import en_dedup_pipeline, exact_document_deduplication, exact_substr_dedup, paragraph
    rng = np.random.Generator(np.random.PCG64(41))
    words = ["Alpha", "beta,", "GAMMA", "delta.", "epsilon", "zeta!", "naïve", "数据", "eta", "theta", "iota", "kappa"]
    paras = [" ".join(rng.choice(words, int(rng.integers(8, 40)))) for _ in range(60)]
    docs = ["\n".join(rng.choice(paras, int(rng.integers(2, 9)))) for _ in range(200)]
    docs += [docs[int(i)] for i in rng.integers(0, 200, 40)]                                  # exact copies
    docs += [docs[int(i)].replace("beta,", "BETA", 1) + " x" for i in rng.integers(0, 200, 40)]  # near copies
    ids = [f"d{int(i):04d}" for i in rng.permutation(len(docs))]
    rows = [{"doc_id": i, "raw_content": t, "url": f"http://x/{k}"} for k, (i, t) in enumerate(zip(ids, docs))]
    rows[7]["raw_content"] = None
    inp = tmp_path / "in"
    inp.mkdir()
    _write_jsonl(inp / "a.json", rows[:150])
    _write_jsonl(inp / "b.json.gz", rows[150:], gz=True)
    w = ["--workers", "2"]
    exact_document_deduplication.main(["--input_dir", str(inp), "--output_dir", str(tmp_path / "s1")] + w)
    minhashLsh.main(["--input_dir", str(tmp_path / "s1"), "--inter_dir", str(tmp_path / "m"), "--output_dir", str(tmp_path / "s2")] + w)
    paragraph.main(["--input_dir", str(tmp_path / "s2"), "--output_dir", str(tmp_path / "s3")] + w)
    exact_substr_dedup.main(["--input_dir", str(tmp_path / "s3"), "--output_dir", str(tmp_path / "s4")] + w)
    chained = _read_parts(tmp_path / "s4")
    st = en_dedup_pipeline.run(en_dedup_pipeline.build_parser().parse_args(["--input_dir", str(inp), "--output_dir", str(tmp_path / "p")] + w))
    fused = _read_parts(tmp_path / "p")
    assert fused == chained and 0 < len(fused) < len(rows)
    assert st["after_exact_document"] < st["rows_with_content"] and st["after_minhash"] < st["after_exact_document"]
    assert st["after_exact_substring"] == len(fused)


def test_full_text_dedup_cli(tmp_path):
    """ZH full-text dedup: first line of every distinct `text` survives byte for byte, parts of --batch_size lines"""
    from map_neo_amd.cli import full_text_dedup as cli
    rng = np.random.Generator(np.random.PCG64(51))
    texts = ["文档%d 内容 %s" % (i, "甲乙丙丁"[i % 4] * (i % 7)) for i in range(120)]
    rows = [{"text": texts[int(i)], "meta": int(k)} for k, i in enumerate(rng.integers(0, 120, 400))]
    inp = tmp_path / "in"
    inp.mkdir()
    lines = [json.dumps(r, ensure_ascii=bool(k % 2)) + "\n" for k, r in enumerate(rows)]
    lines[17] = "not json\n"
    lines[33] = json.dumps({"other": 1}) + "\n"
    with open(inp / "b.jsonl", "w", encoding="utf8") as f:
        f.writelines(lines[250:])
    with open(inp / "a.jsonl", "w", encoding="utf8") as f:
        f.writelines(lines[:250])
    st = cli.run(cli.build_parser().parse_args(["--input_dir", str(inp), "--output_dir", str(tmp_path / "o"), "--batch_size", "50"]))
    seen, exp = set(), []
    for k, line in enumerate(lines):                       # a.jsonl (lines[:250]) sorts before b.jsonl
        if k in (17, 33):
            continue
        t = rows[k]["text"]
        if t not in seen:
            seen.add(t)
            exp.append(line)
    parts = sorted(os.listdir(tmp_path / "o"), key=lambda n: int(n.split("_")[1].split(".")[0]))
    got = [l for p in parts for l in open(tmp_path / "o" / p, encoding="utf8")]
    assert got == exp and st["n_kept"] == len(exp) and st["n_without_content"] == 1
    assert parts[:-1] == [f"part_{i}.jsonl" for i in range(len(parts) - 1)] and parts[-1] == f"part_{len(parts) - 1}"


def test_rccl_exchange_larger_than_one_gib(tmp_path):
    """RCCL's all_to_all_single drops the tail of a message beyond 1 GiB (seen with one rank); the exchange helper
    must move 1.3 GiB intact by running in rounds"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "a2a.py"
    script.write_text(f'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from map_neo_amd.dedup import _all_to_all_v, _all_gather_v
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
n = 27_000_000                      # records of 13 x int32 = 1.31 GiB
x = torch.arange(n * 13, dtype=torch.int32, device="cuda")
y = _all_to_all_v(x, [n], [n], 13)
assert torch.equal(x, y)
g = _all_gather_v(x[: 100_000_000], limit_bytes=64 << 20)
assert torch.equal(g, x[: 100_000_000])
dist.barrier(); dist.destroy_process_group()
print("ok")
''')
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + os.getpid() % 200), RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]


def test_native_rccl_comm_one_rank(tmp_path):
    """the product transport of the multi-GPU path -- collectives issued by libmdd_hip.so against RCCL (mdd_comm_init,
    mdd_exchange_counts / _bands, mdd_allgather_counts / _edges) -- with the one rank a one-GPU box allows: same labels
    and removed set as the single-GPU path, the exchange statistics add up, and a second communicator is refused"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "native.py"
    script.write_text(f'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from map_neo_amd import corpus, lib
from map_neo_amd.dedup import NativeComm, dedup_distributed, dedup_local, default_comm
from map_neo_amd.minhash import Engine
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
eng = Engine(0)
spec = corpus.CorpusSpec(40000, 600, lang="mixed", seed=31, dup_frac=0.2, edit_rate=0.005)
text, offs = corpus.generate(spec, device=eng.device)
ref = dedup_local(eng, text, offs)
for K, bands, rows in ((128, 9, 13), (256, 20, 12)):
    ref = dedup_local(eng, text, offs, K, 13, "char", bands, rows)
    tm = {{}}
    out = dedup_distributed(eng, text, offs, 0, spec.n_docs, K, 13, "char", bands, rows, timings=tm)
    assert isinstance(default_comm(eng), NativeComm) and tm["transport"] == "libmdd_hip + RCCL", tm
    assert torch.equal(out["labels"], ref["labels"]) and torch.equal(out["removed"], ref["removed"])
    assert tm["records_sent"] == tm["records_received"] == spec.n_docs * bands and tm["records_sent_to_other_ranks"] == 0
    assert tm["all_to_all_bytes_sent"] == spec.n_docs * bands * (4 * rows + 8)
    assert tm["edges_global"] == tm["forest_edges_local"] <= tm["edges_local"]
try:
    eng.comm_init(eng.comm_unique_id(), 0, 1)
    raise SystemExit("a second communicator was accepted")
except lib.MddError as e:
    assert "already has a communicator" in str(e)
# an engine without a communicator refuses the exchange calls
e2 = Engine(0)
try:
    e2.comm_world = 1
    e2.allgather_edges(out["edge_u"], out["edge_v"])
    raise SystemExit("exchange without mdd_comm_init was accepted")
except lib.MddError as e:
    assert "mdd_comm_init first" in str(e)
dist.barrier(); dist.destroy_process_group()
print("ok")
''')
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 100), RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=150, env=env)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]


DIST_REHEARSAL = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from map_neo_amd import corpus
from map_neo_amd.dedup import dedup_distributed, dedup_local
from map_neo_amd.minhash import Engine
dist.init_process_group("gloo")                    # gloo moves CUDA tensors too: all ranks share GPU 0, the kernels are the real ones
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
eng = Engine(0)
n = 60000
spec = corpus.CorpusSpec(n, 600, lang="mixed", seed=31, dup_frac=0.2, edit_rate=0.005)
bounds = [0, 9000, 9000, 31000, n] if world == 4 else [(n * r) // world for r in range(world + 1)]    # uneven, one empty rank
lo, hi = bounds[rank], bounds[rank + 1]
text, offs = corpus.generate(spec, lo, hi, eng.device) if hi > lo else (torch.zeros(256, dtype=torch.uint8, device=eng.device), torch.zeros(1, dtype=torch.int64, device=eng.device))
for K, bands, rows in ((128, 9, 13), (256, 20, 12)):
    out = dedup_distributed(eng, text, offs, lo, n, K, 13, "char", bands, rows)
    labels = out["labels"]
    removed_local = int(out["n_removed"].item())
    tot = torch.tensor([removed_local], dtype=torch.int64)
    dist.all_reduce(tot)
    if rank == 0:
        ft, fo = corpus.generate(spec, 0, n, eng.device)
        ref = dedup_local(eng, ft, fo, K, 13, "char", bands, rows)
        assert torch.equal(labels, ref["labels"]), (K, bands)
        assert int(tot.item()) == int(ref["n_removed"].item()) > 1000, (int(tot.item()), int(ref["n_removed"].item()))
    # every rank holds the same labels
    chk = labels.to(torch.int64).sum().reshape(1).cpu()
    allc = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(allc, chk)
    assert len({{int(c.item()) for c in allc}}) == 1
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


@pytest.mark.parametrize("world", [2, 4, 8])
def test_multi_rank_rehearsal_with_real_kernels(tmp_path, world):
    """dedup_distributed with the real HIP engine on several ranks (gloo carries the CUDA tensors; all ranks share
    GPU 0): uneven shards incl. an empty rank, two (K, bands, rows) settings; labels == the single-GPU path"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rehearsal.py"
    script.write_text(DIST_REHEARSAL.format(root=root))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29800 + os.getpid() % 100 + world), str(script)]
    # gloo picks its interface from the host name, which does not always resolve on the GPU boxes: pin it to loopback
    env = dict(os.environ, OMP_NUM_THREADS="2", GLOO_SOCKET_IFNAME="lo")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0 and r.stdout.count("ok") >= world, r.stdout[-1500:] + r.stderr[-4000:]


def test_bench_script_multi_rank_rehearsal():
    """bench.py exactly as the driver launches it for N > 1 (torchrun, one rank per process), rehearsed on one GPU over
    gloo: ONE JSON line from rank 0, whole-job figures, and the same removed count as the single-GPU run of the same
    global corpus"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MDD_BENCH_BACKEND="gloo", GLOO_SOCKET_IFNAME="lo", OMP_NUM_THREADS="2")
    port = str(29850 + os.getpid() % 100)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
                        "--master-port", port, "bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1", "--docs", "50000",
                        "--corpus-config", "4"],          # the corpus settings the driver's N > 1 runs get by default
                       capture_output=True, text=True, timeout=120, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    multi = json.loads(lines[0])
    assert multi["n_gpus"] == 4 and multi["config"]["global_docs"] == 200000 and multi["scaling"] == "weak"
    assert multi["vs_baseline"] is None and "cpu_baseline" not in multi
    ex = multi["exchange"]
    assert ex["transport"] == "torch.distributed" and ex["records_sent"] == 50000 * 9 and ex["all_to_all_ms"] > 0 and ex["all_gather_ms"] > 0
    assert ex["all_to_all_bytes_sent"] == 50000 * 9 * 60 and multi["config"]["workload"].startswith("C4 share")
    r1 = subprocess.run([sys.executable, "bench.py", "--docs", "200000", "--no-cpu", "--no-pcie", "--no-stages", "--steps", "1", "--warmup", "0",
                         "--corpus-config", "4"],
                        capture_output=True, text=True, timeout=120, env=dict(os.environ), cwd=root)
    assert r1.returncode == 0, r1.stderr[-3000:]
    single = json.loads(r1.stdout.strip().splitlines()[-1])
    # (LSH at ~0.8 Jaccard catches the lightly edited copies of C4's edit-rate sweep, not the 10-20 % ones)
    assert multi["n_removed"] == single["n_removed"] > 0.2 * single["n_injected_duplicates"]


def test_stage_cli_multi_rank_rehearsal(tmp_path, mini_docs):
    """`torchrun -m map_neo_amd.cli.dedup` with 3 ranks and the real engine (gloo, all ranks on GPU 0): the stitched
    outputs equal the single-process run's"""
    import subprocess
    import sys
    n = len(mini_docs)
    ids = [f"id{(i * 37) % n:04d}" for i in range(n)]
    inp = tmp_path / "in"
    inp.mkdir()
    for f, (a, b) in enumerate([(0, 30), (30, 90), (90, 140), (140, 170), (170, n)]):
        _write_jsonl(inp / f"s{f}.jsonl", [{"doc_id": ids[i], "raw_content": mini_docs[i]} for i in range(a, b)])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = ["--input_dir", str(inp), "--normalize", "none", "--workers", "3", "--write_shards"]
    env = dict(os.environ, PYTHONPATH=root, MDD_DIST_BACKEND="gloo", GLOO_SOCKET_IFNAME="lo", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
                        "--master-port", str(29950 + os.getpid() % 40), "-m", "map_neo_amd.cli.dedup", "--output_dir", str(tmp_path / "multi")] + base,
                       capture_output=True, text=True, timeout=120, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    r1 = subprocess.run([sys.executable, "-m", "map_neo_amd.cli.dedup", "--output_dir", str(tmp_path / "single")] + base,
                        capture_output=True, text=True, timeout=120, env=dict(os.environ, PYTHONPATH=root), cwd=root)
    assert r1.returncode == 0, r1.stderr[-3000:]
    for name in ("kept_ids.txt", "removed_ids.txt", "clusters.jsonl"):
        assert open(tmp_path / "multi" / name).read() == open(tmp_path / "single" / name).read(), name
    assert sorted(os.listdir(tmp_path / "multi" / "shards")) == sorted(os.listdir(tmp_path / "single" / "shards"))
    for p in os.listdir(tmp_path / "single" / "shards"):
        assert gzip.open(tmp_path / "multi" / "shards" / p).read() == gzip.open(tmp_path / "single" / "shards" / p).read()
    assert json.load(open(tmp_path / "multi" / "summary.json"))["world"] == 3
