#!/usr/bin/env python3
"""End-to-end timing of the EN drop-in command on synthetic jsonl.gz shards (development
tool): writes N docs into S shards under a temp dir, runs map_neo_amd.cli.minhashLsh, prints
where the wall time goes.  The GPU part is a few % -- host JSON/gzip/normalisation dominate."""
import gzip
import json
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def zh(n, shards, workers):
    """The ZH path end to end on plain jsonl files (field `text`, doc identity file@line): the five reference
    commands ZH-1..ZH-5 through their artifacts on disk, then the fused command; both must write the same files."""
    import filecmp
    from map_neo_amd import corpus
    from map_neo_amd.cli import (generate_connected_components, generate_dup_line_id_for_each_file, generate_dup_pairs,
                                 generate_minhash, remove_dup, zh_dedup)
    spec = corpus.CorpusSpec(n, 2048, seed=9, lang="zh")
    text, offs = corpus.generate(spec)
    tb = text.numpy().tobytes()
    o = offs.numpy()
    root = tempfile.mkdtemp(prefix="mdd_e2e_zh_")
    inp = os.path.join(root, "in")
    os.makedirs(inp)
    per = (n + shards - 1) // shards
    for s in range(shards):
        with open(os.path.join(inp, f"part-{s:04d}.jsonl"), "w", encoding="utf8") as f:
            for i in range(s * per, min(n, (s + 1) * per)):
                f.write(json.dumps({"text": tb[o[i]:o[i + 1]].decode(), "url": "x"}, ensure_ascii=False) + "\n")
    M, P, C, D, O, O2 = (os.path.join(root, x) for x in ("minhash", "pairs", "cc.pkl", "lineids", "out", "out_fused"))
    steps = {}

    def timed(name, fn, argv):
        t0 = time.time()
        fn(argv)
        steps[name] = time.time() - t0

    t_all = time.time()
    timed("generate_minhash", generate_minhash.main, ["--input_dir", inp, "--output_dir", M, "--workers", str(workers)])
    timed("generate_dup_pairs", generate_dup_pairs.main, ["--input_dir", M, "--output_dir", P])
    timed("generate_connected_components", generate_connected_components.main, ["--input_dir", P, "--output_file", C])
    timed("generate_dup_line_id_for_each_file", generate_dup_line_id_for_each_file.main, ["--input_file", C, "--output_dir", D])
    timed("remove_dup", remove_dup.main, ["--input_dir", inp, "--output_dir", O, "--dup_line_id_dir", D, "--workers", str(workers)])
    wall5 = time.time() - t_all
    # the fused command with both ingest routes, each twice (the first run of a route pays for page-cache / allocator warm-up)
    fused = {}
    in_bytes = sum(os.path.getsize(os.path.join(inp, f)) for f in os.listdir(inp))
    names = sorted(os.listdir(inp))
    same = True
    for ingest in ("host", "device", "host", "device"):
        shutil.rmtree(O2, ignore_errors=True)
        t0 = time.time()
        zh_dedup.main(["--input_dir", inp, "--output_dir", O2, "--ingest", ingest, "--workers", str(workers)])
        w = time.time() - t0
        fused[ingest] = dict(zh_dedup.last_stats, wall_s=w, docs_per_s=n / w, jsonl_gb_per_s=in_bytes / w / 1e9,
                             ingest_hash_gb_per_s=in_bytes / zh_dedup.last_stats["t_read_extract_signatures_s"] / 1e9)
        same = same and all(filecmp.cmp(os.path.join(O, f), os.path.join(O2, f), shallow=False) for f in names)
    wall_fused = fused["device"]["wall_s"]
    kept = sum(1 for f in names for _ in open(os.path.join(O2, f), "rb"))
    print(json.dumps({"mode": "zh", "docs": n, "text_bytes": int(o[-1]), "jsonl_bytes": in_bytes, "shards": shards, "workers": workers,
                      "five_commands_wall_s": wall5, "five_commands_docs_per_s": n / wall5, "steps_s": steps,
                      "fused_wall_s": wall_fused, "fused_docs_per_s": n / wall_fused, "fused": fused,
                      "outputs_identical": same, "docs_kept": kept, "docs_removed": n - kept}))
    shutil.rmtree(root, ignore_errors=True)
    assert same, "five-step and fused ZH outputs differ"


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    shards = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    workers = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    mode = sys.argv[4] if len(sys.argv) > 4 else "minhash"      # "minhash": the EN drop-in; "pipeline": en_dedup_pipeline;
    #                                                               "zh": the five ZH commands, then the fused zh_dedup
    if mode == "zh":
        return zh(n, shards, workers)
    from map_neo_amd import corpus
    spec = corpus.CorpusSpec(n, 2048, seed=7, lang="en")
    text, offs = corpus.generate(spec)
    tb = text.numpy().tobytes()
    o = offs.numpy()
    root = tempfile.mkdtemp(prefix="mdd_e2e_")
    inp = os.path.join(root, "in")
    os.makedirs(inp)
    t0 = time.time()
    per = (n + shards - 1) // shards
    for s in range(shards):
        with gzip.open(os.path.join(inp, f"shard-{s:04d}.json.gz"), "wt", encoding="utf8", compresslevel=1) as f:
            for i in range(s * per, min(n, (s + 1) * per)):
                f.write(json.dumps({"doc_id": f"shard/{s:04d}/{i:09d}", "raw_content": tb[o[i]:o[i + 1]].decode(), "url": "x"}) + "\n")
    t_write = time.time() - t0
    if mode == "parquet":
        # the stage entry point on Parquet shards: Arrow buffers -> HBM, EN normalisation on the device
        import pyarrow as pa
        import pyarrow.parquet as pq
        from map_neo_amd.cli import dedup as stage
        pin = os.path.join(root, "pq")
        os.makedirs(pin)
        t0 = time.time()
        for s_ in range(shards):
            lo, hi = s_ * per, min(n, (s_ + 1) * per)
            pq.write_table(pa.table({"doc_id": [f"shard/{s_:04d}/{i:09d}" for i in range(lo, hi)],
                                     "raw_content": [tb[o[i]:o[i + 1]].decode() for i in range(lo, hi)]}),
                           os.path.join(pin, f"shard-{s_:04d}.parquet"))
        t_pq = time.time() - t0
        t0 = time.time()
        pargs = stage.build_parser().parse_args(["--input_dir", pin, "--output_dir", os.path.join(root, "out"),
                                                 "--normalize", "en", "--workers", str(workers)])
        if os.environ.get("MDD_E2E_PROFILE"):     # where the host's time goes: top of the cumulative profile on stderr
            import cProfile
            import pstats
            prof = cProfile.Profile()
            stats = prof.runcall(stage.run, pargs)
            pstats.Stats(prof, stream=sys.stderr).sort_stats("cumulative").print_stats(35)
        else:
            stats = stage.run(pargs)
        wall = time.time() - t0
        stats.update({"mode": mode, "docs": n, "text_bytes": int(o[-1]), "wall_s": wall, "docs_per_s": n / wall, "shards": shards,
                      "workers": workers, "make_parquet_s": t_pq})
        print(json.dumps(stats))
        shutil.rmtree(root, ignore_errors=True)
        return
    if mode == "pipeline":
        from map_neo_amd.cli import en_dedup_pipeline as P
        t0 = time.time()
        stats = P.run(P.build_parser().parse_args(["--input_dir", inp, "--output_dir", os.path.join(root, "out"),
                                                   "--workers", str(workers)]))
        wall = time.time() - t0
        stats.update({"mode": mode, "docs": n, "text_bytes": int(o[-1]), "wall_s": wall, "docs_per_s": n / wall, "shards": shards,
                      "workers": workers, "make_input_s": t_write})
        print(json.dumps(stats))
        shutil.rmtree(root, ignore_errors=True)
        return
    from map_neo_amd.cli import minhashLsh
    t0 = time.time()
    stats = minhashLsh.minhashLsh(minhashLsh.build_parser().parse_args(
        ["--input_dir", inp, "--inter_dir", os.path.join(root, "inter"), "--output_dir", os.path.join(root, "out"),
         "--workers", str(workers)]))
    wall = time.time() - t0
    stats.update({"docs": n, "text_bytes": int(o[-1]), "wall_s": wall, "docs_per_s": n / wall, "shards": shards,
                  "workers": workers, "make_input_s": t_write})
    print(json.dumps(stats))
    shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
