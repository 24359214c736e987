#!/usr/bin/env python3
"""bench.py -- MinHash-LSH near-dedup throughput on MI355X (BASELINE.json metric).

A "step" is one pass of the whole hot path (signatures -> band records -> buckets/edges ->
union-find -> removed mask) over a synthetic corpus already resident in HBM (`value`); the same
pass from packed text in PINNED HOST memory to the removed mask on the host (SURVEY.md 8(d)'s
region) is timed next to it and reported as `pinned_host` (never `value`: task contract).
Workload at N=1: BASELINE.json configs[1] ("C2"): 1M synthetic English docs (~2 KB avg),
char 13-grams, num_perm=128, 9 bands x 13 rows.  For N>1 every rank holds one GPU's share of
BASELINE.json configs[3] ("C4": 100M docs on 8 GPUs = 12.5M docs per GPU, weak scaling) and the
band records are exchanged with an RCCL all-to-all.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (sig_kernel, VALU-bound
integer work: SURVEY.md 8(d)); `cpu_baseline` is the oracle port timed on this host's cores
(rank 0, N=1 only; bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WIDTH, K, BANDS, ROWS = 13, 128, 9, 13
OPS_PER_SHINGLE = lambda k: 958 + 14 * k          # SURVEY.md 8(d) algorithmic int32 ops per shingle
PEAK_VALU_TOPS = 78.6                             # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz: one simple int32 op per lane per issue slot
#                                                   (= half of the guide's 157.3 TFLOPS FP32 vector peak, which counts FMA as 2)


def _cpu_worker(args):
    """One shard of the CPU baseline: the reference's per-document loop (ZH-1:23-33 / EN:56-64),
    restated in oracle/ref_py.py -- MinHash object (and permutation table) rebuilt per doc."""
    docs, = args
    from oracle import ref_py as R
    out = []
    for t in docs:
        m = R.MinHash(num_perm=K)                 # regenerates the table, as MinHash(num_perm=128) does
        for f in R.shingles_char(t, WIDTH):
            m.update(f)
        out.append(m.hashvalues)
    return out


def cpu_baseline(n_sample: int, mean_bytes: int, seed: int):
    """Python port (numpy + hashlib), multiprocessing.Pool over host cores like ZH-1:68-78,
    on the first n_sample documents of the same synthetic corpus; bucketing + CC included."""
    import multiprocessing as mp

    import numpy as np

    from map_neo_amd import corpus
    from oracle import ref_c as RC
    from oracle import ref_py as R
    cores = min(16, os.cpu_count() or 1)          # the GPU box gives one GPU a 16-CPU share
    spec = corpus.CorpusSpec(n_sample, mean_bytes, seed=seed, lang="en")
    text, offs = corpus.generate(spec)
    tb = text.numpy().tobytes()
    o = offs.numpy()
    docs = [tb[o[i]:o[i + 1]].decode("utf8") for i in range(n_sample)]
    shards = [(docs[i::cores],) for i in range(cores)]
    t0 = time.time()
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(_cpu_worker, shards)
    sig = np.empty((n_sample, K), np.uint32)
    for i, r in enumerate(res):
        sig[i::cores] = np.asarray(r, dtype=np.uint64).astype(np.uint32).reshape(-1, K)
    bk = R.buckets(sig, BANDS, ROWS)              # dict bucketing like ZH-2:19-40
    edges = set()
    for members in bk.values():
        edges.update(R.star_edges(members))
    labels = R.components_min_label(n_sample, edges)
    dt = time.time() - t0
    # optimised single-core C restatement, same sample
    t1 = time.time()
    sig_c = RC.signatures(text.numpy(), o.astype(np.uint64), K, WIDTH)
    RC.cluster(sig_c, BANDS, ROWS)
    dt_c = time.time() - t1
    assert (sig_c == sig).all()
    # the same C restatement with the documents spread over the host cores (OpenMP): the "optimised CPU" line
    t2 = time.time()
    sig_mt = RC.signatures(text.numpy(), o.astype(np.uint64), K, WIDTH, threads=cores)
    RC.cluster(sig_mt, BANDS, ROWS)
    dt_mt = time.time() - t2
    assert (sig_mt == sig).all()
    nbytes = int(o[-1])
    return {"value": n_sample / dt, "unit": "docs/s", "cores": cores, "kind": "port",
            "sample": f"BASELINE.json configs[0] (C1) in full: {n_sample} synthetic English docs ({nbytes / 1e6:.1f} MB) from the workload's "
                      f"generator, char 13-grams, K=128, 9x13 (the reference's fixed parameters); python port (numpy+hashlib, per-doc "
                      f"MinHash like the reference) on {cores} processes, {dt:.1f} s",
            "gb_per_s": nbytes / dt / 1e9,
            "c_port_1core_docs_per_s": n_sample / dt_c, "c_port_openmp_docs_per_s": n_sample / dt_mt,
            "c_port_openmp_threads": cores, "n_removed_sample": int((labels != np.arange(n_sample)).sum())}


def stage_benchmarks(eng, text, offsets, text_bytes, with_cpu=True):
    """Stages around MinHash in the EN pipeline on the bench corpus, text resident in
    HBM: device EN normalisation, exact document dedup, paragraph dedup, exact substring dedup.  All HBM-bound:
    each entry carries its algorithmic bytes / time against the 8 TB/s peak, and (with_cpu) the oracle
    restatement of the reference's Spark job timed on one host core over a bounded sample."""
    import torch

    from map_neo_amd import exact
    dev, n = eng.device, offsets.numel() - 1
    HBM = 8000.0

    def timed(fn, reps=2):
        r = fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            r = fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps, r

    def docs_of(t, o, k):
        o = o[: k + 1].cpu().numpy()
        b = t[: int(o[-1])].cpu().numpy().tobytes()
        return [b[int(o[i]):int(o[i + 1])].decode("utf8") for i in range(k)]

    out = {}
    # ---- rank 1 (ingest): the corpus as ONE uncompressed JSON-lines shard in HBM, {"text": "<document>"}\n per line, through the
    # device-side field extraction (line index, scan, decode: Engine.jsonl_extract, host bookkeeping and its synchronisations
    # included); the extracted text must be the corpus, byte for byte
    try:
        import ctypes as C

        from map_neo_amd.minhash import _lib, _ptr, padded_size
        pre, suf = b'{"text": "', b'"}\n'
        step = len(pre) + len(suf)
        starts = offsets[:-1] + step * torch.arange(n, device=dev) + len(pre)          # where document k's bytes go
        total = text_bytes + step * n
        raw = torch.zeros(padded_size(total), dtype=torch.uint8, device=dev)
        sel = torch.arange(n, device=dev)
        _lib.check(eng.L.mdd_gather_docs(eng.ctx, _ptr(text), text_bytes, _ptr(offsets), n, _ptr(sel), n, _ptr(starts), _ptr(raw), total),
                   "mdd_gather_docs")
        raw[((starts - len(pre))[:, None] + torch.arange(len(pre), device=dev)).flatten()] = torch.tensor(list(pre), dtype=torch.uint8, device=dev).repeat(n)
        raw[((offsets[1:] + step * torch.arange(n, device=dev) + len(pre))[:, None] + torch.arange(len(suf), device=dev)).flatten()] = \
            torch.tensor(list(suf), dtype=torch.uint8, device=dev).repeat(n)
        del sel, starts
        best = None
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            x = eng.jsonl_extract(raw, total, "text")
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) * 1e3
            best = dt if best is None else min(best, dt)
        same = (x["text_bytes"] == text_bytes and bool(torch.equal(x["offsets"], offsets)) and
                bool(torch.equal(x["text"][:text_bytes], text[:text_bytes])) and int((x["status"] != 0).sum()) == 0)
        assert same, "device-side JSON-lines extraction does not return the corpus"
        alg = 4 * total + text_bytes + 25 * n      # shard read by index count, index write, scan, decode; content written; per-line arrays
        out["jsonl_extract"] = {"ms": best, "lines": n, "jsonl_bytes": total, "jsonl_gb_per_s": total / best / 1e6, "docs_per_s": n / best * 1e3,
                                "algorithmic_bytes": alg, "hbm_frac": alg / best / 1e6 / HBM, "equals_corpus": same,
                                "note": "SURVEY 8(f) rank 1 on the input side: one uncompressed shard resident in HBM -> packed utf-8 + offsets "
                                        "(mdd_jsonl_count_lines / _line_begins / _scan / _decode through Engine.jsonl_extract, its host "
                                        "synchronisations included); not part of en_pipeline_total.  End to end from files: tools/e2e_cli.py"}
        del raw, x
    except Exception as e:      # noqa: BLE001 -- an extra row never costs the others
        out["jsonl_extract"] = {"error": repr(e)}
    g = torch.Generator(device=dev).manual_seed(1)
    r = torch.rand(text_bytes, device=dev, generator=g)
    body = text[:text_bytes]
    # un-normalised look: upper-case 1/8 of the letters, 1/16 of the spaces become commas
    raw = text.clone()
    rb = raw[:text_bytes]
    rb.copy_(torch.where((body >= 97) & (body <= 122) & (r < 0.125), body - 32, body))
    rb.copy_(torch.where((body == 32) & (r > 0.9375), torch.full_like(body, 44), rb))
    ms, (nt, no, fl) = timed(lambda: eng.normalize_en(raw, offsets))
    ob = int(no[-1].item())
    out["normalize_en"] = {"ms": ms, "docs_per_s": n / ms * 1e3, "text_gb_per_s": text_bytes / ms / 1e6,
                           "algorithmic_bytes": 2 * text_bytes + ob, "hbm_frac": (2 * text_bytes + ob) / ms / 1e6 / HBM}
    if with_cpu:
        from map_neo_amd import hostio
        sample = docs_of(raw, offsets, 3000)
        t0 = time.perf_counter()
        from oracle import ref_py as _R
        ref = [_R.normalize_text(d) for d in sample]
        assert ref == [hostio.normalize_text(d) for d in sample]
        dt = time.perf_counter() - t0
        assert ref == docs_of(nt, no, 3000)
        out["normalize_en"]["cpu_python_docs_per_s_1core"] = len(sample) / dt
    del raw, rb, nt, no, fl

    ms, keep = timed(lambda: exact.exact_document_dedup_packed(eng, text, offsets))
    out["exact_document_dedup"] = {"ms": ms, "docs_per_s": n / ms * 1e3, "text_gb_per_s": text_bytes / ms / 1e6,
                                   "algorithmic_bytes": text_bytes + 28 * n, "hbm_frac": (text_bytes + 28 * n) / ms / 1e6 / HBM,
                                   "n_removed": int((~keep).sum().item())}

    # paragraphs: every 40th space becomes a newline
    par = text.clone()
    pb = par[:text_bytes]
    pb.copy_(torch.where((body == 32) & (r < 0.025), torch.full_like(body, 10), body))
    del r
    ms, (pt, po, pd) = timed(lambda: exact.paragraph_dedup_packed(eng, par, offsets, 20))
    n_par = n + int((pb == 10).sum().item())
    out["paragraph_dedup"] = {"ms": ms, "docs_per_s": n / ms * 1e3, "text_gb_per_s": text_bytes / ms / 1e6, "paragraphs": n_par,
                              "bytes_out": int(po[-1].item()), "docs_out": int(pd.numel()),
                              "algorithmic_bytes": 2 * text_bytes + int(po[-1].item()) + 36 * n_par,
                              "hbm_frac": (2 * text_bytes + int(po[-1].item()) + 36 * n_par) / ms / 1e6 / HBM,
                              "note": "algorithmic = text read to cut + to hash, survivors written, 36 B of unit ranges / keys / "
                                      "table per paragraph; the implementation reads the text 4 times (count, units, hash, write "
                                      "-- round 2: 5; the per-document sums now come from the units, mdd_para_doc_sums, and a "
                                      "document that keeps every paragraph is copied)"}
    if with_cpu:
        from oracle import ref_exact
        k = 20000
        sample = docs_of(par, offsets, k)
        t0 = time.perf_counter()
        ref = ref_exact.paragraph_dedup(list(enumerate(sample)), 20)
        dt = time.perf_counter() - t0
        st, so, sd = exact.paragraph_dedup_packed(eng, par, offsets[: k + 1].contiguous(), 20)
        assert [t for _, t in ref] == docs_of(st, so, sd.numel()) and [i for i, _ in ref] == sd.cpu().tolist()
        out["paragraph_dedup"]["cpu_oracle_docs_per_s_1core"] = k / dt
        out["paragraph_dedup"]["cpu_sample"] = "first %d docs, oracle/ref_exact.paragraph_dedup (hashlib sha256), %.1f s" % (k, dt)
    del par, pb, pt, po, pd

    # substring dedup (fused kernels: windows are never materialised)
    k = n
    so = offsets[: k + 1].contiguous()
    sb = int(so[-1].item())
    ms, (st, so2, sd) = timed(lambda: exact.exact_substr_dedup_packed(eng, text, so, 50, 20), reps=1)
    n_win = sb - 50 * k
    out["exact_substr_dedup"] = {"ms": ms, "docs": k, "docs_per_s": k / ms * 1e3, "text_gb_per_s": sb / ms / 1e6,
                                 "windows_per_s": n_win / ms * 1e3, "bytes_out": int(so2[-1].item()), "docs_out": int(sd.numel()),
                                 "method": eng.last_substr_method, "passes": eng.last_substr_parts}
    # ALGORITHMIC bytes of the stage = what it must read and write whatever the method: the text in, one mark per byte
    # position written and read back, the surviving text out (the sort's own traffic is not algorithmic)
    alg = sb + 2 * sb + int(so2[-1].item())
    out["exact_substr_dedup"].update({"algorithmic_bytes": alg, "hbm_frac": alg / ms / 1e6 / HBM})
    if eng.last_substr_method == "sorted":
        cand = eng.last_substr_candidates
        n_sort = cand[0] if cand else n_win
        # implementation bytes (a model; the measured HBM traffic per kernel is in profiles/r3*_pmc_substr*): per SORTED entry
        # 10 B written (entry + d1), 34 B through the two scatter levels, 16 B read by the count pass and the leaves;
        # with the filter the anchors (one per ~10 windows) are such entries and only n_sort windows follow them; the text is
        # read by the map of window starts, the anchor kernel, emit (documents with a candidate) and the two compaction
        # passes; the flag / map / mark arrays (one byte per text byte) are written and read ~7 times in all
        n_anchor = n_win // 10 if cand else 0
        impl = (n_anchor + n_sort) * 60 + (5 if cand else 3) * sb + (7 if cand else 3) * sb + int(so2[-1].item())
        out["exact_substr_dedup"].update({
            "windows": n_win, "windows_to_window_sort": n_sort, "anchors_model": n_anchor,
            "implementation_bytes_model": impl, "implementation_hbm_frac": impl / ms / 1e6 / HBM,
            "note": "round 3: a window can only repeat when the 32-byte k-gram its bytes select (minimizer over its first 19 offsets) "
                    "occurs twice, so the ANCHORS (one per ~10 windows) are sorted into 2^20 leaves instead of the windows; a repeated "
                    "anchor is settled in its leaf by comparing the 68 bytes around it with the earlier anchors of its key (all 19 "
                    "windows at once), and only what that cannot vouch for (keys with > 32 anchors, non-ASCII regions) reaches the "
                    "window sort of round 2 (2^22 leaves, bytes compared on tag matches).  hbm_frac prices the stage against what it "
                    "must move (text in, marks, text out); the method's own traffic is implementation_bytes_model"})
    else:
        out["exact_substr_dedup"].update({
            "random_atomics": n_win, "atomic_floor_ms": n_win / 17.5e9 * 1e3,
            "note": "one random 8-byte atomicCAS per window; the device sustains ~17.5 G/s of those on "
                    "tables >> 256 MB (tools/atomic_ubench.hip), which bounds the stage"})
    if with_cpu:
        from oracle import ref_exact
        k = 1500
        sample = docs_of(text, offsets, k)
        t0 = time.perf_counter()
        ref = ref_exact.exact_substr_dedup([(d, i) for i, d in enumerate(sample)], 50, 20)
        dt = time.perf_counter() - t0
        st, so2, sd = exact.exact_substr_dedup_packed(eng, text, offsets[: k + 1].contiguous(), 50, 20)
        assert [t for _, t in ref] == docs_of(st, so2, sd.numel()) and [i for i, _ in ref] == sd.cpu().tolist()
        out["exact_substr_dedup"]["cpu_oracle_docs_per_s_1core"] = k / dt
        out["exact_substr_dedup"]["cpu_sample"] = "first %d docs, oracle/ref_exact.exact_substr_dedup (hashlib sha256 per window), %.1f s" % (k, dt)
    return out


def gpu_locality(index: int) -> dict:
    """NUMA placement facts for the streamed-input figures: the GPU's PCI address, its NUMA node and
    local CPUs from sysfs, and the CPUs this process may run on."""
    import torch
    info = {"allowed_cpus": len(os.sched_getaffinity(0))}
    try:
        pr = torch.cuda.get_device_properties(index)
        bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        info["pci"] = bdf
        base = "/sys/bus/pci/devices/" + bdf
        info["gpu_numa_node"] = int(open(base + "/numa_node").read())
        info["gpu_local_cpulist"] = open(base + "/local_cpulist").read().strip()
        nodes = [d for d in os.listdir("/sys/devices/system/node") if d.startswith("node")]
        info["numa_nodes"] = len(nodes)
    except Exception as e:      # noqa: BLE001 -- diagnostics only
        info["error"] = repr(e)
    return info


def cpus_of(cpulist: str):
    out = set()
    for part in cpulist.split(","):
        if "-" in part:
            a, b = part.split("-")
            out.update(range(int(a), int(b) + 1))
        elif part.strip():
            out.add(int(part))
    return out


def bench_config_id(n_global: int, corpus_config: str = "auto") -> int:
    return int(corpus_config) if corpus_config != "auto" else (2 if n_global < 50_000_000 else 4)


def bench_spec(n_global: int, mean_bytes: int, config_id: int):
    """The synthetic corpus of a run: ONE global spec (near-duplicates cross the ranks' shards), a function of the job's
    total size only -- so tools/bench_expected.py can hash the same corpus on one GPU (profiles/r3_bench_expected.jsonl)
    and a multi-GPU run can be held against that result (n_removed_matches_single_gpu)."""
    from map_neo_amd import corpus
    seed = 20240519 + config_id                   # config id
    if config_id == 4:      # C4 as SURVEY 8(d) describes it: 15 % near-copies in clusters of 2-50 (Zipf), edit-rate sweep, indels
        return corpus.CorpusSpec(n_global, mean_bytes, seed=seed, lang="en", dup_frac=0.15, clusters="zipf50",
                                 edit_rates=(0.01, 0.05, 0.1, 0.2), indel_rate=0.005)
    return corpus.CorpusSpec(n_global, mean_bytes, seed=seed, lang="en")


def expected_n_removed(n_global: int, mean_bytes: int, config_id: int):
    """n_removed of the single-GPU path on the same global corpus, when tools/bench_expected.py has measured it"""
    try:
        for line in open(os.path.join(ROOT, "profiles", "r3_bench_expected.jsonl")):
            r = json.loads(line)
            if (r["n_docs"], r["mean_bytes"], r["corpus_config"], r["K"], r["bands"], r["rows"]) == (n_global, mean_bytes, config_id, K, BANDS, ROWS):
                return int(r["n_removed"])
    except (OSError, ValueError, KeyError):
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--docs", type=int, default=None,
                    help="documents per GPU (default: 1 000 000 = C2 at N=1; 12 500 000 = one GPU's share of C4 at N>1)")
    ap.add_argument("--mean-bytes", type=int, default=2048)
    ap.add_argument("--corpus-config", choices=["auto", "2", "4"], default="auto",
                    help="synthetic corpus: 2 = C2's generator settings, 4 = C4's (Zipf clusters of 2-50, edit-rate sweep, token "
                         "insert/delete); auto = 4 when the job holds >= 50 M documents (the N > 1 default), else 2")
    ap.add_argument("--cpu-sample-docs", type=int, default=10000,
                    help="documents of the CPU baseline (default: BASELINE.json configs[0] in full = 10 000 docs)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-pcie", action="store_true", help="skip the optional PCIe / pinned-host measurements")
    ap.add_argument("--stages", action="store_true", help="(default at N=1) kept for compatibility")
    ap.add_argument("--no-stages", action="store_true",
                    help="skip the stages around MinHash (SURVEY 8(f) ranks 2-4) that are otherwise timed after the headline "
                         "measurement and reported under next_rows")
    ap.add_argument("--force-dist", action="store_true",
                    help="use the multi-GPU path (RCCL all-to-all / all-gather) even with one rank")
    args = ap.parse_args()

    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints its version banner to
    # fd 1 from C code), so fd 1 is pointed at stderr for the whole run and the result goes to the saved fd.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.docs is None:
        args.docs = 1_000_000 if world == 1 else 12_500_000
    config_id = bench_config_id(args.docs * world, args.corpus_config)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline(args.cpu_sample_docs, args.mean_bytes, 20240519 + 1)   # C1's seed; before any GPU init (uses fork)

    import torch
    import torch.distributed as dist

    from map_neo_amd import corpus
    from map_neo_amd.dedup import StageTimer, dedup_distributed, dedup_local
    from map_neo_amd.minhash import Engine, padded_size

    # rehearsal of the multi-rank script on a one-GPU box: MDD_BENCH_BACKEND=gloo puts every rank on GPU 0 and lets
    # gloo carry the tensors (the driver's runs use the default: one GPU per rank, RCCL)
    backend = os.environ.get("MDD_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    eng = Engine(local_rank)
    dev = eng.device

    n_local = args.docs
    n_global = n_local * world
    # one global corpus spec; each rank generates its contiguous shard (near-duplicates cross shards)
    spec = bench_spec(n_global, args.mean_bytes, config_id)
    d0 = rank * n_local
    t_gen = time.time()
    texts, offs, base = [], [torch.zeros(1, dtype=torch.int64, device=dev)], 0
    for c0 in range(d0, d0 + n_local, 32768):
        c1 = min(d0 + n_local, c0 + 32768)
        t, o = corpus.generate(spec, c0, c1, dev)
        nb = int(o[-1].item())
        texts.append(t[:nb])
        offs.append(o[1:] + base)
        base += nb
    text_bytes = base
    text = torch.zeros(padded_size(text_bytes), dtype=torch.uint8, device=dev)
    torch.cat(texts, out=text[:text_bytes])
    offsets = torch.cat(offs)
    del texts, offs
    torch.cuda.synchronize()
    t_gen = time.time() - t_gen
    # PCIe-inclusive figure (reported, never `value`): time one H2D of the packed text from pinned host memory
    h2d_ms = None
    pipelined_sig_ms = None
    pipelined_equal = None
    pinned = None
    if rank == 0 and world == 1 and text_bytes < (8 << 30) and not args.no_pcie:
        try:
            stage = torch.empty_like(text)

            def h2d_time(buf):
                best = None
                for _ in range(3):
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    stage.copy_(buf, non_blocking=True)
                    e1.record()
                    torch.cuda.synchronize()
                    best = e0.elapsed_time(e1) if best is None else min(best, e0.elapsed_time(e1))
                return best

            # Where the pinned buffer lives matters (58 vs 63 ms between boxes): it is allocated twice,
            # once wherever this thread happens to run and once with the thread confined to the GPU's local CPUs (sysfs
            # local_cpulist) while the pages are allocated and first touched; the faster one carries the streamed passes
            # and both H2D times are reported.
            locality = gpu_locality(dev.index if dev.index is not None else 0)
            host = torch.empty(text.numel(), dtype=torch.uint8, pin_memory=True)
            host.copy_(text)
            h2d_default = h2d_time(host)
            h2d_local = None
            try:
                allowed = os.sched_getaffinity(0)
                local = cpus_of(locality.get("gpu_local_cpulist", "")) & allowed
                if local and local != allowed:
                    os.sched_setaffinity(0, local)
                    try:
                        host2 = torch.empty(text.numel(), dtype=torch.uint8, pin_memory=True)
                        host2.copy_(text)
                    finally:
                        os.sched_setaffinity(0, allowed)
                    h2d_local = h2d_time(host2)
                    if h2d_local < h2d_default:
                        host = host2
                    del host2
                locality["local_cpus_allowed"] = len(local)
            except Exception as e:      # noqa: BLE001 -- diagnostics only
                locality["affinity_error"] = repr(e)
            h2d_ms = min(h2d_default, h2d_local) if h2d_local is not None else h2d_default
            locality.update(h2d_ms_default_placement=h2d_default, h2d_ms_gpu_local_placement=h2d_local)
            del stage
            # SURVEY 8(d)'s region: packed text + offsets in PINNED HOST memory -> removed mask on the host.  ONE
            # signature launch hashes the documents while the text is still arriving over PCIe
            # (mdd_signatures_from_host), then band keys / buckets / union-find, then the mask is copied back.
            from map_neo_amd.dedup import cluster_local
            offs_pin = offsets.cpu().pin_memory()
            removed_pin = torch.empty(n_local, dtype=torch.uint8).pin_memory()
            tdev = torch.empty_like(text)
            sig_buf = torch.empty((n_local, K), dtype=torch.int32, device=dev)

            def pinned_pass():
                od = offs_pin.to(dev, non_blocking=True)
                sg, _ = eng.signatures_from_host(host, od, K, WIDTH, "char", text_bytes=text_bytes, text_dev=tdev, out=sig_buf)
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                o = cluster_local(eng, sg, BANDS, ROWS)
                removed_pin.copy_(o["removed"], non_blocking=True)
                return o, ev

            pinned_pass()
            eng.sync()
            pin_ms, pin_sig_ms = [], []
            for _ in range(3):
                torch.cuda.synchronize()
                e0 = torch.cuda.Event(enable_timing=True)
                tp = time.perf_counter()
                e0.record()
                o, ev = pinned_pass()
                eng.sync()                          # raises if the streamed kernel gave up waiting for its text
                torch.cuda.synchronize()
                pin_ms.append((time.perf_counter() - tp) * 1e3)
                pin_sig_ms.append(e0.elapsed_time(ev))
            pinned = {"ms_per_pass": min(pin_ms), "ms_all_passes": pin_ms, "signatures_ms_all_passes": pin_sig_ms, "locality": locality,
                      "docs_per_s": n_local / (min(pin_ms) * 1e-3),
                      "text_gb_per_s": text_bytes / (min(pin_ms) * 1e6), "signatures_ms": min(pin_sig_ms),
                      "kernel_wait_wave_ms": eng.stream_wait_ms(), "n_removed": int(removed_pin.sum().item()),
                      "region": "packed utf-8 + offsets in pinned host memory -> removed mask in pinned host memory (SURVEY 8(d)); "
                                "best of 3 passes, wall clock around enqueue + synchronize"}
            pipelined_sig_ms = min(pin_sig_ms)
            pipelined_equal = bool(torch.equal(sig_buf, eng.signatures(text, offsets, K, WIDTH, "char", text_bytes=text_bytes)))
            del host, tdev, sig_buf, o
        except Exception as e:  # pinned allocation can fail on small hosts: the figures are optional
            print(f"[bench] PCIe figures skipped: {e!r}", file=sys.stderr)
            pinned = {"error": repr(e)}
    # achievable HBM bandwidth on this device (calibrate with a device copy): read + write of the text
    hbm_copy_gbps = None
    if rank == 0:
        dst = torch.empty_like(text)
        dst.copy_(text)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dst.copy_(text)
        e1.record()
        torch.cuda.synchronize()
        hbm_copy_gbps = 2 * text.numel() * 5 / (e0.elapsed_time(e1) * 1e6)
        del dst
    lens = offsets[1:] - offsets[:-1]
    n_shingles = int(torch.clamp(lens - (WIDTH - 1), min=0).sum().item())      # ASCII corpus: code points == bytes

    dist_timings = {}

    def step(timer=None):
        if not use_dist:
            return dedup_local(eng, text, offsets, K, WIDTH, "char", BANDS, ROWS, text_bytes=text_bytes, timer=timer)
        return dedup_distributed(eng, text, offsets, d0, n_global, K, WIDTH, "char", BANDS, ROWS,
                                 text_bytes=text_bytes, timer=timer, timings=dist_timings)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    out = None
    for _ in range(args.warmup):
        out = step()
    barrier()
    timer = StageTimer()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(timer)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    stages = timer.result_ms()
    n_removed = int(out["n_removed"].item())
    if world > 1:
        nr = torch.tensor([n_removed], dtype=torch.int64, device=dev)
        dist.all_reduce(nr)
        n_removed = int(nr.item())

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        sig_ms = stages.get("signatures", 0.0) / args.steps       # HIP events on the launch stream, timed region
        is_c2 = (n_local == 1_000_000 and args.mean_bytes == 2048)
        # HBM bytes per sig_kernel launch: PMC passes of this same command (tools/pmc_bench.sh ->
        # profiles/<round>_pmc_bench/summary.json, FETCH_SIZE corrected x2 as calibrated); never measured inside this
        # run (counters need rocprofv3), so the line names the file it comes from, or null if the workload differs
        traffic, traffic_source = None, None
        for cand in ("r3g_pmc_bench", "history/r3_pmc_bench", "r2h_pmc_bench", "history/r2_pmc_bench", "history/r1m_pmc_bench"):
            try:
                pm = json.load(open(os.path.join(ROOT, "profiles", cand, "summary.json")))
                if pm["workload"]["text_bytes_per_gpu"] == text_bytes and pm["workload"]["docs_per_gpu"] == n_local:
                    k = pm["kernels"].get("sig_kernel<2, false, false>") or pm["kernels"]["sig_kernel<2, false>"]
                    traffic, traffic_source = k["fetch_bytes"] + k["write_bytes"], "profiles/%s/summary.json" % cand
                    break
            except Exception:
                pass
        # VALU issue utilisation: the instruction mix sig_kernel executes on THIS workload (static ISA histogram x
        # dynamic region counts, closed against SQ_INSTS_VALU) priced with per-opcode SIMD cycles measured on this
        # chip (profiles/r2_issue/issue_model.json, tools/issue_model.py) / the cycles the launch had: kernel time
        # measured live x 1024 SIMDs x clock.  At the 2.4 GHz peak clock this is a LOWER bound of the utilisation
        # (the chip runs this kernel at ~2.38 GHz); it cannot exceed 1 by construction of the costs.
        issue = None
        try:
            im = json.load(open(os.path.join(ROOT, "profiles", "r2_issue", "issue_model.json")))
            w = im["workload"]
            if (w["n_docs"], w["text_bytes"], w["shingles"], w["K"], w["width"]) == (n_local, text_bytes, n_shingles, K, WIDTH) and sig_ms:
                cyc = im["issue_cycles_per_shingle"] * n_shingles / 1024.0
                issue = {"issue_frac": cyc / (sig_ms * 1e-3 * 2.4e9), "clock_ghz_assumed": 2.4,
                         "issue_frac_at_profiled_clock": cyc / (sig_ms * 1e-3 * im["clock_ghz_under_pmc"] * 1e9),
                         "clock_ghz_profiled": im["clock_ghz_under_pmc"],
                         "issue_cycles_per_shingle": im["issue_cycles_per_shingle"],
                         "valu_instructions_per_shingle": im["valu_wave_instructions_per_shingle"],
                         "avg_issue_cycles_per_valu_instruction": im["avg_issue_cycles_per_valu_instruction"],
                         "closure_model_over_SQ_INSTS_VALU": im["closure"]["SQ_INSTS_VALU"]["model_over_pmc"],
                         "source": "profiles/r2_issue/issue_model.json (instruction mix and per-opcode cycles: profiled; kernel time: this run)"}
        except Exception as e:
            print(f"[bench] issue model not applied: {e!r}", file=sys.stderr)
        try:
            peak_cal = eng.valu_calibrate(20000)       # measured: dependency-free v_xor / v_add mix on every CU
        except Exception:
            peak_cal = None
        achieved = n_shingles * OPS_PER_SHINGLE(K) / (sig_ms * 1e-3) / 1e12 if sig_ms else None
        workload = ("C2: %d synthetic English docs per GPU (~%d B avg)" if config_id == 2 else
                    "C4 share: %d synthetic English docs per GPU (~%d B avg; 100M docs on 8 GPUs = 12.5M per GPU), 15 %% injected near-duplicates") \
            % (n_local, args.mean_bytes)
        res = {
            "metric": "docs/sec MinHash-LSH dedup (signatures->bands->buckets->union-find), text resident in HBM",
            "value": n_global / (dt / args.steps),
            "unit": "docs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32/u64",
            "data": "synthetic",
            "config": {"workload": workload + ", char 13-gram shingles, num_perm=128, 9 bands x 13 rows, seed 1",
                       "docs_per_gpu": n_local, "global_docs": n_global, "text_bytes_per_gpu": text_bytes,
                       "shingles_per_gpu": n_shingles, "parallelism": "dp%d" % world,
                       "path": "distributed (RCCL all-to-all + all-gather)" if use_dist else "single-GPU"},
            "text_gb_per_s": text_bytes * world / (dt / args.steps) / 1e9,
            "resident_docs_per_s": n_global / (dt / args.steps),
            # SURVEY 8(d)'s region (pinned host -> removed mask on host); the task contract keeps `value` HBM-resident
            "pinned_host": pinned,
            "pinned_host_docs_per_s": pinned.get("docs_per_s") if pinned else None,
            "n_removed": n_removed,
            # cross-check that survives any path: the generator injected this many (near-)duplicates; a path that loses
            # band records or edges removes visibly fewer documents than that
            "n_injected_duplicates": int(spec.is_dup.sum()),
            # the strongest check of the multi-GPU path that needs no second run: the SAME global corpus clustered on one GPU
            # (tools/bench_expected.py -> profiles/r3_bench_expected.jsonl) must lose exactly as many documents
            "n_removed_single_gpu_same_corpus": expected_n_removed(n_global, args.mean_bytes, config_id),
            "n_removed_matches_single_gpu": (None if expected_n_removed(n_global, args.mean_bytes, config_id) is None
                                             else expected_n_removed(n_global, args.mean_bytes, config_id) == n_removed),
            "stage_ms_per_step_rank0": {k: v / args.steps for k, v in stages.items()},
            "corpus_gen_s": t_gen,
            "h2d_text_ms": h2d_ms,
            "signatures_from_pinned_host_ms": pipelined_sig_ms,
            "signatures_from_pinned_host_equal_resident": pipelined_equal,
            "pcie_inclusive_docs_per_s": (n_local / ((ms_per_step + h2d_ms) * 1e-3)) if (h2d_ms and world == 1) else None,
            "roofline": {"bound": "valu", "kernel": "sig_kernel<2,false>", "achieved": achieved, "peak": PEAK_VALU_TOPS,
                         "unit": "Tops/s (int32 lane-ops)", "frac": (achieved / PEAK_VALU_TOPS) if achieved else None,
                         "frac_is": "speed-up over SURVEY 8(d)'s nominal 958 + 14 K ops per shingle executed one op per lane-slot "
                                    "(NOT a utilisation: fused instructions and the permute-phase filter retire / skip most of them); "
                                    "the utilisation is issue_frac",
                         "issue_frac": issue["issue_frac"] if issue else None, "issue": issue,
                         "peak_calibrated": peak_cal, "peak_calibrated_unit": "Tops/s measured by mdd_valu_calibrate in this run",
                         "traffic": traffic, "traffic_source": traffic_source, "traffic_unit": "HBM bytes per launch (PMC)",
                         "algorithmic_bytes": text_bytes + 8 * (n_local + 1) + 4 * K * n_local, "kernel_ms": sig_ms,
                         "algorithmic_ops_per_shingle": OPS_PER_SHINGLE(K),
                         "hbm_view": ({"bound": "hbm", "achieved": (text_bytes + 8 * (n_local + 1) + 4 * K * n_local) / (sig_ms * 1e6),
                                       "peak": 8000.0, "unit": "GB/s",
                                       "frac": (text_bytes + 8 * (n_local + 1) + 4 * K * n_local) / (sig_ms * 1e6) / 8000.0,
                                       "note": "the same launch priced against HBM: <1 % of 8 TB/s, i.e. the kernel is "
                                               "nowhere near memory-bound; ~2 750 integer ops per text byte put it on the VALU"}
                                      if sig_ms else None),
                         "note": "peak = 256 CU x 128 lanes/clk x 2.4 GHz simple int32 lane-ops.  issue_frac = VALU issue cycles the "
                                 "executed instruction mix needs (per-opcode SIMD cycles measured with s_memtime; instruction counts "
                                 "= ISA histogram x region counts, within 1 % of SQ_INSTS_VALU) / (kernel time x 1024 SIMDs x 2.4 GHz): "
                                 "DESIGN.md 4.1, profiles/r2_issue/"},
        }
        if use_dist:
            st = res["stage_ms_per_step_rank0"]
            ex = dict(dist_timings or {})
            a2a_ms, ag_ms = st.get("all_to_all"), st.get("all_gather")
            ex["all_to_all_ms"] = a2a_ms
            ex["all_gather_ms"] = ag_ms
            if a2a_ms:
                ex["all_to_all_gb_per_s_rank0"] = (ex.get("all_to_all_bytes_sent", 0) + ex.get("all_to_all_bytes_received", 0)) / (a2a_ms * 1e6)
            if ag_ms:
                ex["all_gather_gb_per_s_rank0"] = ex.get("all_gather_bytes_received", 0) / (ag_ms * 1e6)
            ex["note"] = "rank 0's figures for the last step; bytes = records x (4 x rows + 8); ms from events on the exchange stream"
            res["exchange"] = ex
        # HBM-bound stages: SURVEY 8(d)'s algorithmic bytes per document against the 8 TB/s peak
        hbm_alg = {"band_keys": 4 * K + BANDS * 20, "bucket_edges": 2 * BANDS * 20, "components": BANDS * 8 + 4}
        res["hbm_stages"] = {k: {"ms": res["stage_ms_per_step_rank0"].get(k), "algorithmic_bytes_per_doc": v,
                                 "achieved_gb_per_s": v * n_local / (res["stage_ms_per_step_rank0"][k] * 1e6),
                                 "frac_of_8TBps": v * n_local / (res["stage_ms_per_step_rank0"][k] * 1e6) / 8000.0}
                             for k, v in hbm_alg.items() if res["stage_ms_per_step_rank0"].get(k)}
        res["hbm_copy_gb_per_s_measured"] = hbm_copy_gbps
        if cpu is not None:
            res["cpu_baseline"] = cpu
        if world == 1 and not args.no_stages and not use_dist:
            del out
            try:   # never let the extra rows cost the headline line
                res["next_rows"] = stage_benchmarks(eng, text, offsets, text_bytes, with_cpu=not args.no_cpu)
                stage_sum = sum(v["ms"] for k, v in res["next_rows"].items() if isinstance(v, dict) and "ms" in v and k != "jsonl_extract")
                res["next_rows"]["en_pipeline_total"] = {
                    "ms": stage_sum + res["ms_per_step"], "docs_per_s": n_local / (stage_sum + res["ms_per_step"]) * 1e3,
                    "note": "normalisation + exact document + MinHash-LSH (the headline step) + paragraph + exact substring, each "
                            "timed on the full corpus resident in HBM (the CLI runs them on the survivors of the step before)"}
            except Exception as e:
                res["next_rows"] = {"error": repr(e)}
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(res) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
