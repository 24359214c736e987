#!/usr/bin/env python3
"""Every BASELINE.json configuration at its FULL stated size on ONE MI355X (a script, not collected by pytest; the
scaled-down crossing of the same code path is tests/test_gpu_full_path.py).  The text of C3 / C4 / C5 (44 / 218 / 100 GB)
is generated and hashed block by block (dedup.signatures_streamed: text dropped, signatures kept), then ONE clustering
call runs over all band records (C4: 100 M documents, 900 M records, a 2^31-slot bucket table).

No oracle can run at this size, so each run asserts size-independent properties:
  * labels are flattened, labels[d] <= d, every edge joins two documents of one label, n_removed = n - components;
  * every EXACT copy the generator injected (near-copies that drew the edit rate 0) shares its source's signature and label;
  * the clustering repeated with the surrogate key cut to 36 bits (thousands of times more tag collisions, all resolved on
    the exact tuples) gives identical labels;
  * 400 documents sampled from the LAST block match the C oracle's signatures bit for bit (K = 128 and 256);
and reports per-stage milliseconds, docs/s and the share of injected near-copies found per edit rate.
One JSON line per (config, b, r, K) -> profiles/r3_full_configs.jsonl."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from map_neo_amd import corpus  # noqa: E402
from map_neo_amd.dedup import StageTimer, cluster_local, signatures_streamed  # noqa: E402
from map_neo_amd.minhash import Engine, as_u32, pack_texts  # noqa: E402


def blocks_of(spec, device, block_docs, chunk_docs):
    for d0 in range(0, spec.n_docs, block_docs):
        text, offs = corpus.generate_chunked(spec, device, chunk_docs, d0, min(spec.n_docs, d0 + block_docs))
        yield text, offs, int(offs[-1].item())


def check_labels(eng, out, n, tag):
    """size-independent properties of a clustering result"""
    dev = eng.device
    lab = out["labels"].to(torch.int64) & 0xFFFFFFFF
    idx = torch.arange(n, device=dev)
    assert bool((lab <= idx).all()), tag + ": a label above its document"
    assert torch.equal(lab[lab], lab), tag + ": labels are not flattened"
    eu = out["edge_u"].to(torch.int64) & 0xFFFFFFFF
    ev = out["edge_v"].to(torch.int64) & 0xFFFFFFFF
    assert bool((lab[eu] == lab[ev]).all()), tag + ": an edge joins two labels"
    n_comp = int((lab == idx).sum().item())
    n_removed = int(out["n_removed"].item())
    assert n_removed == n - n_comp, tag + ": n_removed != n - components"
    assert int(out["removed"].sum(dtype=torch.int64).item()) == n_removed
    return lab, n_comp, n_removed


def run_config(eng, name, spec, settings, block_docs, sink, width=13, oracle_sample=400):
    from oracle import ref_c as RC            # the checker (test infrastructure)
    n = spec.n_docs
    Ks = tuple(sorted({K for _, _, K in settings}))
    chunk_docs = max(512, int(32768 * 2048 / spec.mean_bytes))
    acc = {"sig_ms": 0.0, "bytes": 0, "last": None}

    def on_block(d0, text, offs, nbytes, ms):
        acc["sig_ms"] += ms
        acc["bytes"] += nbytes
        acc["last"] = (d0, text, offs, nbytes)
        if (d0 // block_docs) % 16 == 0:
            print("  %s: %d / %d documents hashed" % (name, d0 + offs.numel() - 1, n), file=sys.stderr, flush=True)

    t0 = time.time()
    sigs = signatures_streamed(eng, blocks_of(spec, eng.device, block_docs, chunk_docs), n, Ks, width, "char", on_block=on_block)
    torch.cuda.synchronize()
    wall_sig = time.time() - t0
    # --- oracle sample from the last block
    d0, text, offs, nbytes = acc["last"]
    nd = offs.numel() - 1
    pick = np.sort(np.random.Generator(np.random.PCG64(5)).choice(nd, min(oracle_sample, nd), replace=False))
    o = offs.cpu().numpy()
    th = text[:nbytes].cpu().numpy()
    buf, so = pack_texts([bytes(th[o[i]:o[i + 1]]) for i in pick])
    for K in Ks:
        exp = RC.signatures(buf, so, K, width, "char")
        got = as_u32(sigs[K][torch.from_numpy(d0 + pick).to(eng.device)])
        assert (got == exp).all(), "%s: signatures of the oracle sample differ (K=%d)" % (name, K)
    acc["last"] = None
    del text, offs
    # --- injected copies
    src = torch.from_numpy(spec.src).to(eng.device)
    is_dup = torch.from_numpy(spec.is_dup).to(eng.device)
    if spec.rate_idx is not None:
        ridx = torch.from_numpy(spec.rate_idx.astype(np.int64)).to(eng.device)
        rates = [float(r) / 2.0 ** 32 for r in spec.rate_table]
    else:
        ridx = torch.zeros(n, dtype=torch.int64, device=eng.device)
        rates = [float(spec.rate_table[0]) / 2.0 ** 32]
    exact = is_dup & torch.tensor([r == 0.0 for r in rates], device=eng.device)[ridx]
    ex_idx = torch.nonzero(exact).flatten()
    for K in Ks:
        for c in range(0, ex_idx.numel(), 1 << 22):
            e = ex_idx[c:c + (1 << 22)]
            assert torch.equal(sigs[K][e], sigs[K][src[e]]), "%s: an exact copy has another signature than its source" % name
    for (bands, rows, K) in settings:
        sig = sigs[K]
        torch.cuda.synchronize()
        timer = StageTimer()
        timer.mark("start")
        t0 = time.perf_counter()
        out = cluster_local(eng, sig, bands, rows, timer=timer)
        torch.cuda.synchronize()
        cl_ms = (time.perf_counter() - t0) * 1e3
        st = timer.result_ms()
        tag = "%s b%dxr%d K%d" % (name, bands, rows, K)
        lab, n_comp, n_removed = check_labels(eng, out, n, tag)
        assert torch.equal(lab[ex_idx], lab[src[ex_idx]]), tag + ": an exact copy is not in its source's component"
        assert n_removed >= int(ex_idx.numel())
        found = {}
        for i, r in enumerate(rates):
            m = is_dup & (ridx == i)
            cnt = int(m.sum().item())
            found["%g" % r] = round(float((lab[m] == lab[src[m]]).sum().item()) / max(1, cnt), 4)
        n_edges, n_coll = int(out["edge_u"].numel()), out["n_collisions"]
        lab64 = out["labels"].clone()
        del out, lab
        torch.cuda.empty_cache()
        out2 = cluster_local(eng, sig, bands, rows, key_bits=36)
        assert torch.equal(out2["labels"], lab64), tag + ": labels change when the surrogate key is cut to 36 bits"
        coll36 = out2["n_collisions"]
        del out2, lab64
        torch.cuda.empty_cache()
        res = {"config": tag, "n_docs": n, "text_bytes": acc["bytes"], "K": K, "bands": bands, "rows": rows, "n_records": n * bands,
               "signature_kernels_ms_all_K": round(acc["sig_ms"], 1), "K_hashed_per_pass": list(Ks),
               "cluster_ms": round(cl_ms, 1), "stage_ms": {k: round(v, 2) for k, v in st.items()},
               "n_removed": n_removed, "n_components": n_comp, "n_edges": n_edges, "tag_collisions_64bit": n_coll,
               "tag_collisions_36bit": coll36, "injected_copies": int(spec.is_dup.sum()), "exact_copies": int(ex_idx.numel()),
               "copies_found_by_edit_rate": found, "oracle_sample_docs": int(pick.size), "wall_generate_plus_hash_s": round(wall_sig, 1),
               "block_docs": block_docs, "checks": "labels flattened, <= index, edges inside labels, n_removed = n - components, "
                                                   "exact copies share signature + label, labels equal with 36-bit keys, oracle sample"}
        if len(Ks) == 1:
            res["docs_per_s"] = round(n / ((acc["sig_ms"] + cl_ms) * 1e-3))
            res["text_gb_per_s"] = round(acc["bytes"] / ((acc["sig_ms"] + cl_ms) * 1e-3) / 1e9, 2)
        print(json.dumps(res), flush=True)
        if sink:
            with open(sink, "a") as f:
                f.write(json.dumps(res) + "\n")
    del sigs
    torch.cuda.empty_cache()


RATES = (0.0, 0.01, 0.05, 0.1, 0.2)      # SURVEY 8(d)'s sweep plus exact copies (rate 0: the property checks need them)


def main():
    which = sys.argv[1:] or ["C2", "C3", "C4", "C5"]
    scale = float(os.environ.get("MDD_FULL_SCALE", "1"))      # < 1: rehearsal of the script itself on a smaller corpus
    sink = os.environ.get("MDD_FULL_SINK", "")
    eng = Engine(0)
    S = lambda n: max(1000, int(n * scale))      # noqa: E731
    if "C2" in which:
        run_config(eng, "C2", corpus.CorpusSpec(S(1_000_000), 2048, lang="en", edit_rates=RATES), [(9, 13, 128)], 262144, sink)
    if "C3" in which:
        run_config(eng, "C3", corpus.CorpusSpec(S(20_000_000), 2048, seed=20240519 + 3, lang="mixed", edit_rates=RATES),
                   [(20, 12, 256)], 262144, sink)
    if "C4" in which:
        run_config(eng, "C4", corpus.CorpusSpec(S(100_000_000), 2048, seed=20240519 + 4, lang="en", dup_frac=0.15, edit_rates=RATES,
                                                indel_rate=0.005, clusters="zipf50"), [(9, 13, 128)], 262144, sink)
    if "C5" in which:
        spec = corpus.CorpusSpec(S(5_000_000), 20480, seed=20240519 + 5, lang="en", dup_frac=0.15, edit_rates=RATES, indel_rate=0.002,
                                 boiler_frac=0.05, boiler_share=0.9, n_templates=200)
        run_config(eng, "C5", spec, [(9, 13, 128), (14, 9, 128), (20, 12, 256)], 32768, sink)


if __name__ == "__main__":
    main()
