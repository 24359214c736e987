#!/usr/bin/env python3
"""Turns gpurun_out/pmc_bench/* (tools/pmc_bench.sh) into profiles/r1_pmc_bench/summary.json:
HBM bytes per launch for every kernel of the bench step, corrected as MI355X_MICROARCH.md
prescribes (FETCH_SIZE counts 64 B per 128-B request on gfx950: x2, confirmed here by the
calibration kernels with known byte counts; WRITE_SIZE is exact)."""
import collections
import csv
import json
import os
import re
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_bench"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r1_pmc_bench"
os.makedirs(dst, exist_ok=True)


def load(path):
    out = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        m = re.search(r"(\w+)(<[^>]*>)?\(", r["Kernel_Name"].replace("(anonymous namespace)::", ""))
        name = (m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"]
        out.setdefault(name, []).append(float(r["Counter_Value"]) * 1024.0)      # counters are in KiB
    return out


cal_f = load(f"{src}/cal_FETCH_SIZE/cal_counter_collection.csv")
cal_w = load(f"{src}/cal_WRITE_SIZE/cal_counter_collection.csv")
known = float(2 << 30)
fetch_factor = {k: known / v[0] for k, v in cal_f.items() if k.startswith("cal_read")}
write_factor = known / cal_w["cal_write_dword"][0]
# scattered requests (round 3): 2^25 requests, one 64-byte line each -> counted bytes per request
N_REQ = float(1 << 25)
scatter = {"fetch_bytes_counted_per_scattered_16B_load": cal_f.get("cal_read_scatter16", [0])[0] / N_REQ,
           "fetch_bytes_counted_per_scattered_4B_load": cal_f.get("cal_read_scatter4", [0])[0] / N_REQ,
           "write_bytes_counted_per_scattered_1B_store": cal_w.get("cal_write_scatter1", [0])[0] / N_REQ,
           "fetch_bytes_counted_by_the_1B_stores_per_request": cal_f.get("cal_write_scatter1", [0])[0] / N_REQ}
# kernels whose reads are scattered single-line requests: the x2 of streaming reads does not apply to them
SCATTERED = ("leaf_kernel", "bigleaf_kernel", "bucket_insert_kernel", "bucket_ids_kernel", "exact_insert_kernel", "exact_link_kernel",
             "win_kernel", "uf_union_kernel", "uf_flatten_kernel", "anchor_resolve_kernel", "anchor_leaf_kernel")
ff = fetch_factor["cal_read_dword"]
bf = load(f"{src}/bench_FETCH_SIZE/bench_counter_collection.csv")
bw = load(f"{src}/bench_WRITE_SIZE/bench_counter_collection.csv")
kern = {}
for k in bf:
    # last dispatch of each kernel = the timed step (the first one is the warm-up)
    kern[k] = {"fetch_bytes": bf[k][-1] * ff, "write_bytes": bw.get(k, [0])[-1] * write_factor,
               "raw_FETCH_SIZE_KiB": bf[k][-1] / 1024, "raw_WRITE_SIZE_KiB": bw.get(k, [0])[-1] / 1024}
    if k.split("<")[0] in SCATTERED:
        # mostly single-line requests: FETCH_SIZE counts them at face value (see calibration.scattered); the truth lies
        # between the two figures, nearer this one the more of the kernel's reads are scattered
        kern[k]["fetch_bytes_if_all_reads_scattered"] = bf[k][-1]
        kern[k]["fetch_correction_note"] = "x2 validated on streaming reads only; scattered 64-byte requests are counted at face value"
bench_line = json.loads(open(f"{src}/bench_FETCH_SIZE.json").read().strip().splitlines()[-1])
# the exact-substring stage as one figure (PMC total per pass): every kernel of the stage, last launch each
SUBSTR = ("vmap_kernel", "anchor_kernel", "anchor_resolve_kernel", "anchor_leaf_kernel", "smear_kernel", "emit_kernel", "count2_kernel",
          "scatter_kernel", "leaf_kernel", "bigleaf_kernel", "compact4_kernel", "compact_kernel", "scan_", "tile_map_kernel", "seg_plan_kernel")
stage = {"fetch_bytes": 0.0, "write_bytes": 0.0, "kernels": []}
for k, v in kern.items():
    if k.split("<")[0].startswith(SUBSTR):
        n_launch = 2 if k.split("<")[0] in ("scatter_kernel", "count2_kernel", "tile_map_kernel") else 1   # anchors + windows
        for j in range(1, n_launch + 1):
            if len(bf[k]) >= j:
                stage["fetch_bytes"] += bf[k][-j] * (1.0 if k.split("<")[0] in SCATTERED else ff)
                stage["write_bytes"] += (bw.get(k, [0] * j)[-j] if len(bw.get(k, [])) >= j else 0) * write_factor
        stage["kernels"].append(k)
stage["total_bytes"] = stage["fetch_bytes"] + stage["write_bytes"]
summary = {"calibration": {"known_bytes": known, "fetch_correction": fetch_factor, "write_correction": write_factor, "scattered": scatter},
           "workload": bench_line["config"], "kernels": kern,
           "exact_substr_stage_per_pass": stage}
json.dump(summary, open(f"{dst}/summary.json", "w"), indent=1)
for f in ("cal_FETCH_SIZE/cal_counter_collection.csv", "cal_WRITE_SIZE/cal_counter_collection.csv",
          "bench_FETCH_SIZE/bench_counter_collection.csv", "bench_WRITE_SIZE/bench_counter_collection.csv"):
    rows = list(csv.DictReader(open(f"{src}/{f}")))
    keep = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "Counter_Name", "Counter_Value"]
    with open(f"{dst}/{f.replace('/', '_')}", "w", newline="") as fo:
        w = csv.DictWriter(fo, fieldnames=keep)
        w.writeheader()
        for r in rows:
            r["Kernel_Name"] = r["Kernel_Name"][:120]
            w.writerow({k: r[k] for k in keep})
print(json.dumps(summary["kernels"], indent=1))
print(json.dumps(stage, indent=1))
