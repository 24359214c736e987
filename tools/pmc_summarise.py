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
ff = fetch_factor["cal_read_dword"]
bf = load(f"{src}/bench_FETCH_SIZE/bench_counter_collection.csv")
bw = load(f"{src}/bench_WRITE_SIZE/bench_counter_collection.csv")
kern = {}
for k in bf:
    # last dispatch of each kernel = the timed step (the first one is the warm-up)
    kern[k] = {"fetch_bytes": bf[k][-1] * ff, "write_bytes": bw.get(k, [0])[-1] * write_factor,
               "raw_FETCH_SIZE_KiB": bf[k][-1] / 1024, "raw_WRITE_SIZE_KiB": bw.get(k, [0])[-1] / 1024}
bench_line = json.loads(open(f"{src}/bench_FETCH_SIZE.json").read().strip().splitlines()[-1])
summary = {"calibration": {"known_bytes": known, "fetch_correction": fetch_factor, "write_correction": write_factor},
           "workload": bench_line["config"], "kernels": kern}
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
