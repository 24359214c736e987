#!/bin/bash
# PMC passes over the kernels of the sorted exact-substring method on the C2 corpus (run on the GPU box).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_substr
mkdir -p $OUT
run() { name=$1; shift; timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python3 tools/substr_probe.py 1000000 sorted,sorted > $OUT/$name.log 2>&1; }
run p1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
run p2 TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_ANY SQ_WAVES
run p3 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
run p4 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_substr/*/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for name in ("emit_kernel", "scatter_kernel<1>", "scatter_kernel<2>", "leaf_kernel", "count2_kernel"):
            if name in k:
                acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    for name, d in acc.items():
        print(f.split("/")[2], name, dict(d))
PY
