#!/bin/bash
# PMC passes for sig_kernel on the word-5-gram and CJK shapes (run on the GPU box).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_word
mkdir -p $OUT
run() { name=$1; args=$2; shift; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o $name -- ./build/sig_bench $args > $OUT/$name.log 2>&1; }
for cfg in "word:400000 2048 60 128 0 1 5" "cjk:400000 2048 60 128 1 0 13" "ascii:400000 2048 60 128 0 0 13"; do
  n=${cfg%%:*}; a=${cfg#*:}
  run ${n}_p1 "$a" SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU
  run ${n}_p2 "$a" SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_word/*/*counter_collection.csv") + glob.glob("gpurun_out/pmc_word/*/*/*counter_collection.csv")):
    acc = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "sig_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
    print(f.split("/")[2], dict(acc))
PY
