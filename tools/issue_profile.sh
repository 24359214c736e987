#!/bin/bash
# VALU issue utilisation of sig_kernel on bench.py's C2 corpus (run on the GPU box; see tools/issue_model.py).
# Needs build/sig_bench, build/sig_bench_count, build/sig_bench_nomark, build/ubench_valu (built on the CPU box).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/issue
mkdir -p $OUT
DOCS=${1:-1000000}
timeout -k 10 300 python tools/dump_corpus.py /tmp/c2.bin $DOCS 2048 en 2 > $OUT/dump.log 2>&1
export MDD_CORPUS_FILE=/tmp/c2.bin
timeout -k 10 120 ./build/sig_bench_count > $OUT/counts.json 2> $OUT/count.err
timeout -k 10 120 ./build/sig_bench > $OUT/time.txt 2>&1
timeout -k 10 120 ./build/sig_bench_nomark > $OUT/time_nomark.txt 2>&1
timeout -k 10 300 ./build/ubench_valu --iters 5000 > $OUT/ubench.json 2> $OUT/ubench.err
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o $name -- ./build/sig_bench > $OUT/$name.log 2>&1; }
run p1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64
run p2 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU2 SQ_INSTS_SMEM GRBM_GUI_ACTIVE GRBM_COUNT
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- ./build/sig_bench > $OUT/kt.log 2>&1
find $OUT -name "*.csv" | head -20
cat $OUT/time.txt $OUT/time_nomark.txt
