#!/bin/bash
# PMC passes for sig_kernel (run on the GPU box).  One rocprofv3 run per pass, counters only.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_sig
mkdir -p $OUT
ARGS="1000000 2048 60"
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o $name -- ./build/sig_bench $ARGS > $OUT/$name.log 2>&1; }
run p1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU
run p2 SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_I8 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE GRBM_COUNT
run p3 FETCH_SIZE
run p4 WRITE_SIZE
ls -R $OUT | head -40
