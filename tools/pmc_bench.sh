#!/bin/bash
# HBM traffic of the bench's kernels from PMC counters (run on the GPU box): one rocprofv3
# run per counter, counters only; plus calibration kernels with known byte counts.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_bench
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 60 rocprofv3 --pmc $c --output-format csv -d $OUT/cal_$c -o cal -- ./build/pmc_cal > $OUT/cal_$c.log 2>&1
  timeout -s KILL 400 rocprofv3 --pmc $c --kernel-include-regex "sig_kernel|band_keys|bucket_|uf_|removed_kernel|seg_plan|win_kernel|compact|para_|exact_|norm_en|emit_kernel|scatter_kernel|leaf_kernel|count2_kernel|anchor_|vmap_kernel|smear_kernel|scan_" --output-format csv -d $OUT/bench_$c -o bench -- python3 bench.py --no-cpu --no-pcie --steps 1 --warmup 1 > $OUT/bench_$c.json 2> $OUT/bench_$c.log
done
ls -R $OUT | head -30
