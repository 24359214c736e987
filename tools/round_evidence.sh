#!/bin/bash
# End-of-round evidence on the GPU box: default bench line, the same command under rocprofv3 --kernel-trace --stats, unit rates,
# the other BASELINE configurations, the GPU test suite.  Output under gpurun_out/$ROUND_TAG/ (default r3) (copied into profiles/ by hand).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${ROUND_TAG:-r3}; mkdir -p $O
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py --no-cpu --no-pcie > $O/bench_under_rocprof.json 2> $O/kt.err
python3 tools/kstats.py $O/kt 60 > $O/kernel_stats.txt
timeout -k 10 300 python tools/unit_rates.py > $O/unit_rates.txt 2>&1
timeout -k 10 600 python tools/bench_configs.py > $O/bench_configs.jsonl 2> $O/bench_configs.err
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/tests.log 2>&1
