#!/bin/bash
# A/B of sig_bench builds on the C2 corpus.  build/sig_bench_head = tools/sig_bench.hip of the commit to compare with, built in a
# scratch worktree (git worktree add /tmp/w <commit>; hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include -o build/sig_bench_head tools/sig_bench.hip)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/ab; mkdir -p $OUT
timeout -k 10 300 python tools/dump_corpus.py /tmp/c2.bin 1000000 2048 en 2 > $OUT/dump.log 2>&1
export MDD_CORPUS_FILE=/tmp/c2.bin
for A in "1000000 2048 60 128 0 0 13" "1000000 2048 60 128 0 1 5"; do
for r in 1 2; do
for b in ${BUILDS:-sig_bench_head sig_bench}; do echo -n "$b: "; timeout -k 10 120 ./build/$b $A 2>&1 | grep "Gsh/s"; done
done
done
