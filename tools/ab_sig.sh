# development A/B of sig_kernel's filter variants: binaries built from tools/sig_bench.hip with
# -DMDD_FG=<hashes per group> -DMDD_FV=<compare style> (off = -DMDD_FILTER_FROM=0xFFFFFFFFu)
set -e
for args in "400000 2048 60 128 0" "400000 2048 60 256 0" "40000 20480 60 128 0" "4000 204800 100 128 0"; do
  echo "== $args"
  for m in off g4v2 g8v2 g8t512 g8t2048 g16t1024 g16t2048; do echo -n "$m "; timeout -k 10 60 ./build/sig_bench_$m $args | cut -c95-200; done
done
