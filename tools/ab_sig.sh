# development A/B of sig_kernel: waves per SIMD (binaries built from tools/sig_bench.hip with -DMDD_WPE=n)
set -e
for args in "400000 2048 60 128 0" "400000 2048 60 256 0" "40000 20480 60 128 0" "400000 2048 60 128 1" "400000 2048 60 128 0 1 5" "400000 2048 60 128 0 0 20" "400000 300 60 128 0"; do
  echo "== $args"
  for w in 3 4 5 6; do echo -n "wpe=$w "; timeout -k 10 60 ./build/sig_bench_w$w $args | cut -c95-200; done
done
