# development A/B of sig_kernel builds (tools/sig_bench.hip): old = -DMDD_SHA_ORDER_OLD, new = default, new5 = -DMDD_WPE=5
set -e
for args in "400000 2048 60 128 0" "400000 2048 60 256 0" "40000 20480 60 128 0" "400000 2048 60 128 1" "400000 2048 60 128 0 1 5" "400000 2048 60 128 0 0 20"; do
  echo "== $args"
  for m in old new new5; do echo -n "$m "; timeout -k 10 60 ./build/sig_bench_$m $args | cut -c95-160; done
done
