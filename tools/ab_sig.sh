# development A/B of sig_kernel builds (tools/sig_bench.hip): prev = the committed tree, new = the working tree
set -e
for args in "400000 2048 60 128 0" "1000000 2048 60 128 0" "100000 2048 60 128 0" "400000 2048 100 128 0" "400000 2048 60 256 0" "40000 20480 60 128 0" "400000 2048 60 128 1" "400000 2048 60 128 0 1 5"; do
  echo "== $args"
  for m in _prev ""; do echo -n "new$m "; timeout -k 10 60 ./build/sig_bench$m $args | cut -c95-160; done
done
