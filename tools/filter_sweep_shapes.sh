for ff in 64 128 192 256 384 512 1024; do echo -n "cjk ff=$ff: "; timeout -k 10 60 ./build/sig_bench 1000000 2048 60 128 1 0 13 16384 $ff 2>&1 | grep -o "ms=[0-9.]*  Gsh/s=[0-9.]*"; done
for ff in 64 128 256 512; do echo -n "word ff=$ff: "; timeout -k 10 60 ./build/sig_bench 1000000 2048 60 128 0 1 5 16384 $ff 2>&1 | grep -o "ms=[0-9.]*  Gsh/s=[0-9.]*"; done
for ff in 256 512 768; do echo -n "ascii ff=$ff: "; timeout -k 10 60 ./build/sig_bench 1000000 2048 60 128 0 0 13 16384 $ff 2>&1 | grep -o "ms=[0-9.]*  Gsh/s=[0-9.]*"; done
