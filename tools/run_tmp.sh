#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/m2
: > gpurun_out/m2/fuzz_long.txt
for fl in deflate small stream deflate big dstream deflate; do
  timeout 200 python tools/fuzz_parity.py 150 $((RANDOM % 9000 + 1000)) $fl 2>&1 | tail -n 1 >> gpurun_out/m2/fuzz_long.txt
done
BZ_DF_MATCH=walk BZ_DF_PARSE=doubling timeout 200 python tools/fuzz_parity.py 100 4242 deflate 2>&1 | tail -n 1 >> gpurun_out/m2/fuzz_long.txt
cat gpurun_out/m2/fuzz_long.txt
