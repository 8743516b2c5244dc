#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/m2
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/m2/tests_all.txt
timeout 300 python tools/fuzz_parity.py 150 777 deflate 2>&1 | tail -n 1 >> gpurun_out/m2/tests_all.txt
BENCH=bench_deflate.py timeout 1200 bash tools/profile.sh r03df > gpurun_out/m2/prof_df.txt 2>&1
timeout 600 python bench_deflate.py > gpurun_out/r03_bench_deflate_1gib.json 2> gpurun_out/m2/bench_df.err
cat gpurun_out/m2/tests_all.txt; tail -n 1 gpurun_out/r03_bench_deflate_1gib.json | cut -c1-200
