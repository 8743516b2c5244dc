#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/m2
BENCH=bench_deflate.py timeout 1200 bash tools/profile.sh r03df > gpurun_out/m2/prof_df.txt 2>&1
timeout 600 python bench_deflate.py > gpurun_out/r03_bench_deflate_1gib.json 2> gpurun_out/m2/bench_df.err
timeout 900 python bench.py > gpurun_out/r03_bench_full.json 2> gpurun_out/m2/bench.err
tail -n 1 gpurun_out/r03_bench_deflate_1gib.json | cut -c1-200
tail -n 1 gpurun_out/r03_bench_full.json | cut -c1-200
