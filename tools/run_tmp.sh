bash tools/profile.sh r03 --steps 5 --warmup 2 --hang-timeout 200 2>&1 | tail -5
timeout 600 python bench.py --steps 20 --warmup 5 --hang-timeout 200 > gpurun_out/r03_bench_full.json 2> gpurun_out/r03_bench_full.err; tail -c 600 gpurun_out/r03_bench_full.json
