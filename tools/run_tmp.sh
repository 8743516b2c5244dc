cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
timeout 400 bash tools/profile.sh r02 > /dev/null 2>&1
BENCH=bench_decode.py timeout 400 bash tools/profile.sh r02dec > /dev/null 2>&1
BENCH=bench_deflate.py timeout 400 bash tools/profile.sh r02df > /dev/null 2>&1
cd $R
timeout 500 python3 bench.py > gpurun_out/bench_r02_final.json 2> gpurun_out/bench_r02_final.err
timeout 300 python3 bench_decode.py > gpurun_out/bench_decode_r02.json 2> gpurun_out/bench_decode_r02.err
timeout 300 python3 bench_deflate.py > gpurun_out/bench_deflate_r02.json 2> gpurun_out/bench_deflate_r02.err
tail -c 600 gpurun_out/bench_r02_final.json | head -c 600; echo
ls gpurun_out/prof_r02 gpurun_out/prof_r02dec gpurun_out/prof_r02df
