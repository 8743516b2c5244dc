cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2t
timeout 1200 python3 -m pytest tests/test_gpu_sharded.py -x -q -m gpu > gpurun_out/r2t/pytest.txt 2>&1
tail -12 gpurun_out/r2t/pytest.txt
timeout 600 python3 bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline --force-sharded --transport rccl --mib-per-gpu 256 > gpurun_out/r2t/b_rccl1.json 2> gpurun_out/r2t/b_rccl1.err; tail -3 gpurun_out/r2t/b_rccl1.err; tail -c 400 gpurun_out/r2t/b_rccl1.json
