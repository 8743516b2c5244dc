# scratch: the command file of the last gpurun call (kept so that a call can be repeated)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2y
timeout 600 python3 -m pytest tests -x -q -m gpu > gpurun_out/r2y/pytest.txt 2>&1; grep "passed\|failed" gpurun_out/r2y/pytest.txt | tail -2
timeout 300 python3 bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-300
