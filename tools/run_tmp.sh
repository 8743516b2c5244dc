cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2k
timeout 600 python3 tools/e2e_time.py 1024 2>&1 | grep -v amdgpu > gpurun_out/r2k/e2e.txt; cat gpurun_out/r2k/e2e.txt
timeout 1800 python3 -m pytest tests -x -q -m gpu > gpurun_out/r2k/pytest.txt 2>&1
tail -4 gpurun_out/r2k/pytest.txt
