cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2y
timeout 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r2y/pytest.txt 2>&1; grep -n "passed\|failed" gpurun_out/r2y/pytest.txt | tail -3
