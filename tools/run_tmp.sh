cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2v
timeout 1800 python3 -m pytest tests/test_gpu_deflate.py -x -q -m gpu > gpurun_out/r2v/pytest.txt 2>&1
tail -25 gpurun_out/r2v/pytest.txt
