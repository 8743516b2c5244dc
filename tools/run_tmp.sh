cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_deflate.py -x -q -m gpu -k "seams or two_gib or parts" 2>&1 | tail -8
