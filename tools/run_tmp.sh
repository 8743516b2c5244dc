cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2y
(timeout 150 python3 tools/fuzz_parity.py 90 21 small 2>&1 | tail -2) 
(timeout 150 python3 tools/fuzz_parity.py 90 22 big 2>&1 | tail -2)
(timeout 150 python3 tools/fuzz_parity.py 60 23 stream 2>&1 | tail -2)
(timeout 150 python3 tools/fuzz_parity.py 60 24 deflate 2>&1 | tail -2)
(BZ_LOCAL_B=1 timeout 150 python3 tools/fuzz_parity.py 60 25 big 2>&1 | tail -2)
