cd $GRAFT_REPO_ROOT
(timeout 220 python3 tools/fuzz_parity.py 180 31 small 2>&1 | tail -1)
(timeout 220 python3 tools/fuzz_parity.py 180 32 big 2>&1 | tail -1)
(timeout 120 python3 tools/fuzz_parity.py 80 33 dstream 2>&1 | tail -1)
(timeout 120 python3 tools/fuzz_parity.py 80 34 stream 2>&1 | tail -1)
(BZ_DF_PART_MIB=1 timeout 150 python3 tools/fuzz_parity.py 100 35 deflate 2>&1 | tail -1)
