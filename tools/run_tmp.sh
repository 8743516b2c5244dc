#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/m2
timeout 600 python -m pytest tests/test_gpu_deflate.py -x -q 2>&1 | tail -4 > gpurun_out/m2/tests.txt
timeout 300 python tools/df_time.py 1024 2>&1 | grep hash_chains > gpurun_out/m2/time.txt
BZ_DF_CUTS=after timeout 300 python tools/df_time.py 1024 2>&1 | grep hash_chains >> gpurun_out/m2/time.txt
cat gpurun_out/m2/tests.txt gpurun_out/m2/time.txt
