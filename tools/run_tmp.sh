#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/m2
timeout 200 tools/variant_run.sh rust-compression_amd/build/var/ratom.so python tools/df_time.py 1024 2>&1 | grep hash_chains > gpurun_out/m2/ratom.txt
cp rust-compression_amd/libbz2_mi355x.so /tmp/lib_default.so
cp rust-compression_amd/build/var/ratom.so rust-compression_amd/libbz2_mi355x.so
timeout 300 python -m pytest tests/test_gpu_deflate.py -x -q -k "seeded or sort_chunk or big_corpus" 2>&1 | tail -2 >> gpurun_out/m2/ratom.txt
cp /tmp/lib_default.so rust-compression_amd/libbz2_mi355x.so
cat gpurun_out/m2/ratom.txt
