#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/m2
cp rust-compression_amd/libbz2_mi355x.so /tmp/lib_default.so
cp rust-compression_amd/build/var/m2sc.so rust-compression_amd/libbz2_mi355x.so
timeout 120 python tools/df_time.py 1024 2>&1 | grep hash_chains > gpurun_out/m2/sc.txt
timeout 300 bash tools/df_prof.sh 256 "FETCH_SIZE" m2scf 2>&1 | grep k_df_match2 >> gpurun_out/m2/sc.txt
timeout 300 bash tools/df_prof.sh 256 "WRITE_SIZE" m2scw 2>&1 | grep k_df_match2 >> gpurun_out/m2/sc.txt
cp /tmp/lib_default.so rust-compression_amd/libbz2_mi355x.so
timeout 300 bash tools/df_prof.sh 256 "WRITE_SIZE" m2scw0 2>&1 | grep k_df_match2 >> gpurun_out/m2/sc.txt
cat gpurun_out/m2/sc.txt
