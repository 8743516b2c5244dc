cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
for c in 64 128 256; do
  echo "== chunk $c MiB" >> gpurun_out/r2c/e2e.txt
  BZ_ENC_CHUNK_MIB=$c timeout 600 python3 tools/e2e_time.py 1024 >> gpurun_out/r2c/e2e.txt 2>&1
done
BZ_ENC_CHUNK_MIB=128 BZ_ENC_TRACE=1 timeout 600 python3 tools/e2e_time.py 1024 > gpurun_out/r2c/e2e_trace.txt 2>&1
cat gpurun_out/r2c/e2e.txt
timeout 1500 python3 -m pytest tests -x -q -m gpu > gpurun_out/r2c/pytest.txt 2>&1
tail -5 gpurun_out/r2c/pytest.txt
