cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
show() { python3 -c "
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], d['value'], d['ms_per_step'], d['kernel_seconds_last_step_rank0'], d['checks'])" $1; }
timeout 600 python3 bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r2e/b_def.json 2> gpurun_out/r2e/b_def.err; show gpurun_out/r2e/b_def.json
for v in huff256 huff128; do
timeout 600 tools/variant_run.sh rust-compression_amd/build/var/$v.so python3 bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r2e/b_$v.json 2> gpurun_out/r2e/b_$v.err; show gpurun_out/r2e/b_$v.json
done
timeout 600 tools/variant_run.sh rust-compression_amd/build/var/hufftime.so python3 bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline > gpurun_out/r2e/b_time.json 2> gpurun_out/r2e/b_time.err; grep cycles gpurun_out/r2e/b_time.json gpurun_out/r2e/b_time.err | head -3
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r2e/pytest.txt 2>&1
tail -3 gpurun_out/r2e/pytest.txt
