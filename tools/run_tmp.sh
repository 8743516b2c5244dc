cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2n
show() { python3 -c "
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], d['value'], d['ms_per_step'], d['kernel_seconds_last_step_rank0'], d['checks'])
print({k:round(v['ms']/d['steps'],2) for k,v in d['kernels'].items()})" $1; }
timeout 600 python3 bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r2n/b_def.json 2> gpurun_out/r2n/b_def.err; show gpurun_out/r2n/b_def.json
timeout 600 tools/variant_run.sh rust-compression_amd/build/var/kt.so python3 bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r2n/b_kt.json 2> gpurun_out/r2n/b_kt.err; show gpurun_out/r2n/b_kt.json
timeout 1800 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r2n/pytest.txt 2>&1
tail -4 gpurun_out/r2n/pytest.txt
BZ_LASTCOL_PASS=1 timeout 600 python3 bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline --mib-per-gpu 128 > gpurun_out/r2n/b_lastcol.json 2> gpurun_out/r2n/b_lastcol.err; show gpurun_out/r2n/b_lastcol.json
