cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/m2
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/m2/parity.log 2>&1; tail -2 gpurun_out/m2/parity.log
timeout 200 python tools/fuzz_parity.py 60 4001 small 2>&1 | tail -1
timeout 200 python tools/fuzz_sharded.py 60 4002 2>&1 | tail -1
for i in 1 2 3; do
timeout 300 python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/m2/new$i.json 2>/dev/null
tools/variant_run.sh rust-compression_amd/build/var/base.so timeout 300 python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/m2/base$i.json 2>/dev/null
done
python - <<'PY'
import json
for f in ['new1','base1','new2','base2','new3','base3']:
    d=json.load(open('gpurun_out/m2/%s.json'%f))
    k=d['kernels']
    print(f, d['value'], d['step_ms'], d['kernel_seconds_last_step_rank0']['rle1_crc_split'], d['kernel_seconds_last_step_rank0']['bwt'], k['k_radix_scatter_lb']['ms']/k['k_radix_scatter_lb']['launches'])
PY
