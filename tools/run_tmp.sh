cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r2d/pytest.txt 2>&1
tail -5 gpurun_out/r2d/pytest.txt
timeout 900 python3 bench.py --steps 3 --warmup 1 --no-extras > gpurun_out/r2d/bench1.json 2> gpurun_out/r2d/bench1.err
tail -3 gpurun_out/r2d/bench1.err
python3 -c "
import json
d=json.load(open('gpurun_out/r2d/bench1.json'))
print(d['value'], d['ms_per_step'], d['kernel_seconds_last_step_rank0'], d['checks'])
print({k:round(v['ms']/d['steps'],2) for k,v in d['kernels'].items()})"
( time python3 -c "
import importlib,time
t=time.time(); pkg=importlib.import_module('rust-compression_amd'); a=pkg.compress(b'hello world'*10, 9); print('first compress', time.time()-t)
t=time.time(); a=pkg.compress(b'hello world'*10, 9); print('second compress', time.time()-t)" ) 2>&1 | tail -6
