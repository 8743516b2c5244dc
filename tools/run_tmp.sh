cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2o
show() { python3 -c "
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], d['value'], d['ms_per_step'], d['kernel_seconds_last_step_rank0'], d['checks'])
print({k:round(v['ms']/d['steps'],2) for k,v in d['kernels'].items()})" $1; }
timeout 600 python3 bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r2o/b_def.json 2> gpurun_out/r2o/b_def.err; show gpurun_out/r2o/b_def.json
timeout 1800 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py -x -q -m gpu > gpurun_out/r2o/pytest.txt 2>&1
tail -4 gpurun_out/r2o/pytest.txt
timeout 900 python3 bench.py --gpus 2 --share-gpu --mib-per-gpu 512 --steps 2 --warmup 1 > gpurun_out/r2o/bench2.json 2> gpurun_out/r2o/bench2.err
tail -2 gpurun_out/r2o/bench2.err; tail -c 600 gpurun_out/r2o/bench2.json
