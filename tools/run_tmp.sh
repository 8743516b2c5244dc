cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2y
timeout 300 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2 3; do
timeout 200 python3 bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r2y/bench_q.json 2> gpurun_out/r2y/bench_q.err
python3 -c "
import json
line=[l for l in open('gpurun_out/r2y/bench_q.json') if l.startswith('{')][-1]
d=json.loads(line); print(d['value'], d['ms_per_step'], d['kernel_seconds_last_step_rank0'])"
done
