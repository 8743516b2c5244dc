cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2y
for t in 1 2 3 4 6; do
BZ_TAIL_SLICES=$t timeout 200 python3 bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r2y/bench_q.json 2> gpurun_out/r2y/bench_q.err
python3 -c "
import json
line=[l for l in open('gpurun_out/r2y/bench_q.json') if l.startswith('{')][-1]
d=json.loads(line); print('slices $t', d['value'], d['ms_per_step'], d['checks']['stream_sha_equals_oracle_golden'], d['kernel_seconds_last_step_rank0'])"
done
