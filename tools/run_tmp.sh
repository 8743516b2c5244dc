cd $GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -6
timeout 600 python3 bench.py --steps 5 --warmup 2 > gpurun_out/r2y/bench_full.json 2> gpurun_out/r2y/bench_full.err; tail -2 gpurun_out/r2y/bench_full.err
python3 -c "
import json
line=[l for l in open('gpurun_out/r2y/bench_full.json') if l.startswith('{')][-1]
d=json.loads(line); print(d['value'], d['ms_per_step'], d['checks']); print({k:(v.get('value') if isinstance(v,dict) else v) for k,v in d.get('extra',{}).items()})"
