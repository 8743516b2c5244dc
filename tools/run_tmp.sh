cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2x
timeout 600 python3 bench.py --gpus 2 --share-gpu --mib-per-gpu 256 --steps 2 --warmup 1 > gpurun_out/r2x/bench2.json 2> gpurun_out/r2x/bench2.err
python3 -c "
import json
line=[l for l in open('gpurun_out/r2x/bench2.json') if l.startswith('{')][-1]
d=json.loads(line); print(d['value'], d['n_gpus'], d['checks'], (d.get('extra') or {}).get('decode',{}).get('value'))"
timeout 600 python3 bench.py --gpus 3 --share-gpu --mib-per-gpu 128 --steps 1 --warmup 1 --transport rccl > gpurun_out/r2x/bench3.json 2> gpurun_out/r2x/bench3.err; tail -2 gpurun_out/r2x/bench3.err
python3 -c "
import json
line=[l for l in open('gpurun_out/r2x/bench3.json') if l.startswith('{')][-1]
d=json.loads(line); print(d['value'], d['n_gpus'], d['checks'], (d.get('extra') or {}).get('decode',{}).get('value'))"
