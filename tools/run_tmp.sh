cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2q
timeout 1800 python3 -m pytest tests -x -q -m gpu > gpurun_out/r2q/pytest.txt 2>&1
tail -4 gpurun_out/r2q/pytest.txt
timeout 900 python3 bench.py --steps 5 --warmup 2 > gpurun_out/r2q/bench.json 2> gpurun_out/r2q/bench.err; tail -3 gpurun_out/r2q/bench.err
python3 -c "
import json
d=json.load(open('gpurun_out/r2q/bench.json'))
for k in ('value','ms_per_step','checks','end_to_end','t2_stress'): print(k, d.get(k))"
