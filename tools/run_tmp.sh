cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2j
timeout 600 python3 tools/e2e_time.py 1024 2>&1 | grep -v amdgpu > gpurun_out/r2j/e2e.txt; cat gpurun_out/r2j/e2e.txt
timeout 900 python3 bench.py --steps 5 --warmup 2 > gpurun_out/r2j/bench.json 2> gpurun_out/r2j/bench.err; tail -3 gpurun_out/r2j/bench.err
python3 -c "
import json
d=json.load(open('gpurun_out/r2j/bench.json'))
for k in ('value','ms_per_step','checks','end_to_end','t2_stress','cpu_baseline','cpu_baseline_all_cores','roofline'): print(k, d.get(k))
print({k:(v['value'],v['ms_per_step'],v['roofline'],v.get('cpu_baseline')) for k,v in d['extra'].items()})"
