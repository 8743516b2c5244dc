#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/m2
timeout 900 python bench.py > gpurun_out/r03_bench_full.json 2> gpurun_out/m2/bench.err
tail -n 1 gpurun_out/r03_bench_full.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['roofline']['frac'], {k:(v.get('value') if isinstance(v,dict) else v) for k,v in d['extra'].items()}, all(d['checks'].values()), d['end_to_end'].get('bz_encode_buffer_multi'), d['t2_stress']['value'])"
