cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2y
timeout 200 python3 tools/e2e_time.py 1024 2>&1 | grep -v amdgpu.ids | tail -5
timeout 600 python3 -m pytest tests -x -q -m gpu > gpurun_out/r2y/pytest.txt 2>&1; grep -n "passed\|failed" gpurun_out/r2y/pytest.txt | tail -2
timeout 500 python3 bench.py > gpurun_out/bench_r02_final.json 2> gpurun_out/bench_r02_final.err
python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/bench_r02_final.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['end_to_end'], all(d['checks'].values()))"
