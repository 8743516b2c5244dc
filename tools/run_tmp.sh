timeout 1600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > /tmp/o.txt 2>&1; echo "parity rc=$? $(tail -1 /tmp/o.txt)"; grep -v "^tests\|^$\|^\.\|passed" /tmp/o.txt | head -40 | cut -c1-220
timeout 300 python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline --hang-timeout 100 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['step_ms'], d['checks'], d['kernel_seconds_last_step_rank0']); print({k:(v['launches'],round(v['ms']/10,2)) for k,v in d['kernels'].items()})"
