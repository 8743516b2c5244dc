timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3k -o t -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 --hang-timeout 100 > $R/gpurun_out/r3k.json 2> $R/gpurun_out/r3k.err
cd $R
python3 - <<'PY'
import csv,glob,re,json
for fn in glob.glob("gpurun_out/r3k/**/*kernel_trace.csv", recursive=True):
    rows=[r for r in csv.DictReader(open(fn))]
    rows.sort(key=lambda r:int(r["Start_Timestamp"]))
    sel=[r for r in rows if re.search("k_huff|k_emit_payload", r["Kernel_Name"])]
    half=sel[len(sel)//2:]
    for r in half:
        m=re.search(r"(k_[a-z_0-9]+)(<[^>]*>)?", r["Kernel_Name"])
        print("  %-32s %.3f" % (m.group(0)[:32], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6))
PY
for m in 1; do
BZ_HUFF_SPLIT=$m timeout 300 python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline --hang-timeout 100 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split=$m', d['value'], d['ms_per_step'], d['step_ms'], d['checks'], d['kernel_seconds_last_step_rank0'])"
done
