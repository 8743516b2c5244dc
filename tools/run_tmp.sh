python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in 1; do
  export BZ_FUSED_REFINE=$mode
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3f_$mode -o t -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 > $R/gpurun_out/r3f_$mode.json 2> $R/gpurun_out/r3f_$mode.err
done
unset BZ_FUSED_REFINE
cd $R
python3 - <<'PY'
import csv,glob,re,json
for mode in (1,):
    print("== BZ_FUSED_REFINE=%d" % mode, json.loads(open("gpurun_out/r3f_%d.json"%mode).read().strip().splitlines()[-1])["ms_per_step"])
    for fn in glob.glob("gpurun_out/r3f_%d/**/*kernel_trace.csv"%mode, recursive=True):
        rows=[r for r in csv.DictReader(open(fn))]
        rows.sort(key=lambda r:int(r["Start_Timestamp"]))
        sel=[r for r in rows if re.search("k_group_|k_rank_place|k_survivor|k_radix_scatter_lb|k_ghist", r["Kernel_Name"])]
        half=sel[len(sel)//2:]
        for r in half:
            m=re.search(r"(k_[a-z_]+)(<[^>]*>)?", r["Kernel_Name"])
            print("  %-32s %.3f" % (m.group(0)[:32], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6))
PY
python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['step_ms'], d['checks'])"
