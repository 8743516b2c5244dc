cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2y
timeout 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r2y/pytest.txt 2>&1; grep -n "passed\|failed" gpurun_out/r2y/pytest.txt | tail -3
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_q
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o q -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 1 > $OUT/bench_trace.json 2> $OUT/trace.err
python3 - <<PY
import csv, glob, json
f = glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    n = r["Name"]
    if any(x in n for x in ("scatter_lb<3", "scatter_lb<6", "group_flags<true", "ghist_text", "k_pack_text", "k_block_symbols")):
        print(n[:70], r["Calls"], "%.3f" % (float(r["AverageNs"]) / 1e6))
line=[l for l in open("$OUT/bench_trace.json") if l.startswith('{')][-1]
d=json.loads(line); print(d['value'], d['ms_per_step'], d['checks']['stream_sha_equals_oracle_golden'])
PY
