cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/decpmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $OUT -o d -- python3 $R/bench_decode.py --steps 1 --warmup 0 > $OUT/bench.json 2> $OUT/err.txt
tail -3 $OUT/err.txt
python3 - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for fn in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
        if m: agg[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    if 'walk' in k or 'tscatter' in k or 'seg_copy' in k: print(k, {a: "%.4g" % b for a, b in v.items()})
PY
