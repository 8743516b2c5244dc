#!/bin/bash
# PMC pass over one decode (run on the GPU box): tools/dec_pmc.sh <tag> <walk wgs> "<counters>"
set -u
TAG=$1; WGS=$2; CTRS=${3:-"TCC_HIT_sum TCC_MISS_sum"}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/decpmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT -o $TAG -- python3 $R/tools/dec_variants.py $TAG 1024 $WGS > $OUT/out.txt 2> $OUT/err.txt
python3 - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for fn in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        m = re.search(r"(k_dec_[a-z0-9_]+)", r["Kernel_Name"])
        if m:
            agg[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[m.group(1)] += 1
for k, v in sorted(agg.items()):
    if k in ("k_dec_walk_lengths", "k_dec_seg_copy", "k_dec_rank_samples", "k_dec_tscatter"):
        nl = cnt[k] / max(len(v), 1)
        print("$TAG wgs $WGS %-22s launches %d per launch:" % (k, nl), {a: "%.4g" % (b / nl) for a, b in v.items()})
PY
