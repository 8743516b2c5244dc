#!/bin/bash
# PMC passes over the Deflate bench (run on the GPU box): tools/df_prof.sh <mib> "<counters>" <tag>
set -u
MIB=${1:-64}; CTRS=${2:-"SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"}; TAG=${3:-dfpmc}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT -o $TAG -- python3 $R/bench_deflate.py --no-cpu-baseline --steps 1 --warmup 0 --mib $MIB > $OUT/bench.json 2> $OUT/err.txt
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for fn in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "k_df" in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("(")[0].split("::")[-1]
            agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    print(k, dict(v))
PY
