#!/bin/bash
# PMC pass over bench.py (run on the GPU box): tools/bz_pmc.sh <mib> "<counters>" <tag>
set -u
MIB=${1:-256}; CTRS=${2:-"SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES"}; TAG=${3:-bzpmc}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT -o $TAG -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 0 --mib-per-gpu $MIB > $OUT/bench.json 2> $OUT/err.txt
python3 - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float)
for fn in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
        if m: agg[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
for fn in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
        if m: dur[m.group(1)] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for k, v in sorted(agg.items(), key=lambda kv: -dur[kv[0]])[:14]:
    print("%-22s %7.2f ms " % (k, dur[k]), {a: "%.3g" % b for a, b in v.items()})
PY
