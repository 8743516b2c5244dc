#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun): kernel-trace stats, then PMC passes.
# Usage: [BENCH=bench_decode.py] tools/profile.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
R=${GRAFT_REPO_ROOT:-/root/repo}
BENCH=${BENCH:-bench.py}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o $TAG -- python3 $R/$BENCH --no-cpu-baseline --no-extras "$@" > $OUT/bench_trace.json 2> $OUT/trace.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o $TAG -- python3 $R/$BENCH --no-cpu-baseline --no-extras --steps 1 --warmup 0 > $OUT/bench_fetch.json 2> $OUT/fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o $TAG -- python3 $R/$BENCH --no-cpu-baseline --no-extras --steps 1 --warmup 0 > $OUT/bench_write.json 2> $OUT/write.err
ls -R $OUT | head -40
