cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lv in 9 3; do
  timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/gpurun_out/r3dec_$lv -o t -- python3 $R/bench_decode.py --level $lv --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/r3dec_$lv.json 2> $R/gpurun_out/r3dec_$lv.err
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r3decf_$lv -o t -- python3 $R/bench_decode.py --level $lv --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
done
cd $R
python3 - <<'PY'
import csv,glob,collections
for lv in (9,3):
    agg=collections.defaultdict(float); dur=0
    for fn in glob.glob("gpurun_out/r3dec_%d/**/*counter_collection.csv"%lv, recursive=True):
        for r in csv.DictReader(open(fn)):
            if "k_dec_walk_lengths" in r["Kernel_Name"]: agg[r["Counter_Name"]]+=float(r["Counter_Value"])
    for fn in glob.glob("gpurun_out/r3decf_%d/**/*counter_collection.csv"%lv, recursive=True):
        for r in csv.DictReader(open(fn)):
            if "k_dec_walk_lengths" in r["Kernel_Name"]: agg[r["Counter_Name"]]+=float(r["Counter_Value"])
    for fn in glob.glob("gpurun_out/r3dec_%d/**/*kernel_trace.csv"%lv, recursive=True):
        for r in csv.DictReader(open(fn)):
            if "k_dec_walk_lengths" in r["Kernel_Name"]: dur+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
    print("level",lv,dict(agg),"ms",dur)
PY
