cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
for v in zle3 zle4; do
OUT=$R/gpurun_out/prof_$v
mkdir -p $OUT
cp rust-compression_amd/libbz2_mi355x.so /tmp/lib_default.so
cp rust-compression_amd/build/var/$v.so rust-compression_amd/libbz2_mi355x.so
(cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o q -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > $OUT/bench_trace.json 2> $OUT/trace.err)
cp /tmp/lib_default.so rust-compression_amd/libbz2_mi355x.so
echo "== $v"; grep "k_zle_emit" $OUT/trace/q_kernel_stats.csv | cut -d, -f1-4
done
