for v in ds64 ds32; do
for w in 256 512; do
echo "--- $v walk_wgs $w"
BZ_DEC_WALK_WGS=$w bash tools/variant_run.sh rust-compression_amd/build/var/$v.so timeout 200 python bench_decode.py --level 9 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('stages_s'), (d.get('roofline') or {}).get('avg_launch_ms'), d.get('checks'))"
done
done
