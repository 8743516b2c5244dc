#!/bin/bash
# PMC counters of the sort kernels (run on the GPU box through gpurun): which pipe is full in k_radix_scatter_lb?
#   tools/scatter_counters.sh [mib] [tag]
# One rocprofv3 pass per counter set (SQ: 8 slots, TCC: 4), the program itself behind "--" (no wrapper: the profiler's
# preloaded library initialises the GPU before the program starts), --kernel-trace only.  Output: gpurun_out/<tag>/summary.md
set -u
MIB=${1:-256}; TAG=${2:-r04_scatter}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_available.txt 2>&1
PASSES=(
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
 "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_WR"
 "TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_WRITE_sum"
 "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"
 "GRBM_GUI_ACTIVE GRBM_COUNT"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pass$i -o p$i -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 0 --mib-per-gpu $MIB > $OUT/pass$i.json 2> $OUT/pass$i.err
  echo "pass $i: rc $? ($P)" >> $OUT/passes.txt
done
python3 - <<PY
import csv, glob, collections, re, os
out = "$OUT"
agg = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float); calls = collections.defaultdict(int); regs = {}
for fn in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
        if m:
            agg[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
            regs[m.group(1)] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Workgroup_Size"))
seen = False
for fn in sorted(glob.glob(out + "/pass1/**/*kernel_trace.csv", recursive=True)):
    for r in csv.DictReader(open(fn)):
        m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
        if m:
            dur[m.group(1)] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            calls[m.group(1)] += 1
with open(out + "/summary.md", "w") as f:
    for k in ("k_radix_scatter_lb", "k_group_refine", "k_rank_place", "k_ghist_text", "k_mtf_ranks_small", "k_zle_fused"):
        if k not in agg: continue
        f.write("## %s  (%d launches, %.3f ms under the profiler; VGPR/AGPR/SGPR/LDS/WG = %s)\n\n| counter | total | per launch |\n|---|---|---|\n" % (k, calls[k], dur[k], regs.get(k)))
        for c, v in sorted(agg[k].items()):
            f.write("| %s | %.4g | %.4g |\n" % (c, v, v / max(calls[k], 1)))
        f.write("\n")
print(open(out + "/summary.md").read()[:6000])
PY
