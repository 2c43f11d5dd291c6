#!/bin/bash
# Memory-side counters of the demod kernel and of its loads-only / loads+stores ablations (diagnostic library):
# where do the 6 % of output bytes cost their time, and what does the load path look like from L1 / L2 / EA?
# One rocprofv3 run per counter group (never together with a trace); python3 straight after `--`.
#   usage: bash tools/pmc_memside.sh <outdir> [variants...]     variant = name:RD_K1_DEBUG:RD_K1_STAMPS
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${1:-$ROOT/gpurun_out/pmc_memside}; shift
VARS=${@:-"full:0:2 loads_stores:2:2 loads_only:6:2"}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GPU_FORCE_BLIT_COPY_SIZE=0
export RTLDAVIS_HIP_LIB=$ROOT/rtldavis_amd/librtldavis_hip_diag.so
declare -A G
G[ea_latency]="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum"
G[ea_stalls]="TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum"
G[l2_hits]="TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum"
G[l2_busy]="TCC_BUSY_sum TCC_CYCLE_sum TCC_WRITEBACK_sum TCC_EA0_WRREQ_64B_sum"
G[l1_latency]="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
# (a TA_* group - TA_BUSY, TA_ADDR_STALLED_BY_TC_CYCLES, TA_DATA_STALLED_BY_TC_CYCLES, TA_FLAT_READ_LDS_WAVEFRONTS - made
# rocprofv3 abort with signal 6 on this pool: not collected)
G[utcl1]="TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_STALL_MULTI_MISS_sum"
G[sq]="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS"
for V in $VARS; do
  IFS=: read NAME DBG OPT <<< "$V"
  export RD_K1_DEBUG=$DBG RD_K1_STAMPS=$OPT RD_TAIL_IMPL=legacy
  for K in ea_latency ea_stalls l2_hits l2_busy l1_latency utcl1 sq; do
    D=$OUT/${NAME}_$K
    rm -rf $D
    timeout -k 10 180 rocprofv3 --pmc ${G[$K]} --output-format csv -d $D -- python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-verify --sustain 0 --settle 0 --live-traffic 0 > $D.log 2>&1
    RC=$?
    echo "[pmc_memside] $NAME $K rc $RC" | tee -a $OUT/progress.log
    if [ $RC -ge 124 ]; then exit $RC; fi
  done
done
python3 - $OUT <<'PY' | tee $OUT/summary.csv
import csv, glob, collections, os, sys
root = sys.argv[1]
print("variant,counter,mean_per_dispatch,n")
for d in sorted(glob.glob(root + "/*_*")):
    if not os.path.isdir(d):
        continue
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_demod_mfma" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    name = os.path.basename(d)
    for c, v in sorted(agg.items()):
        v = v[len(v) // 3:]  # the first dispatches run cold
        print(f"{name},{c},{sum(v)/len(v):.1f},{len(v)}")
PY
