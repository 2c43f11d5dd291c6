# SQ counters of the fused demod kernel under the bench load (own run: counters only).
# usage: pmc_k1.sh [tag]   (environment such as RD_K1_IMPL / RD_K1_DEBUG is passed through)
cd /tmp && export TMPDIR=/tmp
TAG=${1:-sq}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
rm -rf $OUT
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-verify --sustain 0 --live-traffic 0 > $OUT.log 2>&1
python3 - <<PY
import csv, glob, collections, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_$TAG"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("kernel,counter,mean_value,n")
for k, d in agg.items():
    if "rocclr" in k: continue
    for c, v in sorted(d.items()):
        print(f"{k},{c},{sum(v)/len(v):.1f},{len(v)}")
PY
