# scalar-side SQ counters of the demod kernel under the bench load (own run: counters only)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_salu
rm -rf $OUT
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/a -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-verify --sustain 0 --live-traffic 0 > $OUT.a.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_IFETCH SQ_IFETCH_LEVEL --output-format csv -d $OUT/b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-verify --sustain 0 --live-traffic 0 > $OUT.b.log 2>&1
python3 - <<PY
import csv, glob, collections, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_salu"
agg = collections.defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_demod" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(agg.items()):
    print(f"k_demod_mfma,{c},{sum(v)/len(v):.1f},{len(v)}")
PY
tail -3 $OUT.b.log
