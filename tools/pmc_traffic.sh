# HBM traffic of k_demod_bits from PMC counters, separate passes (MI355X_MICROARCH.md, HBM section)
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$C
  rm -rf $OUT
  rocprofv3 --pmc $C --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT.log 2>&1
done
python3 - <<PY
import csv, glob, collections, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/pmc_*SIZE/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("kernel,counter,mean_value,n")
for k, d in agg.items():
    for c, v in d.items():
        print(f"{k},{c},{sum(v)/len(v):.1f},{len(v)}")
PY
