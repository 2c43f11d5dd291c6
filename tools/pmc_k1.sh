# PMC pass for k_demod_bits (own run: counters only, no traces)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$1
rm -rf $OUT
RD_K1_DEBUG=${2:-0} rocprofv3 --pmc $3 --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/step_profile.py 4096 notiming > $OUT.log 2>&1
python3 - <<PY
import csv, glob, collections
files = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        if 'k_demod_bits' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    print(f"$1 {k:28s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
PY
