cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_mix
rm -rf $OUT
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT -- $GRAFT_REPO_ROOT/tools/ubench/valu_mix > $OUT.log 2>&1
python3 - <<PY
import csv, glob, collections
cc = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)
kt = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)
dur = {}
for f in kt:
    for r in csv.DictReader(open(f)):
        dur[r['Dispatch_Id']] = (int(r['End_Timestamp'])-int(r['Start_Timestamp']), r['Kernel_Name'], r['Grid_Size_X'])
rows = collections.defaultdict(dict)
for f in cc:
    for r in csv.DictReader(open(f)):
        rows[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
        rows[r['Dispatch_Id']]['name'] = r['Kernel_Name']; rows[r['Dispatch_Id']]['grid'] = r['Grid_Size_X']
seen = {}
for d, v in rows.items():
    key = (v['name'], v['grid'])
    seen[key] = (v, dur.get(d, (0,))[0])
for (name, grid), (v, ns) in seen.items():
    w = int(grid)//256//256
    cyc = v.get('GRBM_GUI_ACTIVE',0)/8
    per16 = cyc/(w*4096) if w else 0
    print(f"{name[:28]:28s} w/SIMD={w} cycles {cyc:10.0f}  per 16 instr {per16:6.1f}  dur {ns/1e3:8.1f} us  clock {cyc/ns if ns else 0:5.2f} GHz")
PY
