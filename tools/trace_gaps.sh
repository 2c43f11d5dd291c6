# rocprofv3 kernel trace of the default bench: idle gaps between consecutive kernels on the compute queue
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_gaps
rm -rf $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 3 --no-cpu-baseline > $OUT.log 2>&1
tail -1 $OUT.log | cut -c1-200
F=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
main = [r for r in rows if "rocclr" not in r["Kernel_Name"] or "fill" in r["Kernel_Name"]]
# steady state: last 25 runs
gaps = collections.defaultdict(list); dur = collections.defaultdict(list)
prev = None
for r in main[-4 * 25:]:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:16]
    if prev is not None:
        gaps[(prev[0], name)].append((int(r["Start_Timestamp"]) - prev[1]) / 1e3)
    prev = (name, int(r["End_Timestamp"]))
    dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in dur.items(): print(f"kernel {k:18s} mean {sum(v)/len(v):8.1f} us  n={len(v)}")
for k, v in gaps.items(): print(f"gap {k[0]:>16s} -> {k[1]:16s} mean {sum(v)/len(v):7.1f} us  min {min(v):6.1f} max {max(v):7.1f} n={len(v)}")
copies = [r for r in rows if "copyBuffer" in r["Kernel_Name"]][-50:]
print("copyBuffer durations (us):", [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in copies][-16:])
PY
