# rocprofv3 kernel trace of the two-stream bench: do kernels of the two resident batches overlap?
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace2s
rm -rf $OUT
RD_K1_WGS_PER_CU=${WGS:-5} rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --sustain 0 --two-streams > $OUT.log 2>&1
tail -1 $OUT.log | cut -c1-300
F=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(rows[0].keys())
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
last = rows[-60:]
for r in last:
    print(r.get("Queue_Id"), r.get("Stream_Id", ""), r["Kernel_Name"][:28].ljust(28), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
PY
