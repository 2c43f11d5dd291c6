# engine clock of the demod kernel variants: GRBM_GUI_ACTIVE (cycles, summed over the 8 XCDs) / duration
# usage: pmc_clock.sh   (runs the variants one after the other, counters only)
cd /tmp && export TMPDIR=/tmp
export GPU_FORCE_BLIT_COPY_SIZE=0
for V in 0 6 2 1 5 4; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_clk_$V
  rm -rf $OUT
  RD_K1_DEBUG=$V timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-verify --sustain 0 > $OUT.log 2>&1
  python3 - $OUT $V <<'PY'
import csv, glob, sys
root, v = sys.argv[1], sys.argv[2]
cyc, dur = [], {}
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "demod" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cyc.append((r["Dispatch_Id"], float(r["Counter_Value"])))
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "demod" in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
per = {}
for d, c in cyc:
    per[d] = per.get(d, 0.0) + c
rows = [(per[d], dur[d]) for d in per if d in dur]
if rows:
    rows = rows[len(rows) // 2:]
    mc = sum(r[0] for r in rows) / len(rows); md = sum(r[1] for r in rows) / len(rows)
    print(f"RD_K1_DEBUG={v}: GRBM_GUI_ACTIVE {mc:.0f} (sum over XCDs), duration {md:.1f} us -> {mc / 8 / md / 1e3:.3f} GHz  (n={len(rows)})")
else:
    print(f"RD_K1_DEBUG={v}: no rows", len(cyc), len(dur))
PY
done
