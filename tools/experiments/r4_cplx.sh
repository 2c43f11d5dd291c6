#!/bin/bash
# kernel time of the streaming one-launch blocks, next to the host's view of the same calls
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/r4cplx
timeout -k 10 200 python3 tools/experiments/r4_cplx_kernel.py | tee gpurun_out/r4cplx/host.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4cplx/prof -o cplx -- python3 tools/experiments/r4_cplx_kernel.py > gpurun_out/r4cplx/prof.log 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r4cplx/prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:60], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
