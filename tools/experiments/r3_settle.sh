#!/bin/bash
# the driver's short timed region (--steps 20 --warmup 5) after different lengths of the untimed settle phase
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3settle
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
for round in 1 2 3; do
for s in 0.05 0.2 0.5 0; do
  timeout -k 10 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --settle $s --no-cpu-baseline --sustain 0 --live-traffic 0 --no-verify > $OUT/b_${s}_$round.json 2> $OUT/b_${s}_$round.err
  python3 - $OUT/b_${s}_$round.json $s <<'PY' | tee -a $OUT/progress.log
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("settle", sys.argv[2], "value", d["value"], "ms_per_step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"])
PY
done
done
