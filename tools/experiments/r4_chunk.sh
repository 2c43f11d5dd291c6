#!/bin/bash
# round 4: tiles per chunk of the demod kernel (RD_K1_CHUNK) with the one-launch tail, interleaved on one box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r4chunk}
mkdir -p $OUT
cd $ROOT
for rep in 1 2 3; do
  for c in 16 8 12 4 24; do
    RD_K1_CHUNK=$c timeout -k 10 200 python3 bench.py --no-cpu-baseline --live-traffic 0 --sustain 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('chunk $c rep $rep: value', d['value'], 'ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'], d['kernels_ms'], 'fix', d['fixup_runs_frac'])"
  done
done | tee $OUT/chunk.txt
