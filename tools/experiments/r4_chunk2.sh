#!/bin/bash
# tiles per chunk with the sparse-instruction kernel: the default (16) against 8 and 12, three interleaved rounds of the bench
mkdir -p gpurun_out/r4chunk2
for r in 1 2 3; do
  for c in 8 4 6 10; do
    RD_K1_CHUNK=$c timeout -k 10 200 python3 bench.py --no-cpu-baseline --live-traffic 0 --sustain 3 > gpurun_out/r4chunk2/c${c}_$r.json 2> gpurun_out/r4chunk2/c${c}_$r.err
    python3 -c "
import json
d=json.loads(open('gpurun_out/r4chunk2/c${c}_$r.json').read().strip().splitlines()[-1])
k=d['kernels_ms']; s=d['sustained']
print('chunk $c round $r demod', k['demod'], 'total', k['total'], 'step', d['ms_per_step'], '| sustained kernel', s['kernel_ms'], 'step', s['ms_per_step'], 'fixup_runs_frac', d.get('fixup_runs_frac'))"
  done
done
