#!/bin/bash
# A/B of two builds of the library inside one gpurun call: rtldavis_amd/librtldavis_hip_a.so (before) against the tree's (after),
# three interleaved rounds of the default bench (no CPU leg, no live traffic, no sustained leg)
set -e
mkdir -p gpurun_out/ab
for r in 1 2 3; do
  for v in a b; do
    if [ $v = a ]; then export RTLDAVIS_HIP_LIB=$PWD/rtldavis_amd/librtldavis_hip_a.so; else unset RTLDAVIS_HIP_LIB; fi
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --live-traffic 0 --sustain 0 > gpurun_out/ab/${v}_$r.json 2> gpurun_out/ab/${v}_$r.err
    python3 -c "
import json
d=json.loads(open('gpurun_out/ab/${v}_$r.json').read().strip().splitlines()[-1])
print('$v', $r, 'demod', d['kernels_ms']['demod'], 'total', d['kernels_ms']['total'], 'step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'verified', d.get('verified_vs_reference_fixtures'))"
  done
done
