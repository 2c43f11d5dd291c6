#!/bin/bash
# A/B of builds of the library inside one gpurun call: every rtldavis_amd/librtldavis_hip_<tag>.so named on the command
# line against the tree's, three interleaved rounds of the default bench (no CPU leg, no live traffic, no sustained leg);
# first the tail's tests on each variant
set -e
mkdir -p gpurun_out/ab
for v in "$@"; do
  RTLDAVIS_HIP_LIB=$PWD/rtldavis_amd/librtldavis_hip_$v.so timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tail or long_streams or golden or batch" 2>&1 | tail -2
done
for r in 1 2 3; do
  for v in tree "$@"; do
    if [ $v = tree ]; then unset RTLDAVIS_HIP_LIB; else export RTLDAVIS_HIP_LIB=$PWD/rtldavis_amd/librtldavis_hip_$v.so; fi
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --live-traffic 0 --sustain 0 > gpurun_out/ab/${v}_$r.json 2> gpurun_out/ab/${v}_$r.err
    python3 -c "
import json
d=json.loads(open('gpurun_out/ab/${v}_$r.json').read().strip().splitlines()[-1])
k=d['kernels_ms']
print('$v', $r, 'demod', k['demod'], 'total', k['total'], 'tail', round(k['total']-k['demod'],4), 'step', d['ms_per_step'], 'verified', d.get('verified_vs_reference_fixtures'))"
  done
done
