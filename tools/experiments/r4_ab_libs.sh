#!/bin/bash
# A/B of builds of the library inside one gpurun call: every rtldavis_amd/librtldavis_hip_<tag>.so named on the command
# line against the tree's, three interleaved rounds of the default bench (no CPU leg, no live traffic, 4 s sustained leg);
# first the batch tests on each variant
set -e
mkdir -p gpurun_out/ab
for v in "$@"; do
  RTLDAVIS_HIP_LIB=$PWD/rtldavis_amd/librtldavis_hip_$v.so timeout -k 10 300 python3 -m pytest tests/test_gpu_mfma.py tests/test_gpu_parity.py -x -q -m gpu -k "(mfma or pipe or tail or long_streams or golden or batch) and not dense_matrix" 2>&1 | tail -1
done
for r in 1 2 3; do
  for v in tree "$@"; do
    if [ $v = tree ]; then unset RTLDAVIS_HIP_LIB; else export RTLDAVIS_HIP_LIB=$PWD/rtldavis_amd/librtldavis_hip_$v.so; fi
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --live-traffic 0 --sustain 4 > gpurun_out/ab/${v}_$r.json 2> gpurun_out/ab/${v}_$r.err
    python3 -c "
import json
d=json.loads(open('gpurun_out/ab/${v}_$r.json').read().strip().splitlines()[-1])
k=d['kernels_ms']; s=d['sustained']
print('$v', $r, 'demod', k['demod'], 'total', k['total'], 'step', d['ms_per_step'], 'frac', d['roofline']['frac'], '| sustained kernel', s['kernel_ms'], 'step', s['ms_per_step'], 'power', s.get('package_power_W'), 'sclk', s.get('sclk_MHz'), 'verified', d.get('verified_vs_reference_fixtures'))"
  done
done
