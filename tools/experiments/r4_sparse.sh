#!/bin/bash
# the demod kernel on the 2:4-sparse matrix instruction (the tree) against the dense pair (librtldavis_hip_dense.so, make dense):
# the whole GPU suite (the dense build is one of its tests), then three interleaved rounds of the bench
set -e
mkdir -p gpurun_out/r4sparse
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r4sparse/tests.log 2>&1 || { tail -40 gpurun_out/r4sparse/tests.log; exit 1; }
tail -2 gpurun_out/r4sparse/tests.log
for r in 1 2 3; do
  for v in dense tree; do
    if [ $v = tree ]; then unset RTLDAVIS_HIP_LIB; else export RTLDAVIS_HIP_LIB=$PWD/rtldavis_amd/librtldavis_hip_$v.so; fi
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --live-traffic 0 --sustain 4 > gpurun_out/r4sparse/${v}_$r.json 2> gpurun_out/r4sparse/${v}_$r.err
    python3 -c "
import json
d=json.loads(open('gpurun_out/r4sparse/${v}_$r.json').read().strip().splitlines()[-1])
k=d['kernels_ms']; s=d['sustained']
print('$v', $r, 'demod', k['demod'], 'total', k['total'], 'step', d['ms_per_step'], 'frac', d['roofline']['frac'], '| sustained kernel', s['kernel_ms'], 'frac', s['roofline_frac'], 'power', s.get('package_power_W'), 'sclk', s.get('sclk_MHz'), 'verified', d.get('verified_vs_reference_fixtures'))"
  done
done
