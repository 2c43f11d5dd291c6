#!/bin/bash
# host push of the streaming blocks into device memory (large BAR) against the pinned slot (RD_PUSH_INPUT=0)
set -e
mkdir -p gpurun_out/r4push
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "complex or mixed or stream or one_launch or worker or pipeline or multi or ring or deadline or submit" > gpurun_out/r4push/tests.log 2>&1 || { tail -30 gpurun_out/r4push/tests.log; exit 1; }
tail -2 gpurun_out/r4push/tests.log
for v in 1 0; do
  echo "== RD_PUSH_INPUT=$v"
  RD_PUSH_INPUT=$v RTLDAVIS_HIP_LIB=$PWD/rtldavis_amd/librtldavis_hip_diag.so RD_SB_STAMPS=1 timeout -k 10 200 python3 tools/stream_stamps.py | grep -E "block of|link|whole"
  RD_PUSH_INPUT=$v timeout -k 10 300 python3 tools/stream_latency.py > gpurun_out/r4push/lat_$v.txt 2>&1
  grep -E "^demodulate|complex input, (demod|submit)|^submit|16 receivers" gpurun_out/r4push/lat_$v.txt
done
