#!/bin/bash
# the streaming blocks after their rework: the whole GPU suite, then stamps and the host's view
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/r4cplx3
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r4cplx3/tests.log 2>&1 || { tail -30 gpurun_out/r4cplx3/tests.log; exit 1; }
tail -3 gpurun_out/r4cplx3/tests.log
RTLDAVIS_HIP_LIB=$PWD/rtldavis_amd/librtldavis_hip_diag.so RD_SB_STAMPS=1 timeout -k 10 200 python3 tools/experiments/r4_sb_stamps.py | tee gpurun_out/r4cplx3/stamps.txt
timeout -k 10 200 python3 tools/experiments/r4_cplx_kernel.py | tee gpurun_out/r4cplx3/host.txt
timeout -k 10 300 python3 tools/stream_latency.py | tee gpurun_out/r4cplx3/stream_latency.txt
