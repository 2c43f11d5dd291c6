#!/bin/bash
# the complex one-launch block after its rework: parity tests first, then stamps and the host's view
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/r4cplx2
timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "complex or mixed or stream or ring or one_launch or worker or block" > gpurun_out/r4cplx2/tests.log 2>&1 || { tail -30 gpurun_out/r4cplx2/tests.log; exit 1; }
tail -3 gpurun_out/r4cplx2/tests.log
RTLDAVIS_HIP_LIB=$PWD/rtldavis_amd/librtldavis_hip_diag.so RD_SB_STAMPS=1 timeout -k 10 200 python3 tools/experiments/r4_sb_stamps.py | tee gpurun_out/r4cplx2/stamps.txt
timeout -k 10 200 python3 tools/experiments/r4_cplx_kernel.py | tee gpurun_out/r4cplx2/host.txt
