#!/bin/bash
# One script that regenerates every measured artefact under profiles/ for the binary in the tree.
#   usage (from the repo root on the GPU box):  bash tools/profile_round.sh r03 <commit>
# Writes gpurun_out/<tag>/... ; copy the listed files to profiles/ afterwards (tools/profile_collect.py).
# Every rocprofv3 run is its own process with python3 straight after `--`; counters (--pmc) never share a
# run with a trace; each step is time-limited and the script stops at the first failure.
set -e -o pipefail
TAG=${1:-r04}
COMMIT=${2:-unknown}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the profiler initialises HIP before python runs: what rtldavis_amd/__init__.py and bench.py put into the
# environment themselves has to be there already (readback on the SDMA engines, not on a blit kernel)
export GPU_FORCE_BLIT_COPY_SIZE=0
step() { echo "[profile_round] $*" | tee -a $OUT/progress.log; }

step "1/9 default bench line"
timeout -k 10 400 python3 $ROOT/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 400 $OUT/bench_default.json | tee -a $OUT/progress.log

step "2/9 rocprofv3 --kernel-trace --stats of the same command"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline --live-traffic 0 > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
python3 $ROOT/tools/profile_collect.py stats $OUT/trace $OUT | tee -a $OUT/progress.log

step "3/9 SQ counters (own run)"
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --sustain 0 --live-traffic 0 > $OUT/pmc_sq.log 2>&1
python3 $ROOT/tools/profile_collect.py pmc $OUT/pmc_sq > $OUT/pmc_sq_counters.csv

step "4/9 HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --sustain 0 --live-traffic 0 > $OUT/pmc_$C.log 2>&1
done
python3 $ROOT/tools/profile_collect.py traffic $OUT $COMMIT > $OUT/traffic.json
cat $OUT/traffic.json | tee -a $OUT/progress.log

step "5/9 ablations of the demod kernel (tools/k1_ab.py, diagnostic library: librtldavis_hip_diag.so)"
# RD_AB_TIMING=1: demod kernel + whole run.  RD_K1_DEBUG (wrong results): 1 compute only, 2 loads + stores only, 6 loads only,
# 7 no guard band; valu = the round-1 kernel; legacy_tail = the separate tail kernels + host ordering
RD_AB_TIMING=1 timeout -k 10 700 python3 $ROOT/tools/k1_ab.py --key all --rounds 2 product no_guard=RD_K1_DEBUG=7 compute_only=RD_K1_DEBUG=1 loads_stores_only=RD_K1_DEBUG=2 loads_only=RD_K1_DEBUG=6 plain_stores=RD_K1_STFLAGS=1 valu=RD_K1_IMPL=valu wgs3=RD_K1_WGS_PER_CU=3 chunk8=RD_K1_CHUNK=8 chunk28=RD_K1_CHUNK=28 half_mfma=RD_K1_DEBUG=8 half_mfma_init=RD_K1_DEBUG=9 no_exchange=RD_K1_DEBUG=10 legacy_tail=RD_TAIL_IMPL=legacy > $OUT/ablation.txt 2>&1
cat $OUT/ablation.txt | tee -a $OUT/progress.log

step "6/9 wideband (channelizer) line and its kernel stats"
timeout -k 10 300 python3 $ROOT/bench.py --wideband --steps 20 --warmup 3 > $OUT/bench_wideband.json 2> $OUT/bench_wideband.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_wb -- python3 $ROOT/bench.py --wideband --steps 20 --warmup 3 > $OUT/bench_wideband_under_rocprof.json 2> $OUT/trace_wb.err
python3 $ROOT/tools/profile_collect.py stats $OUT/trace_wb $OUT wideband_ | tee -a $OUT/progress.log

step "7/9 streaming handle: uint8 and complex input, the multi-launch form, the worker's queue hop against the ring"
timeout -k 10 200 python3 $ROOT/tools/stream_latency.py > $OUT/stream_latency.txt 2>&1
echo "== RD_PUSH_INPUT=0 (the pinned slot, read by the kernel across the bus)" >> $OUT/stream_latency.txt
RD_PUSH_INPUT=0 timeout -k 10 200 python3 $ROOT/tools/stream_latency.py 2>&1 | grep -E "complex|^demodulate|^submit|16 receivers" >> $OUT/stream_latency.txt
echo "== RD_STREAM_IMPL=legacy" >> $OUT/stream_latency.txt
RD_STREAM_IMPL=legacy timeout -k 10 200 python3 $ROOT/tools/stream_latency.py 2>&1 | grep -E "complex|^demodulate" >> $OUT/stream_latency.txt
echo "== worker: queue hop against the shared-memory ring (tools/worker_rate.py)" >> $OUT/stream_latency.txt
timeout -k 10 300 python3 $ROOT/tools/worker_rate.py 2>&1 | grep -v "DSP worker\|stop signal" >> $OUT/stream_latency.txt
cat $OUT/stream_latency.txt | tee -a $OUT/progress.log

step "8/9 host cost of rd_batch_results, one-shot latency"
timeout -k 10 200 python3 $ROOT/tools/results_cost.py > $OUT/results_cost.txt 2>&1
cat $OUT/results_cost.txt | tee -a $OUT/progress.log

step "8b/9 host-fed leg and the one-launch tail's phase stamps"
timeout -k 10 300 python3 $ROOT/bench.py --no-cpu-baseline --live-traffic 0 --sustain 0 --host-fed > $OUT/host_fed.json 2> $OUT/host_fed.err
tail -c 500 $OUT/host_fed.json | tee -a $OUT/progress.log
RTLDAVIS_HIP_LIB=$ROOT/rtldavis_amd/librtldavis_hip_diag.so RD_FT_STAMPS=1 timeout -k 10 300 python3 $ROOT/tools/tail_stamps.py > $OUT/tail_stamps.txt 2>&1
cat $OUT/tail_stamps.txt | tee -a $OUT/progress.log
RTLDAVIS_HIP_LIB=$ROOT/rtldavis_amd/librtldavis_hip_diag.so RD_SB_STAMPS=1 timeout -k 10 300 python3 $ROOT/tools/stream_stamps.py > $OUT/stream_stamps.txt 2>&1
cat $OUT/stream_stamps.txt | tee -a $OUT/progress.log

step "9/9 N = 2 rehearsal on one GPU (code path only, not a measurement)"
timeout -k 10 400 python3 $ROOT/bench.py --gpus 2 --steps 6 --warmup 2 --rehearse-shared-gpu --streams 1024 --sustain 0 > $OUT/rehearse_shared_gpu.txt 2>&1
tail -c 600 $OUT/rehearse_shared_gpu.txt | tee -a $OUT/progress.log
step "done"
