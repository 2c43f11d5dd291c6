#!/bin/bash
# round 3, eighth GPU call: the 8-output formulation (16 MFMAs per tile) - parity through the whole GPU suite with the
# diagnostic library selecting it, then A/B against the 16-output kernel; the ordered tail's kernels once more
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3h
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
export GPU_FORCE_BLIT_COPY_SIZE=0
echo "[r3h] pytest (product)" | tee -a $OUT/progress.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; RC=$?
tail -5 $OUT/pytest.log | tee -a $OUT/progress.log
[ $RC -ne 0 ] && exit $RC
echo "[r3h] pytest (B8 through the diagnostic library)" | tee -a $OUT/progress.log
RTLDAVIS_HIP_LIB=$ROOT/rtldavis_amd/librtldavis_hip_diag.so RD_K1_OPT=10 timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest_b8.log 2>&1; RC=$?
tail -15 $OUT/pytest_b8.log | tee -a $OUT/progress.log
[ $RC -ge 124 ] && exit $RC
echo "[r3h] A/B" | tee -a $OUT/progress.log
timeout -k 10 600 python3 tools/k1_ab.py --key all --rounds 3 b16=RD_AB_TIMING=1,RD_K1_OPT=2 b8=RD_AB_TIMING=1,RD_K1_OPT=10 b8_nohalo=RD_AB_TIMING=1,RD_K1_OPT=8 b8_noguard=RD_AB_TIMING=1,RD_K1_DEBUG=7,RD_K1_OPT=10 legacy_tail=RD_AB_TIMING=1,RD_K1_OPT=2,RD_TAIL_IMPL=legacy > $OUT/ab.txt 2>&1; RC=$?
cat $OUT/ab.txt | tee -a $OUT/progress.log
[ $RC -ge 124 ] && exit $RC
echo "[r3h] stamps" | tee -a $OUT/progress.log
timeout -k 10 200 python3 tools/k1_stamps.py b16=RD_K1_OPT=6 b8=RD_K1_OPT=14 > $OUT/stamps.txt 2>&1
grep -E "^==|demod_ms|cycles_per_tile|wait_per|gap_per|comp_per|clock_GHz_med|kernel_span" $OUT/stamps.txt | tee -a $OUT/progress.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-verify --sustain 0 > $OUT/bench_trace.json 2> $OUT/trace.err; RC=$?
python3 $ROOT/tools/profile_collect.py stats $OUT/trace $OUT > /dev/null
grep -v "first 12\|last 12\|copyBuffer" $OUT/kernel_durations.txt | cut -c1-120 | tee -a $OUT/progress.log
echo "[r3h] done rc $RC" | tee -a $OUT/progress.log
