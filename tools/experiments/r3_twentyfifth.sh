#!/bin/bash
# round 3, twenty-fifth GPU call: k_rssi_ord under 128 VGPRs (four waves per SIMD: its whole grid resident at once) -
# GPU suite, kernel trace, ordered against unordered tail
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3z
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
export GPU_FORCE_BLIT_COPY_SIZE=0
echo "[r3z] pytest" | tee -a $OUT/progress.log
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; RC=$?
tail -5 $OUT/pytest.log | tee -a $OUT/progress.log
[ $RC -ne 0 ] && exit $RC
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-verify --sustain 0 --live-traffic 0 > $OUT/bench_trace.json 2> $OUT/trace.err; RC=$?
python3 $ROOT/tools/profile_collect.py stats $OUT/trace $OUT > /dev/null
grep -v "first 12\|last 12\|copyBuffer" $OUT/kernel_durations.txt | cut -c1-120 | tee -a $OUT/progress.log
[ $RC -ge 124 ] && exit $RC
cd $ROOT
RD_AB_TIMING=1 timeout -k 10 400 python3 tools/k1_ab.py --key all --rounds 3 ordered=RD_K1_OPT=10 legacy_tail=RD_TAIL_IMPL=legacy > $OUT/ab_tail.txt 2>&1
cat $OUT/ab_tail.txt | tee -a $OUT/progress.log
echo "[r3z] done" | tee -a $OUT/progress.log
