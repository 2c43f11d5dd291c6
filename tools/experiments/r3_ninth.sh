#!/bin/bash
# round 3, ninth GPU call: product = 8-output demod kernel + ordered tail over 32-stream lists + one-launch streaming block
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3i
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
export GPU_FORCE_BLIT_COPY_SIZE=0
echo "[r3i] pytest" | tee -a $OUT/progress.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; RC=$?
tail -30 $OUT/pytest.log | tee -a $OUT/progress.log
[ $RC -ge 124 ] && exit $RC
echo "[r3i] stream latency" | tee -a $OUT/progress.log
timeout -k 10 200 python3 tools/stream_latency.py 2>&1 | tee $OUT/stream_latency.txt | tee -a $OUT/progress.log
RD_STREAM_IMPL=legacy timeout -k 10 200 python3 tools/stream_latency.py 2>&1 | tee $OUT/stream_latency_legacy.txt | tee -a $OUT/progress.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-verify --sustain 0 > $OUT/bench_trace.json 2> $OUT/trace.err; RC=$?
python3 $ROOT/tools/profile_collect.py stats $OUT/trace $OUT > /dev/null
grep -v "first 12\|last 12\|copyBuffer" $OUT/kernel_durations.txt | cut -c1-120 | tee -a $OUT/progress.log
cd $ROOT
echo "[r3i] A/B tails" | tee -a $OUT/progress.log
timeout -k 10 400 python3 tools/k1_ab.py --key all --rounds 3 ordered=RD_AB_TIMING=1,RD_K1_OPT=10 legacy=RD_AB_TIMING=1,RD_K1_OPT=10,RD_TAIL_IMPL=legacy > $OUT/ab_tail.txt 2>&1; RC=$?
cat $OUT/ab_tail.txt | tee -a $OUT/progress.log
echo "[r3i] bench" | tee -a $OUT/progress.log
timeout -k 10 300 python3 bench.py --no-cpu-baseline --sustain 2 > $OUT/bench.json 2> $OUT/bench.err; RC=$?
tail -c 1300 $OUT/bench.json | tee -a $OUT/progress.log
timeout -k 10 300 python3 bench.py --no-cpu-baseline --sustain 0 --steps 20 --warmup 5 > $OUT/bench20.json 2> $OUT/bench20.err
tail -c 700 $OUT/bench20.json | tee -a $OUT/progress.log
echo "[r3i] done rc $RC" | tee -a $OUT/progress.log
