#!/bin/bash
# round 3, twelfth GPU call: GPU suite (one-launch block at other block sizes), the fix-up grid sized from the last
# run's list, kernel trace of a step
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3m
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
export GPU_FORCE_BLIT_COPY_SIZE=0
echo "[r3m] pytest" | tee -a $OUT/progress.log
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; RC=$?
tail -30 $OUT/pytest.log | tee -a $OUT/progress.log
[ $RC -ne 0 ] && exit $RC
echo "[r3m] kernel trace" | tee -a $OUT/progress.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-verify --sustain 0 > $OUT/bench_trace.json 2> $OUT/trace.err; RC=$?
python3 $ROOT/tools/profile_collect.py stats $OUT/trace $OUT > /dev/null
grep -v "first 12\|last 12\|copyBuffer" $OUT/kernel_durations.txt | cut -c1-120 | tee -a $OUT/progress.log
[ $RC -ge 124 ] && exit $RC
cd $ROOT
echo "[r3m] bench" | tee -a $OUT/progress.log
timeout -k 10 300 python3 bench.py --no-cpu-baseline --sustain 3 > $OUT/bench.json 2> $OUT/bench.err; RC=$?
tail -c 1500 $OUT/bench.json | tee -a $OUT/progress.log
echo "[r3m] done rc $RC" | tee -a $OUT/progress.log
