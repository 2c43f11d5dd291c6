#!/bin/bash
# round 3, seventh GPU call: reworked ordered tail - parity, per-kernel times against the unordered one
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3g
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
export GPU_FORCE_BLIT_COPY_SIZE=0
echo "[r3g] pytest" | tee -a $OUT/progress.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; RC=$?
tail -15 $OUT/pytest.log | tee -a $OUT/progress.log
[ $RC -ne 0 ] && exit $RC
cd /tmp && export TMPDIR=/tmp
for V in ordered legacy; do
  if [ $V = legacy ]; then export RD_TAIL_IMPL=legacy; else unset RD_TAIL_IMPL; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$V -- python3 $ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-verify --sustain 0 > $OUT/bench_$V.json 2> $OUT/trace_$V.err; RC=$?
  echo "[r3g] trace $V rc $RC" | tee -a $OUT/progress.log
  [ $RC -ge 124 ] && exit $RC
  python3 $ROOT/tools/profile_collect.py stats $OUT/trace_$V $OUT ${V}_ > /dev/null
  grep -v "first 12\|last 12\|copyBuffer" $OUT/${V}_kernel_durations.txt | cut -c1-120 | tee -a $OUT/progress.log
done
unset RD_TAIL_IMPL
cd $ROOT
echo "[r3g] A/B tails" | tee -a $OUT/progress.log
timeout -k 10 400 python3 tools/k1_ab.py --key all --rounds 3 ordered=RD_AB_TIMING=1,RD_K1_OPT=2 legacy=RD_AB_TIMING=1,RD_K1_OPT=2,RD_TAIL_IMPL=legacy > $OUT/ab_tail.txt 2>&1; RC=$?
cat $OUT/ab_tail.txt | tee -a $OUT/progress.log
timeout -k 10 200 python3 tools/results_cost.py 2>&1 | tee -a $OUT/results_cost.txt | tee -a $OUT/progress.log
echo "[r3g] done rc $RC" | tee -a $OUT/progress.log
