#!/bin/bash
# round 3, fifth GPU call: ordered tail (parity, cost), MFMA count A/B once more
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3e
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
export GPU_FORCE_BLIT_COPY_SIZE=0
echo "[r3e] pytest" | tee -a $OUT/progress.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; RC=$?
tail -15 $OUT/pytest.log | tee -a $OUT/progress.log
[ $RC -ge 124 ] && exit $RC
echo "[r3e] results cost" | tee -a $OUT/progress.log
timeout -k 10 200 python3 tools/results_cost.py 2>&1 | tee -a $OUT/results_cost.txt | tee -a $OUT/progress.log
RD_TAIL_IMPL=legacy timeout -k 10 200 python3 tools/results_cost.py 2>&1 | tee -a $OUT/results_cost.txt | tee -a $OUT/progress.log
echo "[r3e] A/B tails" | tee -a $OUT/progress.log
timeout -k 10 400 python3 tools/k1_ab.py --key all --rounds 3 ordered=RD_AB_TIMING=1,RD_K1_OPT=2 legacy=RD_AB_TIMING=1,RD_K1_OPT=2,RD_TAIL_IMPL=legacy > $OUT/ab_tail.txt 2>&1; RC=$?
cat $OUT/ab_tail.txt | tee -a $OUT/progress.log
[ $RC -ge 124 ] && exit $RC
echo "[r3e] A/B mfma count" | tee -a $OUT/progress.log
timeout -k 10 500 python3 tools/k1_ab.py --key demod_ms --rounds 4 no_guard=RD_K1_DEBUG=7,RD_K1_OPT=0 mfma16=RD_K1_DEBUG=9,RD_K1_OPT=0 > $OUT/ab_mfma.txt 2>&1; RC=$?
cat $OUT/ab_mfma.txt | tee -a $OUT/progress.log
[ $RC -ge 124 ] && exit $RC
echo "[r3e] bench" | tee -a $OUT/progress.log
timeout -k 10 300 python3 bench.py --no-cpu-baseline --sustain 2 > $OUT/bench.json 2> $OUT/bench.err; RC=$?
tail -c 1200 $OUT/bench.json | tee -a $OUT/progress.log
timeout -k 10 300 python3 bench.py --no-cpu-baseline --sustain 0 --steps 20 --warmup 5 > $OUT/bench20.json 2> $OUT/bench20.err; RC=$?
tail -c 1200 $OUT/bench20.json | tee -a $OUT/progress.log
echo "[r3e] done rc $RC" | tee -a $OUT/progress.log
