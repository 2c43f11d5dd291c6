#!/bin/bash
# round 3, third GPU call: sign-chain / tail cleanups against the previous build (same box, interleaved), parity
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3c
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
export GPU_FORCE_BLIT_COPY_SIZE=0
echo "[r3c] pytest" | tee -a $OUT/progress.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; RC=$?
tail -5 $OUT/pytest.log | tee -a $OUT/progress.log
[ $RC -ge 124 ] && exit $RC
echo "[r3c] A/B" | tee -a $OUT/progress.log
P=$ROOT/tools/ab_libs/diag_r3b.so
timeout -k 10 700 python3 tools/k1_ab.py --key all --rounds 3 prev=RTLDAVIS_HIP_LIB=$P,RD_K1_OPT=2 new=RD_K1_OPT=2 \
   prev_noguard=RTLDAVIS_HIP_LIB=$P,RD_K1_DEBUG=7,RD_K1_OPT=0 new_noguard=RD_K1_DEBUG=7,RD_K1_OPT=0 new_pipe=RD_K1_OPT=3 > $OUT/ab.txt 2>&1; RC=$?
cat $OUT/ab.txt | tee -a $OUT/progress.log
[ $RC -ge 124 ] && exit $RC
echo "[r3c] bench" | tee -a $OUT/progress.log
timeout -k 10 300 python3 bench.py --no-cpu-baseline --sustain 2 > $OUT/bench.json 2> $OUT/bench.err; RC=$?
tail -c 1500 $OUT/bench.json | tee -a $OUT/progress.log
echo "[r3c] done rc $RC" | tee -a $OUT/progress.log
