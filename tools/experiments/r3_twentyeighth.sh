#!/bin/bash
# round 3, twenty-eighth GPU call: pacing experiments on the demod kernel (RD_K1_STFLAGS 2048 / 16384 / 32768: short
# sleeps behind the tile loads; diagnostic library)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3ac
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
export GPU_FORCE_BLIT_COPY_SIZE=0
RD_AB_TIMING=1 timeout -k 10 600 python3 tools/k1_ab.py --key all --rounds 3 full=RD_K1_OPT=10 sleep_each=RD_K1_STFLAGS=2048 sleep_store_tiles=RD_K1_STFLAGS=16384 sleep_odd_waves=RD_K1_STFLAGS=32768 no_stores=RD_K1_STFLAGS=1024 no_stores_sleep=RD_K1_STFLAGS=17408 > $OUT/ab_pace.txt 2>&1; RC=$?
cat $OUT/ab_pace.txt | tee -a $OUT/progress.log
echo "[r3ac] done rc $RC" | tee -a $OUT/progress.log
