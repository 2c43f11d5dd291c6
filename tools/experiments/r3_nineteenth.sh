#!/bin/bash
# round 3, nineteenth GPU call: what the whole demod kernel gains when most (or all) of its word stores are skipped
# (diagnostic library, RD_K1_STFLAGS 512 / 1024: wrong results, timing only)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3t
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
export GPU_FORCE_BLIT_COPY_SIZE=0
RD_AB_TIMING=1 timeout -k 10 600 python3 tools/k1_ab.py --key all --rounds 3 full=RD_K1_OPT=10 skip3of4=RD_K1_STFLAGS=512 skip_all=RD_K1_STFLAGS=1024 ls_only=RD_K1_DEBUG=2 ls_skip3of4=RD_K1_DEBUG=2,RD_K1_STFLAGS=512 loads_only=RD_K1_DEBUG=6 > $OUT/ab_skip.txt 2>&1; RC=$?
cat $OUT/ab_skip.txt | tee -a $OUT/progress.log
echo "[r3t] done rc $RC" | tee -a $OUT/progress.log
