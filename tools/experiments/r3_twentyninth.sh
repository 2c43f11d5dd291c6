#!/bin/bash
# round 3, twenty-ninth GPU call: pacing sweep (RD_K1_STFLAGS bits 24-31, diagnostic library): which subset of waves
# sleeping how long behind the tile loads shortens the demod kernel
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3ad
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
export GPU_FORCE_BLIT_COPY_SIZE=0
RD_AB_TIMING=1 timeout -k 10 1000 python3 tools/k1_ab.py --key all --rounds 3 full=RD_K1_OPT=10 m1_n1=RD_K1_STFLAGS=285212672 m1_n2=RD_K1_STFLAGS=553648128 m1_n4=RD_K1_STFLAGS=1090519040 m2_n1=RD_K1_STFLAGS=301989888 m2_n2=RD_K1_STFLAGS=570425344 m3_n1=RD_K1_STFLAGS=318767104 m3_n2=RD_K1_STFLAGS=587202560 m5_n1=RD_K1_STFLAGS=352321536 m6_n1=RD_K1_STFLAGS=369098752 m7_n1=RD_K1_STFLAGS=385875968 m8_n1=RD_K1_STFLAGS=402653184 m8_n4=RD_K1_STFLAGS=1207959552 full2=RD_K1_OPT=10 > $OUT/ab_pace.txt 2>&1; RC=$?
cat $OUT/ab_pace.txt | tee -a $OUT/progress.log
echo "[r3ad] done rc $RC" | tee -a $OUT/progress.log
