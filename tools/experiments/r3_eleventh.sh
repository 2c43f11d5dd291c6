#!/bin/bash
# round 3, eleventh GPU call: what the word stores cost beside the tile loads, by cache policy of the store
# (RD_K1_STFLAGS bits 3-5, diagnostic library), and the chunk length once more on the 8-output kernel
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3l
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
export GPU_FORCE_BLIT_COPY_SIZE=0
echo "[r3l] A/B store policy (loads + stores only, then the whole kernel)" | tee -a $OUT/progress.log
RD_AB_TIMING=1 timeout -k 10 900 python3 tools/k1_ab.py --key all --rounds 2 \
  ls_nt=RD_K1_DEBUG=2 ls_plain=RD_K1_DEBUG=2,RD_K1_STFLAGS=1 ls_sc0=RD_K1_DEBUG=2,RD_K1_STFLAGS=8 ls_sc1=RD_K1_DEBUG=2,RD_K1_STFLAGS=16 \
  ls_sc0sc1=RD_K1_DEBUG=2,RD_K1_STFLAGS=24 ls_sc0nt=RD_K1_DEBUG=2,RD_K1_STFLAGS=32 ls_sc1nt=RD_K1_DEBUG=2,RD_K1_STFLAGS=40 ls_sc0sc1nt=RD_K1_DEBUG=2,RD_K1_STFLAGS=48 \
  loads_only=RD_K1_DEBUG=6 \
  full_nt=RD_K1_OPT=10 full_plain=RD_K1_STFLAGS=1 full_sc0=RD_K1_STFLAGS=8 full_sc1=RD_K1_STFLAGS=16 full_sc0sc1=RD_K1_STFLAGS=24 full_sc1nt=RD_K1_STFLAGS=40 full_sc0sc1nt=RD_K1_STFLAGS=48 \
  chunk8=RD_K1_CHUNK=8 chunk12=RD_K1_CHUNK=12 chunk16=RD_K1_CHUNK=16 chunk20=RD_K1_CHUNK=20 > $OUT/ab_store.txt 2>&1; RC=$?
cat $OUT/ab_store.txt | tee -a $OUT/progress.log
echo "[r3l] done rc $RC" | tee -a $OUT/progress.log
