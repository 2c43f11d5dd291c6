#!/bin/bash
# round 3, fourth GPU call: which part of the kernel costs the clock when the loads run (stamps), memory-side counters
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3d
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
export GPU_FORCE_BLIT_COPY_SIZE=0
echo "[r3d] stamps" | tee -a $OUT/progress.log
timeout -k 10 400 python3 tools/k1_stamps.py halo=RD_K1_OPT=6 no_mfma_loads=RD_K1_DEBUG=10,RD_K1_OPT=4 mfma_only_loads=RD_K1_DEBUG=11,RD_K1_OPT=4 no_guard=RD_K1_DEBUG=7,RD_K1_OPT=4 > $OUT/stamps.txt 2>&1; RC=$?
cat $OUT/stamps.txt | tee -a $OUT/progress.log
[ $RC -ge 124 ] && exit $RC
echo "[r3d] A/B" | tee -a $OUT/progress.log
timeout -k 10 400 python3 tools/k1_ab.py --key all --rounds 2 halo=RD_K1_OPT=2 no_guard=RD_K1_DEBUG=7,RD_K1_OPT=0 no_mfma_loads=RD_K1_DEBUG=10,RD_K1_OPT=0 mfma_only_loads=RD_K1_DEBUG=11,RD_K1_OPT=0 > $OUT/ab.txt 2>&1; RC=$?
cat $OUT/ab.txt | tee -a $OUT/progress.log
[ $RC -ge 124 ] && exit $RC
echo "[r3d] pmc" | tee -a $OUT/progress.log
bash tools/pmc_memside.sh $OUT/pmc; RC=$?
echo "[r3d] done rc $RC" | tee -a $OUT/progress.log
