#!/bin/bash
# round 3, second GPU call: the clock each variant of the demod kernel really holds (stamps), memory-side counters
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3b
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
export GPU_FORCE_BLIT_COPY_SIZE=0
echo "[r3b] stamps" | tee -a $OUT/progress.log
timeout -k 10 600 python3 tools/k1_stamps.py base=RD_K1_OPT=4 halo=RD_K1_OPT=6 pipe=RD_K1_OPT=5 \
   no_guard=RD_K1_DEBUG=7,RD_K1_OPT=4 mfma16=RD_K1_DEBUG=9,RD_K1_OPT=4 compute_only=RD_K1_DEBUG=1,RD_K1_OPT=4 \
   no_mfma=RD_K1_DEBUG=4,RD_K1_OPT=4 mfma_only=RD_K1_DEBUG=5,RD_K1_OPT=4 \
   loads_stores_halo=RD_K1_DEBUG=2,RD_K1_OPT=6 loads_only_halo=RD_K1_DEBUG=6,RD_K1_OPT=6 > $OUT/stamps.txt 2>&1; RC=$?
cat $OUT/stamps.txt | tee -a $OUT/progress.log
[ $RC -ge 124 ] && exit $RC
echo "[r3b] A/B mfma16" | tee -a $OUT/progress.log
timeout -k 10 300 python3 tools/k1_ab.py --key all --rounds 2 halo=RD_K1_OPT=2 no_guard=RD_K1_DEBUG=7,RD_K1_OPT=0 mfma16=RD_K1_DEBUG=9,RD_K1_OPT=0 > $OUT/ab.txt 2>&1; RC=$?
cat $OUT/ab.txt | tee -a $OUT/progress.log
[ $RC -ge 124 ] && exit $RC
echo "[r3b] pmc" | tee -a $OUT/progress.log
bash tools/pmc_memside.sh $OUT/pmc; RC=$?
echo "[r3b] done rc $RC" | tee -a $OUT/progress.log
