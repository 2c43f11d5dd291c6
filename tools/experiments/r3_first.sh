#!/bin/bash
# round 3, first GPU call: parity of the new product kernel, the counter list, A/B of the new variants, stamps
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3a
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
export GPU_FORCE_BLIT_COPY_SIZE=0
echo "[r3a] pytest" | tee -a $OUT/progress.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; RC=$?
tail -5 $OUT/pytest.log | tee -a $OUT/progress.log
if [ $RC -ne 0 ]; then echo "pytest rc $RC" | tee -a $OUT/progress.log; [ $RC -ge 124 ] && exit $RC; fi
echo "[r3a] counters" | tee -a $OUT/progress.log
(cd /tmp && timeout -k 10 120 rocprofv3 -L > $OUT/counters.txt 2>&1)
grep -c . $OUT/counters.txt | tee -a $OUT/progress.log
echo "[r3a] A/B" | tee -a $OUT/progress.log
timeout -k 10 700 python3 tools/k1_ab.py --key all --rounds 2 base=RD_K1_OPT=0 pipe=RD_K1_OPT=1 halo=RD_K1_OPT=2 both=RD_K1_OPT=3 \
  compute_only=RD_K1_DEBUG=1,RD_K1_OPT=0 compute_only_pipe=RD_K1_DEBUG=1,RD_K1_OPT=1 \
  loads_only=RD_K1_DEBUG=6,RD_K1_OPT=0 loads_only_halo=RD_K1_DEBUG=6,RD_K1_OPT=2 \
  loads_stores=RD_K1_DEBUG=2,RD_K1_OPT=0 loads_stores_halo=RD_K1_DEBUG=2,RD_K1_OPT=2 > $OUT/ab.txt 2>&1; RC=$?
cat $OUT/ab.txt | tee -a $OUT/progress.log
[ $RC -ge 124 ] && exit $RC
echo "[r3a] stamps" | tee -a $OUT/progress.log
timeout -k 10 400 python3 tools/k1_stamps.py base=RD_K1_OPT=4 both=RD_K1_OPT=7 loads_stores=RD_K1_DEBUG=2,RD_K1_OPT=4 loads_only=RD_K1_DEBUG=6,RD_K1_OPT=4 > $OUT/stamps.txt 2>&1; RC=$?
cat $OUT/stamps.txt | tee -a $OUT/progress.log
echo "[r3a] done rc $RC" | tee -a $OUT/progress.log
