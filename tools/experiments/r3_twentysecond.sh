#!/bin/bash
# round 3, twenty-second GPU call: the channelizer's epilogue with the phasor rotated per time block - its tests, then
# bench.py --wideband against the previous epilogue (tools/ab_libs/chan_npf3.so: older sources, same speed as npf2)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3w
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
export GPU_FORCE_BLIT_COPY_SIZE=0
echo "[r3w] pytest channelizer" | tee -a $OUT/progress.log
timeout -k 10 300 python -m pytest tests/test_channelizer.py -m gpu -x -q > $OUT/pytest.log 2>&1; RC=$?
tail -15 $OUT/pytest.log | tee -a $OUT/progress.log
[ $RC -ne 0 ] && exit $RC
for round in 1 2; do
for v in new old; do
  case $v in new) unset RTLDAVIS_HIP_LIB;; *) export RTLDAVIS_HIP_LIB=$ROOT/tools/ab_libs/chan_npf3.so;; esac
  timeout -k 10 200 python3 bench.py --wideband --steps 40 --warmup 5 > $OUT/wb_${v}_$round.json 2> $OUT/wb_${v}_$round.err; RC=$?
  python3 - $OUT/wb_${v}_$round.json $v <<'PY' | tee -a $OUT/progress.log
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "value", d["value"], "ms_per_step", d["ms_per_step"], "channelize_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], d.get("packets_recovered"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  [ $RC -ge 124 ] && exit $RC
done
done
unset RTLDAVIS_HIP_LIB
echo "[r3w] done" | tee -a $OUT/progress.log
