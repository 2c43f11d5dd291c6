#!/bin/bash
# round 3, fourteenth GPU call: the channelizer's A prefetch depth (RD_CHAN_NPF = 2 product, 3 / 4 / 6 alternates)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3o
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
export GPU_FORCE_BLIT_COPY_SIZE=0
for round in 1 2; do
for v in npf2 npf3 npf4 npf6; do
  case $v in npf2) unset RTLDAVIS_HIP_LIB;; *) export RTLDAVIS_HIP_LIB=$ROOT/tools/ab_libs/chan_$v.so;; esac
  timeout -k 10 200 python3 bench.py --wideband --steps 40 --warmup 5 > $OUT/wb_${v}_$round.json 2> $OUT/wb_${v}_$round.err; RC=$?
  python3 - $OUT/wb_${v}_$round.json $v <<'PY' | tee -a $OUT/progress.log
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "value", d["value"], "ms_per_step", d["ms_per_step"], "channelize_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], d.get("packets_recovered"), d.get("verified"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  [ $RC -ge 124 ] && exit $RC
done
done
echo "[r3o] done" | tee -a $OUT/progress.log
