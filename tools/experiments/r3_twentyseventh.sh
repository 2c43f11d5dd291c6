#!/bin/bash
# round 3, twenty-seventh GPU call: the channelizer with the window staged as raw bytes (f16-subnormal operands) -
# its tests, then bench.py --wideband
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3ab
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
export GPU_FORCE_BLIT_COPY_SIZE=0
echo "[r3ab] pytest channelizer" | tee -a $OUT/progress.log
timeout -k 10 300 python -m pytest tests/test_channelizer.py -m gpu -x -q > $OUT/pytest.log 2>&1; RC=$?
tail -25 $OUT/pytest.log | cut -c1-200 | tee -a $OUT/progress.log
python3 tools/chan_model_diff.py 2>&1 | tail -5 | tee -a $OUT/progress.log
[ $RC -ne 0 ] && exit $RC
for round in 1 2 3; do
  timeout -k 10 200 python3 bench.py --wideband --steps 40 --warmup 5 > $OUT/wb_$round.json 2> $OUT/wb_$round.err; RC=$?
  python3 - $OUT/wb_$round.json <<'PY' | tee -a $OUT/progress.log
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("value", d["value"], "ms_per_step", d["ms_per_step"], "channelize_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], d.get("packets_recovered"))
except Exception as e:
    print("FAILED", e)
PY
  [ $RC -ge 124 ] && exit $RC
done
echo "[r3ab] done" | tee -a $OUT/progress.log
