#!/bin/bash
# round 3, seventeenth GPU call: the self-fix variant as its own kernel - GPU suite, default bench line (traffic from
# profiles/r03_traffic.json must be accepted again), bench --self-fix
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3r
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
export GPU_FORCE_BLIT_COPY_SIZE=0
echo "[r3r] pytest" | tee -a $OUT/progress.log
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; RC=$?
tail -30 $OUT/pytest.log | tee -a $OUT/progress.log
[ $RC -ne 0 ] && exit $RC
for v in "default:" "selffix:--self-fix" "default2:" "selffix2:--self-fix"; do
  name=${v%%:*}; f=${v#*:}
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --sustain 3 $f > $OUT/bench_$name.json 2> $OUT/bench_$name.err; RC=$?
  python3 - $OUT/bench_$name.json $name <<'PY' | tee -a $OUT/progress.log
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "value", d["value"], "ms_per_step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], d["roofline"].get("traffic_note", "")[:40],
          "total", d["kernels_ms"].get("total"), "sustained", d.get("sustained", {}).get("value"), d.get("sustained", {}).get("roofline_frac"), d.get("fixup"), "verified", d["verified_vs_reference_fixtures"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  [ $RC -ge 124 ] && exit $RC
done
echo "[r3r] done rc $RC" | tee -a $OUT/progress.log
