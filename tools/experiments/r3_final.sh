#!/bin/bash
# round 3, last GPU call: smoke(), the GPU suite and the default bench line on the final tree
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3final
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a $OUT/progress.log
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; RC=$?
tail -4 $OUT/pytest.log | tee -a $OUT/progress.log
[ $RC -ne 0 ] && exit $RC
timeout -k 10 500 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; RC=$?
python3 - $OUT/bench_default.json <<'PY' | tee -a $OUT/progress.log
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"], "kernel_ms", d["roofline"]["kernel_ms"], "traffic", d["roofline"]["traffic"], d["roofline"].get("traffic_source", "")[:30], "sustained", d.get("sustained", {}).get("value"), "cpu", d["cpu_baseline"]["value"], d["verified_vs_reference_fixtures"])
PY
echo "[r3final] done rc $RC" | tee -a $OUT/progress.log
