#!/bin/bash
# round 3, twenty-fourth GPU call: bench.py measuring roofline.traffic itself (two child runs under rocprofv3 --pmc);
# the same line under an outer rocprofv3 --kernel-trace (the live leg must step aside)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3y
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
export GPU_FORCE_BLIT_COPY_SIZE=0
echo "[r3y] default bench" | tee -a $OUT/progress.log
T0=$(date +%s); timeout -k 10 500 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; RC=$?
echo "wall seconds: $(( $(date +%s) - T0 ))" | tee -a $OUT/progress.log
python3 - $OUT/bench_default.json <<'PY' | tee -a $OUT/progress.log
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "roofline", d["roofline"])
PY
[ $RC -ge 124 ] && exit $RC
echo "[r3y] driver-style flags" | tee -a $OUT/progress.log
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; RC=$?
python3 - $OUT/bench_driver.json <<'PY' | tee -a $OUT/progress.log
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], d["roofline"].get("traffic_source", "")[:40], "sustained", d.get("sustained", {}).get("value"))
PY
[ $RC -ge 124 ] && exit $RC
cd /tmp && export TMPDIR=/tmp
echo "[r3y] under an outer rocprofv3 --kernel-trace" | tee -a $OUT/progress.log
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline --steps 20 --sustain 0 > $OUT/bench_nested.json 2> $OUT/nested.err; RC=$?
python3 - $OUT/bench_nested.json <<'PY' | tee -a $OUT/progress.log
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("nested: traffic", d["roofline"]["traffic"], d["roofline"].get("traffic_source"), d["roofline"].get("traffic_note"))
PY
echo "[r3y] done rc $RC" | tee -a $OUT/progress.log
