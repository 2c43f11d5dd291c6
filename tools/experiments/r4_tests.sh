#!/bin/bash
# round 4: the GPU suite (optionally a -k selection), then the default bench line
set -o pipefail
TAG=${1:-r4t}
SEL=${2:-}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
if [ -n "$SEL" ]; then
  timeout -k 10 700 python -m pytest tests -m gpu -x -q -k "$SEL" > $OUT/pytest.log 2>&1; RC=$?
else
  timeout -k 10 700 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; RC=$?
fi
tail -30 $OUT/pytest.log | tee -a $OUT/progress.log
[ $RC -ne 0 ] && exit $RC
timeout -k 10 500 python3 bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err; RC=$?
tail -5 $OUT/bench_default.err
python3 - $OUT/bench_default.json <<'PY' | tee -a $OUT/progress.log
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"], "kernel_ms", d["roofline"]["kernel_ms"], "traffic", d["roofline"]["traffic"], "sustained", d.get("sustained", {}).get("value"), d["verified_vs_reference_fixtures"])
print("kernels_ms", d.get("kernels_ms"))
PY
if [ -n "$3" ]; then
  cd /tmp && export TMPDIR=/tmp && export GPU_FORCE_BLIT_COPY_SIZE=0
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline --live-traffic 0 --sustain 0 > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
  python3 $ROOT/tools/profile_collect.py stats $OUT/trace $OUT | tee -a $OUT/progress.log
  cd $ROOT
fi
echo "[$TAG] done rc $RC" | tee -a $OUT/progress.log
exit $RC
