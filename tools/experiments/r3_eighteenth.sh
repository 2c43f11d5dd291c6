#!/bin/bash
# round 3, eighteenth GPU call: the fused search (RD_SEARCH_IMPL=fused) - GPU suite, bench A/B, kernel trace
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3s
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
export GPU_FORCE_BLIT_COPY_SIZE=0
echo "[r3s] pytest (fused-search test first)" | tee -a $OUT/progress.log
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused" > $OUT/pytest_fused.log 2>&1; RC=$?
tail -30 $OUT/pytest_fused.log | tee -a $OUT/progress.log
[ $RC -ne 0 ] && exit $RC
echo "[r3s] pytest, batch tests under RD_SEARCH_IMPL=fused" | tee -a $OUT/progress.log
RD_SEARCH_IMPL=fused timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batch or ordered or full_size or two_bursts or dedupe or edge or degenerate or sharded" > $OUT/pytest_env.log 2>&1; RC=$?
tail -8 $OUT/pytest_env.log | tee -a $OUT/progress.log
[ $RC -ne 0 ] && exit $RC
echo "[r3s] pytest (whole suite, default)" | tee -a $OUT/progress.log
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; RC=$?
tail -5 $OUT/pytest.log | tee -a $OUT/progress.log
[ $RC -ne 0 ] && exit $RC
for v in "default:" "fused:RD_SEARCH_IMPL=fused" "default2:" "fused2:RD_SEARCH_IMPL=fused"; do
  name=${v%%:*}; e=${v#*:}
  env $e timeout -k 10 300 python3 bench.py --no-cpu-baseline --sustain 3 > $OUT/bench_$name.json 2> $OUT/bench_$name.err; RC=$?
  python3 - $OUT/bench_$name.json $name <<'PY' | tee -a $OUT/progress.log
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "value", d["value"], "ms_per_step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"],
          "total", d["kernels_ms"].get("total"), "sustained", d.get("sustained", {}).get("value"), d.get("sustained", {}).get("roofline_frac"), "packets", d["packets_per_step"], "verified", d["verified_vs_reference_fixtures"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  [ $RC -ge 124 ] && exit $RC
done
cd /tmp && export TMPDIR=/tmp
echo "[r3s] kernel trace (fused)" | tee -a $OUT/progress.log
RD_SEARCH_IMPL=fused timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-verify --sustain 0 > $OUT/bench_trace.json 2> $OUT/trace.err; RC=$?
python3 $ROOT/tools/profile_collect.py stats $OUT/trace $OUT > /dev/null
grep -v "first 12\|last 12\|copyBuffer" $OUT/kernel_durations.txt | cut -c1-120 | tee -a $OUT/progress.log
echo "[r3s] done rc $RC" | tee -a $OUT/progress.log
