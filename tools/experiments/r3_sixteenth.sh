#!/bin/bash
# round 3, sixteenth GPU call: self-fix (the demod kernel's waves re-evaluate their own flagged groups, no k_fixup
# launch) - GPU suite, then A/B against RD_FIXUP_IMPL=kernel, kernel trace, bench
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3q
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
export GPU_FORCE_BLIT_COPY_SIZE=0
echo "[r3q] pytest" | tee -a $OUT/progress.log
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; RC=$?
tail -30 $OUT/pytest.log | tee -a $OUT/progress.log
[ $RC -ne 0 ] && exit $RC
echo "[r3q] A/B" | tee -a $OUT/progress.log
RD_AB_TIMING=1 timeout -k 10 500 python3 tools/k1_ab.py --key all --rounds 3 self_fix=RD_K1_OPT=10 k_fixup=RD_FIXUP_IMPL=kernel self_fix_legacy_tail=RD_TAIL_IMPL=legacy k_fixup_legacy_tail=RD_FIXUP_IMPL=kernel,RD_TAIL_IMPL=legacy > $OUT/ab_selffix.txt 2>&1; RC=$?
cat $OUT/ab_selffix.txt | tee -a $OUT/progress.log
[ $RC -ge 124 ] && exit $RC
cd /tmp && export TMPDIR=/tmp
echo "[r3q] kernel trace" | tee -a $OUT/progress.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-verify --sustain 0 > $OUT/bench_trace.json 2> $OUT/trace.err; RC=$?
python3 $ROOT/tools/profile_collect.py stats $OUT/trace $OUT > /dev/null
grep -v "first 12\|last 12\|copyBuffer" $OUT/kernel_durations.txt | cut -c1-120 | tee -a $OUT/progress.log
[ $RC -ge 124 ] && exit $RC
cd $ROOT
echo "[r3q] bench" | tee -a $OUT/progress.log
for v in "self:" "kfix:RD_FIXUP_IMPL=kernel" "self2:" "kfix2:RD_FIXUP_IMPL=kernel"; do
  name=${v%%:*}; e=${v#*:}
  env $e timeout -k 10 300 python3 bench.py --no-cpu-baseline --sustain 3 > $OUT/bench_$name.json 2> $OUT/bench_$name.err; RC=$?
  python3 - $OUT/bench_$name.json $name <<'PY' | tee -a $OUT/progress.log
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "value", d["value"], "ms_per_step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"],
          "total", d["kernels_ms"].get("total"), "sustained", d.get("sustained", {}).get("value"), d.get("sustained", {}).get("roofline_frac"), "fix frac", d["fixup_runs_frac"], "verified", d["verified_vs_reference_fixtures"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  [ $RC -ge 124 ] && exit $RC
done
echo "[r3q] done rc $RC" | tee -a $OUT/progress.log
