#!/bin/bash
# round 3, tenth GPU call: pipelined completion (rd_batch_set_pipelined) against the per-run event, the in-tile search
# probe (RD_OPT_FPROBE: what fusing k_search into the demod kernel would cost), bench.py launching its own ranks
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3k
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
export GPU_FORCE_BLIT_COPY_SIZE=0
DIAG=$ROOT/rtldavis_amd/librtldavis_hip_diag.so
echo "[r3k] pytest" | tee -a $OUT/progress.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; RC=$?
tail -30 $OUT/pytest.log | tee -a $OUT/progress.log
[ $RC -ne 0 ] && exit $RC
echo "[r3k] A/B: in-tile search probe" | tee -a $OUT/progress.log
timeout -k 10 400 python3 tools/k1_ab.py --key all --rounds 3 b8=RD_AB_TIMING=1,RD_K1_OPT=10 b8_probe=RD_AB_TIMING=1,RD_K1_OPT=26 > $OUT/ab_probe.txt 2>&1; RC=$?
cat $OUT/ab_probe.txt | tee -a $OUT/progress.log
[ $RC -ge 124 ] && exit $RC
echo "[r3k] bench: completion per run / pipelined" | tee -a $OUT/progress.log
for v in "perrun2:--pipelined 0 --resident 2" "pipe2:--pipelined 1 --resident 2" "pipe3:--pipelined 1 --resident 3" "perrun3:--pipelined 0 --resident 3"; do
  name=${v%%:*}; flags=${v#*:}
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --sustain 3 --steps 100 $flags > $OUT/bench_$name.json 2> $OUT/bench_$name.err; RC=$?
  python3 - $OUT/bench_$name.json $name <<'PY' | tee -a $OUT/progress.log
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "value", d["value"], "ms_per_step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"],
          "total", d["kernels_ms"].get("total"), "sustained", d.get("sustained", {}).get("value"), d.get("sustained", {}).get("ms_per_step"), "verified", d["verified_vs_reference_fixtures"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  [ $RC -ge 124 ] && exit $RC
done
echo "[r3k] bench.py --gpus 2 --rehearse-shared-gpu (self-launch)" | tee -a $OUT/progress.log
timeout -k 10 300 python3 bench.py --gpus 2 --rehearse-shared-gpu --steps 5 --warmup 2 --no-cpu-baseline --sustain 0 --streams 1024 > $OUT/bench_rehearse.json 2> $OUT/bench_rehearse.err; RC=$?
echo "rc $RC" | tee -a $OUT/progress.log
tail -c 600 $OUT/bench_rehearse.json | tee -a $OUT/progress.log
tail -3 $OUT/bench_rehearse.err | tee -a $OUT/progress.log
[ $RC -ge 124 ] && exit $RC
echo "[r3k] kernel trace" | tee -a $OUT/progress.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-verify --sustain 0 --pipelined 1 --resident 3 > $OUT/bench_trace.json 2> $OUT/trace.err; RC=$?
python3 $ROOT/tools/profile_collect.py stats $OUT/trace $OUT > /dev/null
grep -v "first 12\|last 12\|copyBuffer" $OUT/kernel_durations.txt | cut -c1-120 | tee -a $OUT/progress.log
[ $RC -ge 124 ] && exit $RC
echo "[r3k] SQ counters: b8 / b8 + probe" | tee -a $OUT/progress.log
cd $ROOT
RTLDAVIS_HIP_LIB=$DIAG RD_K1_OPT=10 bash tools/pmc_k1.sh r3k_b8 > $OUT/pmc_b8.csv 2>&1; RC=$?
[ $RC -ge 124 ] && exit $RC
RTLDAVIS_HIP_LIB=$DIAG RD_K1_OPT=26 bash tools/pmc_k1.sh r3k_probe > $OUT/pmc_probe.csv 2>&1; RC=$?
grep "k_demod_mfma" $OUT/pmc_b8.csv | cut -c1-140 | tee -a $OUT/progress.log
grep "k_demod_mfma" $OUT/pmc_probe.csv | cut -c1-140 | tee -a $OUT/progress.log
echo "[r3k] done rc $RC" | tee -a $OUT/progress.log
