#!/bin/bash
# round 3, thirteenth GPU call: tail kernels after the unrolled dedupe / rank loops of k_classify_ord, the speculative
# first task of k_rssi_ord and the fix-up grid sized from the last run; k_search with 8 output words per lane
# (tools/ab_libs/search8.so = the product sources with -DRD_SEARCH_OUT=8) against 4
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3n
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
export GPU_FORCE_BLIT_COPY_SIZE=0
echo "[r3n] pytest" | tee -a $OUT/progress.log
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; RC=$?
tail -30 $OUT/pytest.log | tee -a $OUT/progress.log
[ $RC -ne 0 ] && exit $RC
echo "[r3n] pytest, batch tests with 8 search words per lane" | tee -a $OUT/progress.log
RTLDAVIS_HIP_LIB=$ROOT/tools/ab_libs/search8.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batch or ordered or random or soak or search or full_size" > $OUT/pytest8.log 2>&1; RC=$?
tail -5 $OUT/pytest8.log | tee -a $OUT/progress.log
[ $RC -ge 124 ] && exit $RC
cd /tmp && export TMPDIR=/tmp
for v in out4 out8 out4b out8b; do
  echo "[r3n] kernel trace $v" | tee -a $OUT/progress.log
  case $v in out8*) export RTLDAVIS_HIP_LIB=$ROOT/tools/ab_libs/search8.so;; *) unset RTLDAVIS_HIP_LIB;; esac
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$v -- python3 $ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-verify --sustain 0 > $OUT/bench_trace_$v.json 2> $OUT/trace_$v.err; RC=$?
  mkdir -p $OUT/$v
  python3 $ROOT/tools/profile_collect.py stats $OUT/trace_$v $OUT/$v > /dev/null
  grep -v "first 12\|last 12\|copyBuffer" $OUT/$v/kernel_durations.txt | cut -c1-120 | tee -a $OUT/progress.log
  [ $RC -ge 124 ] && exit $RC
done
unset RTLDAVIS_HIP_LIB
cd $ROOT
echo "[r3n] bench" | tee -a $OUT/progress.log
timeout -k 10 300 python3 bench.py --no-cpu-baseline --sustain 3 > $OUT/bench.json 2> $OUT/bench.err; RC=$?
tail -c 1300 $OUT/bench.json | tee -a $OUT/progress.log
RTLDAVIS_HIP_LIB=$ROOT/tools/ab_libs/search8.so timeout -k 10 300 python3 bench.py --no-cpu-baseline --sustain 3 > $OUT/bench8.json 2> $OUT/bench8.err; RC=$?
tail -c 1300 $OUT/bench8.json | tee -a $OUT/progress.log
echo "[r3n] done rc $RC" | tee -a $OUT/progress.log
