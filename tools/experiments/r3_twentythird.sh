#!/bin/bash
# round 3, twenty-third GPU call: smoke(), the GPU suite, the wideband line and its kernel stats on the final code
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3x
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
export GPU_FORCE_BLIT_COPY_SIZE=0
echo "[r3x] smoke" | tee -a $OUT/progress.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee -a $OUT/progress.log
echo "[r3x] pytest" | tee -a $OUT/progress.log
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; RC=$?
tail -5 $OUT/pytest.log | tee -a $OUT/progress.log
[ $RC -ne 0 ] && exit $RC
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $ROOT/bench.py --wideband --steps 20 --warmup 3 > $OUT/bench_wideband.json 2> $OUT/bench_wideband.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_wb -- python3 $ROOT/bench.py --wideband --steps 20 --warmup 3 > $OUT/bench_wideband_under_rocprof.json 2> $OUT/trace_wb.err
python3 $ROOT/tools/profile_collect.py stats $OUT/trace_wb $OUT wideband_ | tee -a $OUT/progress.log
tail -c 700 $OUT/bench_wideband.json | tee -a $OUT/progress.log
echo "[r3x] done" | tee -a $OUT/progress.log
