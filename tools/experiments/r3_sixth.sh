#!/bin/bash
# round 3, sixth GPU call: per-kernel times of the ordered and the unordered tail, placement of the batch buffers
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3f
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GPU_FORCE_BLIT_COPY_SIZE=0
for V in ordered legacy; do
  if [ $V = legacy ]; then export RD_TAIL_IMPL=legacy; else unset RD_TAIL_IMPL; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$V -- python3 $ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-verify --sustain 0 > $OUT/bench_$V.json 2> $OUT/trace_$V.err; RC=$?
  echo "[r3f] trace $V rc $RC" | tee -a $OUT/progress.log
  [ $RC -ge 124 ] && exit $RC
  python3 $ROOT/tools/profile_collect.py stats $OUT/trace_$V $OUT ${V}_ | tee -a $OUT/progress.log
done
unset RD_TAIL_IMPL
cd $ROOT
echo "[r3f] placement, whole kernel" | tee -a $OUT/progress.log
timeout -k 10 300 python3 tools/placement.py 8 2>&1 | tee $OUT/placement_full.txt | tee -a $OUT/progress.log
echo "[r3f] done" | tee -a $OUT/progress.log
