#!/bin/bash
# round 3, twenty-first GPU call: the whole GPU suite under each opt-in form (they must be transparent)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3v
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
export GPU_FORCE_BLIT_COPY_SIZE=0
for v in "fused:RD_SEARCH_IMPL=fused" "self:RD_FIXUP_IMPL=self" "legacy_tail:RD_TAIL_IMPL=legacy" "stream_legacy:RD_STREAM_IMPL=legacy"; do
  name=${v%%:*}; e=${v#*:}
  echo "[r3v] pytest -m gpu under $e" | tee -a $OUT/progress.log
  env $e timeout -k 10 500 python -m pytest tests -m gpu -q > $OUT/pytest_$name.log 2>&1; RC=$?
  tail -12 $OUT/pytest_$name.log | cut -c1-200 | tee -a $OUT/progress.log
  [ $RC -ge 124 ] && exit $RC
done
echo "[r3v] done" | tee -a $OUT/progress.log
