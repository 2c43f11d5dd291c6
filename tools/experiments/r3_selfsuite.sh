#!/bin/bash
# the GPU suite under RD_FIXUP_IMPL=self and under RD_SEARCH_IMPL=fused once more (last host-side change)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3self
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
for e in RD_FIXUP_IMPL=self RD_SEARCH_IMPL=fused; do
  env $e timeout -k 10 500 python -m pytest tests -m gpu -q > $OUT/pytest_$e.log 2>&1
  echo "$e: $(tail -1 $OUT/pytest_$e.log)" | tee -a $OUT/progress.log
done
timeout -k 10 500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "default: $(tail -1 $OUT/pytest.log)" | tee -a $OUT/progress.log
