#!/bin/bash
# round 3, fifteenth GPU call: the batch path on quiet / loud / constant inputs (tests/tool_input_classes.py: it uses the C oracle as the checker, so it lives with the tests)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3p
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
export GPU_FORCE_BLIT_COPY_SIZE=0
timeout -k 10 800 python3 tests/tool_input_classes.py 2>&1 | tee $OUT/input_classes.txt
echo "[r3p] done" | tee -a $OUT/progress.log
