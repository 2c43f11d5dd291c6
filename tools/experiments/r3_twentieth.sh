#!/bin/bash
# round 3, twentieth GPU call: long randomised soaks against the C oracle - small random configurations (batch and
# streaming), production-shape bursts on block boundaries under the default form and the opt-in forms
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3u
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
export GPU_FORCE_BLIT_COPY_SIZE=0
( echo "== soak.py 3000 41 (random small configs, batch + streaming)"; timeout -k 10 500 python3 tools/soak.py 3000 41 2>&1 | tail -4 ) | tee -a $OUT/soak.txt
( echo "== soak.py bursts 6000 9 (default form)"; timeout -k 10 300 python3 tools/soak.py bursts 6000 9 2>&1 | tail -2 ) | tee -a $OUT/soak.txt
( echo "== RD_SEARCH_IMPL=fused soak.py bursts 6000 10"; RD_SEARCH_IMPL=fused timeout -k 10 300 python3 tools/soak.py bursts 6000 10 2>&1 | tail -2 ) | tee -a $OUT/soak.txt
( echo "== RD_FIXUP_IMPL=self soak.py bursts 6000 11"; RD_FIXUP_IMPL=self timeout -k 10 300 python3 tools/soak.py bursts 6000 11 2>&1 | tail -2 ) | tee -a $OUT/soak.txt
( echo "== RD_TAIL_IMPL=legacy soak.py bursts 6000 12"; RD_TAIL_IMPL=legacy timeout -k 10 300 python3 tools/soak.py bursts 6000 12 2>&1 | tail -2 ) | tee -a $OUT/soak.txt
echo "[r3u] done" | tee -a $OUT/progress.log
