#!/bin/bash
# round 4: long randomised soaks against the C oracle under the one-launch tail (default) and the separate kernels
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r4soak}
mkdir -p $OUT
cd $ROOT
{
echo "== soak.py 1500 51 (random small configs, batch + streaming)"
timeout -k 10 400 python3 tools/soak.py 1500 51 2>&1 | tail -2
echo "== soak.py bursts 6000 19 (default form: k_tail)"
timeout -k 10 300 python3 tools/soak.py bursts 6000 19 2>&1 | tail -2
echo "== soak.py bursts 6000 20 (default form, second seed)"
timeout -k 10 300 python3 tools/soak.py bursts 6000 20 2>&1 | tail -2
echo "== RD_TAIL_IMPL=legacy soak.py bursts 6000 21"
RD_TAIL_IMPL=legacy timeout -k 10 300 python3 tools/soak.py bursts 6000 21 2>&1 | tail -2
echo "== RD_TEST_BUCKET_CAP=3 soak.py bursts 3000 22 (every stream overflows its match list: the fallback)"
RD_TEST_BUCKET_CAP=3 timeout -k 10 300 python3 tools/soak.py bursts 3000 22 2>&1 | tail -2
echo "== RD_TEST_FIX_BCAP=8 soak.py bursts 3000 23 (every group overflows its fix-up bucket: the fallback)"
RD_TEST_FIX_BCAP=8 timeout -k 10 300 python3 tools/soak.py bursts 3000 23 2>&1 | tail -2
} | tee $OUT/soak.txt
