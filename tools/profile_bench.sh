# rocprofv3 kernel trace + stats of the bench command; summary copied to profiles/ by hand
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$1
rm -rf $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT.log 2>&1
tail -1 $OUT.log | cut -c1-600
find $OUT -name "*kernel_stats.csv" | head -1 | xargs cat | head -12
