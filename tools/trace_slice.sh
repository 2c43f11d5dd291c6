cd /tmp && export TMPDIR=/tmp
export GPU_FORCE_BLIT_COPY_SIZE=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_slice
rm -rf $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 100 --warmup 5 --sustain 0 > $OUT.log 2>&1
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
cut -c1-60 $F | paste -d, - <(cut -d, -f2- $F | rev | cut -d, -f1-7 | rev) | head -12
