#!/bin/bash
# round 3, twenty-sixth GPU call: k_search - wave-groups per trip (RD_SEARCH_UNROLL 1 / 2 product / 3) and persistent
# workgroups per CU (RD_K2_WGS_PER_CU 6 / 8 product / 10 / 12)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3aa
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GPU_FORCE_BLIT_COPY_SIZE=0
for v in "u2:" "u1:RTLDAVIS_HIP_LIB=$ROOT/tools/ab_libs/search_u1.so" "u3:RTLDAVIS_HIP_LIB=$ROOT/tools/ab_libs/search_u3.so" "cap6:RD_K2_WGS_PER_CU=6" "cap10:RD_K2_WGS_PER_CU=10" "cap12:RD_K2_WGS_PER_CU=12" "u2b:"; do
  name=${v%%:*}; e=${v#*:}
  [ -n "$e" ] && export $e
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$name -- python3 $ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-verify --sustain 0 --live-traffic 0 > $OUT/bench_$name.json 2> $OUT/trace_$name.err; RC=$?
  [ -n "$e" ] && unset ${e%%=*}
  mkdir -p $OUT/$name; python3 $ROOT/tools/profile_collect.py stats $OUT/trace_$name $OUT/$name > /dev/null
  echo "$name $(grep 'k_search' $OUT/$name/kernel_durations.txt | cut -c1-110)" | tee -a $OUT/progress.log
  [ $RC -ge 124 ] && exit $RC
done
echo "[r3aa] done" | tee -a $OUT/progress.log
