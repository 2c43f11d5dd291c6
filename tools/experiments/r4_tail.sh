#!/bin/bash
# round 4: the one-launch tail - its tests, its phase stamps (diagnostic library), the default bench line
set -o pipefail
TAG=${1:-r4tail}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "one_launch or long_streams or batch_streams or parse_front or api_call or full_size" > $OUT/pytest.log 2>&1; RC=$?
tail -15 $OUT/pytest.log | tee -a $OUT/progress.log
[ $RC -ne 0 ] && exit $RC
RTLDAVIS_HIP_LIB=$ROOT/rtldavis_amd/librtldavis_hip_diag.so RD_FT_STAMPS=1 timeout -k 10 300 python3 tools/tail_stamps.py 2>&1 | tee $OUT/stamps.txt
timeout -k 10 500 python3 bench.py --no-cpu-baseline --live-traffic 0 > $OUT/bench_default.json 2> $OUT/bench_default.err; RC=$?
python3 - $OUT/bench_default.json <<'PY' | tee -a $OUT/progress.log
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"], "kernel_ms", d["roofline"]["kernel_ms"], "sustained", d.get("sustained", {}).get("value"), d["verified_vs_reference_fixtures"])
print("kernels_ms", d.get("kernels_ms"))
PY
exit $RC
