#!/bin/bash
# round 4: streaming handle - tests of the complex one-launch block and the host-wait deadlines, latency of both input forms
set -o pipefail
TAG=${1:-r4stream}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "complex or mixed or demodulator or submit_fetch or worker or deadline or multi_demod or one_launch_block or zero_signal or zero_copy" > $OUT/pytest.log 2>&1; RC=$?
tail -15 $OUT/pytest.log | tee -a $OUT/progress.log
[ $RC -ne 0 ] && exit $RC
timeout -k 10 300 python3 tools/stream_latency.py 2>&1 | tee $OUT/stream_latency.txt
echo "== RD_STREAM_IMPL=legacy" | tee -a $OUT/stream_latency.txt
RD_STREAM_IMPL=legacy timeout -k 10 300 python3 tools/stream_latency.py 2>&1 | grep -E "complex|^demodulate" | tee -a $OUT/stream_latency.txt
echo "== worker: queue hop against the shared-memory ring (tools/worker_rate.py)" | tee -a $OUT/stream_latency.txt
timeout -k 10 300 python3 tools/worker_rate.py 2>&1 | grep -v "DSP worker\|stop signal" | tee -a $OUT/stream_latency.txt
