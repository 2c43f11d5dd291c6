# does giving each resident batch its own HIP stream AND hardware queue let the tail kernels of
# batch i run beside the demod kernel of batch i+1?  (K1 grid reduced so that slots stay free)
set -e
for q in 4 8; do
for w in 7 5 4; do
  for ts in "" "--two-streams"; do
    echo "== GPU_MAX_HW_QUEUES=$q WGS=$w $ts"
    GPU_MAX_HW_QUEUES=$q RD_K1_WGS_PER_CU=$w timeout -k 10 120 python bench.py --no-cpu-baseline --steps 60 --warmup 5 $ts 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['kernels_ms'])"
  done
done
done
