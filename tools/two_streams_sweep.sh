# does giving each resident batch its own HIP stream let the tail kernels of batch i run beside the
# demod kernel of batch i+1?  (demod grid reduced so that slots stay free)
for w in 4 3 2; do
  for ts in "" "--two-streams"; do
    echo "== WGS=$w $ts"
    RD_K1_WGS_PER_CU=$w timeout -k 10 120 python bench.py --no-cpu-baseline --steps 60 --warmup 5 --sustain 0 $ts 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d.get('kernels_ms'))"
  done
done
