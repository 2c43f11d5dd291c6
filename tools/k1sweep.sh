timeout -k 10 200 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for n in 0 2 4 5 6 7 9; do
  echo "== RD_K1_NPK=$n"
  RD_K1_NPK=$n timeout -k 10 120 python tools/step_profile.py 4096 2>&1 | grep demod_ms
done
echo "== DEBUG=2 (loads only)"; RD_K1_DEBUG=2 timeout -k 10 120 python tools/step_profile.py 4096 2>&1 | grep demod_ms
echo "== DEBUG=1 (compute only)"; RD_K1_DEBUG=1 timeout -k 10 120 python tools/step_profile.py 4096 2>&1 | grep demod_ms
