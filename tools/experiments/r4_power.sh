#!/bin/bash
# what the chip reports (clock, power, temperature) while the default bench's sustained leg runs
mkdir -p gpurun_out/r4power
python3 bench.py --no-cpu-baseline --live-traffic 0 --sustain 12 > gpurun_out/r4power/bench.json 2> gpurun_out/r4power/bench.err &
BP=$!
sleep 6
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks --showtemp --showperflevel -d 0 2>&1 | grep -E "Power|sclk|mclk|fclk|socclk|Temperature|Performance" | tr -s ' ' | head -14
  echo "--"
  sleep 1.5
done
wait $BP
tail -c 400 gpurun_out/r4power/bench.json
rocm-smi --showmaxpower -d 0 2>&1 | grep -i "power" | head -3
