#!/bin/bash
# samples rocm-smi power / clocks while the headline bench runs (is the slow drift of kernel durations a clock effect?)
mkdir -p gpurun_out
python bench.py --steps 6 --warmup 1 --no-cpu-baseline > gpurun_out/power_bench.log 2>&1 &
BP=$!
sleep 6
for i in $(seq 1 14); do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "GPU\[0\].*(Power|sclk|mclk|fclk|socclk|Temperature \(Sensor (junction|memory))" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ';'
  echo
  sleep 0.5
done
wait $BP
tail -1 gpurun_out/power_bench.log | cut -c1-200
