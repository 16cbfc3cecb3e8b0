#!/bin/bash
# Sample socket power / shader clock while bench.py runs (evidence for the power-limited clock): bash tools/power_sample.sh
python bench.py --steps 600 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/pw_bench.log 2>&1 &
BP=$!
for i in $(seq 1 26); do
  sleep 1
  echo "t=$i $(rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E 'sclk|Package Power|GPU use' | sed 's/.*: //' | tr '\n' ' ')"
done > gpurun_out/pw_smi.log 2>&1
wait $BP
tail -1 gpurun_out/pw_bench.log | cut -c1-160
cat gpurun_out/pw_smi.log
