#!/bin/bash
# sample sclk / power with rocm-smi while the real training step runs (is the chip power-limited under the step?)
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.5; done ) > gpurun_out/clock_watch.txt &
W=$!
python bench.py --no-cpu-baseline --no-via-trainer --no-roofline --steps 600 --warmup 20 2>/dev/null | tail -1
wait $W
cat gpurun_out/clock_watch.txt | cut -c1-200 | awk 'NR%2==0' | head -24
