#!/bin/bash
# clocks / power while the bench loop runs (is the chip throttling under k_mm8?)
rocm-smi --showclocks --showpower --showmaxpower 2>&1 | grep -E "sclk|mclk|Power|power" | head -8
python bench.py --steps 150000 --warmup 10 --cpu-sample 0 --no-two-streams-extra > /tmp/b.log 2>&1 &
PID=$!
sleep 20
for i in 1 2 3 4; do
  rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -E "sclk|Power|junction|hotspot|edge" | head -6
  echo ---
  sleep 1
done
wait $PID
tail -c 300 /tmp/b.log
