#!/bin/bash
# longer runs of the four randomised differential scripts (through gpurun): scratch/stress_long.sh [seconds each] -> gpurun_out/stress_long/
cd "$GRAFT_REPO_ROOT"; exec < /dev/null
T="${1:-240}"; OUT=gpurun_out/stress_long; mkdir -p $OUT
timeout $((T + 120)) python scratch/stress_decoder.py $T 161 > $OUT/stress_decoder.txt 2>&1
timeout $((T + 120)) python scratch/stress_gao.py $T 162 > $OUT/stress_gao.txt 2>&1
timeout $((T + 120)) python scratch/stress_open_paths.py $T 163 > $OUT/stress_open_paths.txt 2>&1
timeout $((T + 120)) python scratch/stress_narrow.py $T 164 > $OUT/stress_narrow.txt 2>&1
tail -n 2 $OUT/*.txt
