#!/bin/bash
# Re-collects the decoder-side artefacts of collect_r04.sh (and the cfg4 bench line) into the same gpurun_out/<tag>/:   profiles/collect_r04_part2.sh r04s
# (the first pass of r04s ran them against a library one commit behind its Python: every robust-phase row failed on an argument check)
set -u
TAG="${1:-r04s}"
OUT="gpurun_out/$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
exec < /dev/null
timeout 900 python scratch/bench_device_decoder.py > "$OUT/device_decoder.txt" 2>&1
timeout 900 python scratch/bench_device_decoder.py --copy > "$OUT/device_decoder_copied_columns.txt" 2>&1
timeout 600 python scratch/decoder_cfg5_shape.py > "$OUT/decoder_cfg5_shape.txt" 2>&1
timeout 300 python scratch/dec21.py > "$OUT/dec21_cfg3.txt" 2>&1
timeout 300 python scratch/dec21.py 256 85 > "$OUT/dec21_cfg5.txt" 2>&1
timeout 300 python scratch/dec21.py 64 21 spread freeze > "$OUT/dec21_cfg3_spread.txt" 2>&1
timeout 300 python scratch/dec21.py 256 85 spread freeze > "$OUT/dec21_cfg5_spread.txt" 2>&1
timeout 300 python scratch/time_fetch.py > "$OUT/symbols_fetch.txt" 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/stats_dec21" -o run -- python scratch/dec21.py 256 85 > "$OUT/stats_dec21.log" 2>&1
python profiles/summarize_rocpd.py "$OUT/stats_dec21/run_results.db" > "$OUT/kernel_stats_dec21_cfg5.txt" 2>&1
timeout 600 python scratch/time_first_sight.py > "$OUT/first_sight_host_phases.txt" 2>&1
timeout 900 python scratch/bench_coalescer.py > "$OUT/coalescer.txt" 2>&1
timeout 400 python scratch/stress_decoder.py 240 31 > "$OUT/stress_decoder.txt" 2>&1
rm -rf "$OUT"/stats_dec21
tail -3 "$OUT/decoder_cfg5_shape.txt"; tail -1 "$OUT/stress_decoder.txt"
