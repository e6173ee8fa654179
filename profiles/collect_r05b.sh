#!/bin/bash
# Round-5 artefacts re-collected after the probe's rebuild, k_eval_few and k_gao_pair (run through gpurun; outputs under gpurun_out/<tag>/):
#   profiles/collect_r05b.sh r05b
# the device decoder at first sight and under attack, the four dec21 opens, the probe's kernel trace, the other workloads' lines (their
# under-attack figures), and longer stress runs of the decoder (n up to 256: the probe's six-workgroup launches) and of the robust decoders
set -u
TAG="${1:-r05b}"
OUT="gpurun_out/$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
exec < /dev/null
timeout 900 python scratch/bench_device_decoder.py > "$OUT/device_decoder.txt" 2>&1
timeout 600 python scratch/decoder_cfg5_shape.py > "$OUT/decoder_cfg5_shape.txt" 2>&1
timeout 300 python scratch/dec21.py > "$OUT/dec21_cfg3.txt" 2>&1
timeout 300 python scratch/dec21.py 256 85 > "$OUT/dec21_cfg5.txt" 2>&1
timeout 300 python scratch/dec21.py 64 21 spread freeze > "$OUT/dec21_cfg3_spread.txt" 2>&1
timeout 300 python scratch/dec21.py 256 85 spread freeze > "$OUT/dec21_cfg5_spread.txt" 2>&1
for w in 1 2 3 4 5 6 7 8; do echo "HB_PROBE_WGS=$w $(HB_PROBE_WGS=$w timeout 300 python scratch/dec21.py 256 85 spread freeze 2>&1 | tail -1 | cut -c1-70)"; done > "$OUT/probe_workgroups_sweep.txt" 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_probe" -o run -- python scratch/dec21.py 256 85 spread freeze > /dev/null 2>&1
timeout 300 python profiles/summarize_rocpd.py "$OUT/prof_probe/run_results.db" > "$OUT/kernel_stats_dec21_cfg5_spread.txt" 2>&1
rm -rf "$OUT/prof_probe"
for w in cfg5-shard cfg3-omega cfg2 cfg5; do
  timeout 600 python bench.py --workload $w --steps 30 --warmup 5 --cpu-sample 0 --no-two-streams-extra > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err"
done
timeout 500 python scratch/stress_decoder.py 400 61 > "$OUT/stress_decoder.txt" 2>&1
timeout 300 python scratch/stress_gao.py 200 62 > "$OUT/stress_gao.txt" 2>&1
timeout 300 python scratch/stress_open_paths.py 90 63 > "$OUT/stress_open_paths.txt" 2>&1
tail -n 3 "$OUT/stress_decoder.txt" "$OUT/stress_gao.txt" "$OUT/stress_open_paths.txt" "$OUT/probe_workgroups_sweep.txt"
