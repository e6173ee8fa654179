#!/bin/bash
# Round-4 artefacts of profiles/ (run through gpurun; outputs under gpurun_out/<tag>/):   profiles/collect_r04.sh r04p
# 1. default bench line  2. kernel-trace stats of the same command  3. PMC passes (separate runs, counters only) for the headline workload
# 4. the other workloads' lines  5. config 4 as a bench workload: line, kernel stats, HBM-traffic passes  6. device decoder with liars
# (config 3's shape and config 5's shard shape), first-sight timing of one decode, coalesced small opens
set -u
TAG="${1:-r04p}"
OUT="gpurun_out/$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
exec < /dev/null
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o run -- python bench.py --cpu-sample 0 > "$OUT/stats.log" 2>&1
timeout 900 python profiles/summarize_rocpd.py "$OUT/stats/run_results.db" > "$OUT/kernel_stats_cfg3.txt" 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  name=$(echo "$pass" | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$OUT/pmc_$name" -o p -- python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-two-streams-extra > "$OUT/pmc_$name.log" 2>&1
done
timeout 900 python profiles/summarize_pmc.py "$OUT"/pmc_* > "$OUT/pmc_summary_cfg3.txt" 2>&1
timeout 900 python profiles/make_traffic.py "$OUT/pmc_summary_cfg3.txt" cfg3 "profiles/r04_pmc_cfg3.txt (timeout 600 rocprofv3 --pmc, separate passes: FETCH_SIZE | WRITE_SIZE | SQ counters; FETCH x2 gfx950 correction)" > "$OUT/traffic_cfg3.json"
for w in cfg5-shard cfg3-omega cfg2 cfg5; do
  python bench.py --workload $w --steps 30 --warmup 5 --cpu-sample 0 --no-two-streams-extra > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err"
done
timeout 900 python bench.py --workload cfg4 > "$OUT/bench_cfg4.json" 2> "$OUT/bench_cfg4.err"
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/stats_cfg4" -o run -- python bench.py --workload cfg4 --steps 5 --cpu-sample 0 > "$OUT/stats_cfg4.log" 2>&1
python profiles/summarize_rocpd.py "$OUT/stats_cfg4/run_results.db" > "$OUT/kernel_stats_cfg4.txt" 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$OUT/c4_pmc_$pass" -o p -- python bench.py --workload cfg4 --steps 2 --warmup 1 --cpu-sample 0 > "$OUT/c4_pmc_$pass.log" 2>&1
done
timeout 900 python profiles/summarize_pmc.py "$OUT"/c4_pmc_* > "$OUT/pmc_summary_cfg4.txt" 2>&1
timeout 900 python profiles/make_traffic.py "$OUT/pmc_summary_cfg4.txt" cfg4 "profiles/r04_pmc_cfg4.txt (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE passes of bench.py --workload cfg4; FETCH x2 gfx950 correction)" > "$OUT/traffic_cfg4.json"
timeout 900 python scratch/bench_device_decoder.py > "$OUT/device_decoder.txt" 2>&1
timeout 900 python scratch/bench_device_decoder.py --copy > "$OUT/device_decoder_copied_columns.txt" 2>&1
timeout 600 python scratch/decoder_cfg5_shape.py > "$OUT/decoder_cfg5_shape.txt" 2>&1
timeout 600 python scratch/time_first_sight.py > "$OUT/first_sight_host_phases.txt" 2>&1
timeout 600 python scratch/first_sight_outliers.py 1500 all > "$OUT/first_sight_outliers.txt" 2>&1
timeout 600 python scratch/first_sight_outliers.py 1500 all nogc >> "$OUT/first_sight_outliers.txt" 2>&1
timeout 300 python scratch/dec21.py > "$OUT/dec21_cfg3.txt" 2>&1
timeout 300 python scratch/dec21.py 256 85 > "$OUT/dec21_cfg5.txt" 2>&1
timeout 300 python scratch/dec21.py 64 21 spread freeze > "$OUT/dec21_cfg3_spread.txt" 2>&1
timeout 300 python scratch/dec21.py 256 85 spread freeze > "$OUT/dec21_cfg5_spread.txt" 2>&1
timeout 300 python scratch/time_fetch.py > "$OUT/symbols_fetch.txt" 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/stats_dec21" -o run -- python scratch/dec21.py 256 85 > "$OUT/stats_dec21.log" 2>&1
python profiles/summarize_rocpd.py "$OUT/stats_dec21/run_results.db" > "$OUT/kernel_stats_dec21_cfg5.txt" 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU --kernel-trace --output-format csv -d "$OUT/c4_sq" -o p -- python bench.py --workload cfg4 --steps 2 --warmup 1 --cpu-sample 0 > "$OUT/c4_sq.log" 2>&1
timeout 900 python profiles/summarize_pmc.py "$OUT"/c4_sq > "$OUT/pmc_sq_cfg4.txt" 2>&1
timeout 400 python scratch/stress_decoder.py 240 31 > "$OUT/stress_decoder.txt" 2>&1
timeout 900 python scratch/bench_coalescer.py > "$OUT/coalescer.txt" 2>&1
timeout 900 python scratch/boundary_rates.py > "$OUT/boundary_rates.txt" 2>&1
# the raw traces (rocpd databases, counter CSVs) stay on the box: gpurun brings back 64 MiB at most, and the summaries above are what profiles/ keeps
rm -rf "$OUT"/stats "$OUT"/stats_cfg4 "$OUT"/stats_dec21 "$OUT"/pmc_FETCH_SIZE "$OUT"/pmc_WRITE_SIZE "$OUT"/pmc_SQ_WAVES "$OUT"/c4_pmc_* "$OUT"/c4_sq
tail -1 "$OUT/bench_default.json" | cut -c1-300
