#!/bin/bash
# Round-3 artefacts of profiles/ (run through gpurun; outputs under gpurun_out/<tag>/):
#   profiles/collect_r03.sh r03p
# 1. default bench line  2. kernel-trace stats of the same command  3. PMC passes (separate runs, counters only) for the
# headline workload  4. stats + HBM-traffic passes for the omega-point shard of config 5 and for config 3 at omega points
# 5. config 4 (robust decoders) timing + HBM-traffic passes  6. device decoder with liars, coalesced small opens
set -u
TAG="${1:-r03p}"
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
timeout 900 python profiles/make_traffic.py "$OUT/pmc_summary_cfg3.txt" cfg3 "profiles/r03_pmc_cfg3.txt (timeout 600 rocprofv3 --pmc, separate passes: FETCH_SIZE | WRITE_SIZE | SQ counters; FETCH x2 gfx950 correction)" > "$OUT/traffic_cfg3.json"
for w in cfg5-shard cfg3-omega cfg2 cfg5; do
  python bench.py --workload $w --steps 30 --warmup 5 --cpu-sample 0 --no-two-streams-extra > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err"
done
for w in cfg5-shard cfg3-omega; do
  timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/stats_$w" -o run -- python bench.py --workload $w --steps 20 --warmup 3 --cpu-sample 0 --no-two-streams-extra > "$OUT/stats_$w.log" 2>&1
  python profiles/summarize_rocpd.py "$OUT/stats_$w/run_results.db" > "$OUT/kernel_stats_$w.txt" 2>&1
done
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  name=$(echo "$pass" | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$OUT/c5_pmc_$name" -o p -- python bench.py --workload cfg5-shard --steps 3 --warmup 1 --cpu-sample 0 --no-two-streams-extra > "$OUT/c5_pmc_$name.log" 2>&1
done
timeout 900 python profiles/summarize_pmc.py "$OUT"/c5_pmc_* > "$OUT/pmc_summary_cfg5-shard.txt" 2>&1
timeout 900 python profiles/make_traffic.py "$OUT/pmc_summary_cfg5-shard.txt" cfg5-shard "profiles/r03_pmc_cfg5-shard.txt (same passes and correction)" > "$OUT/traffic_cfg5-shard.json"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  name=$(echo "$pass" | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$OUT/c3o_pmc_$name" -o p -- python bench.py --workload cfg3-omega --steps 3 --warmup 1 --cpu-sample 0 --no-two-streams-extra > "$OUT/c3o_pmc_$name.log" 2>&1
done
timeout 900 python profiles/summarize_pmc.py "$OUT"/c3o_pmc_* > "$OUT/pmc_summary_cfg3-omega.txt" 2>&1
timeout 900 python profiles/make_traffic.py "$OUT/pmc_summary_cfg3-omega.txt" cfg3-omega "profiles/r03_pmc_cfg3-omega.txt (same passes and correction)" > "$OUT/traffic_cfg3-omega.json"
timeout 900 python scratch/bench_robust.py 262144 > "$OUT/robust_cfg4.txt" 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$OUT/c4_pmc_$pass" -o p -- python scratch/bench_robust.py 16384 > "$OUT/c4_pmc_$pass.log" 2>&1
done
timeout 900 python profiles/summarize_pmc.py "$OUT"/c4_pmc_* > "$OUT/pmc_summary_cfg4.txt" 2>&1
timeout 900 python scratch/bench_device_decoder.py > "$OUT/device_decoder.txt" 2>&1
timeout 900 python scratch/bench_coalescer.py > "$OUT/coalescer.txt" 2>&1
timeout 600 python scratch/decoder_cfg5_shape.py > "$OUT/decoder_cfg5_shape.txt" 2>&1
timeout 900 python scratch/boundary_rates.py > "$OUT/boundary_rates.txt" 2>&1
timeout 900 python scratch/plan_create_cost.py > "$OUT/plan.txt" 2>&1
if [ -f honeybadgermpc_amd/lib/libhbmpc_hip_timing.so ]; then HBMPC_HIP_LIB=honeybadgermpc_amd/lib/libhbmpc_hip_timing.so python scratch/mm8w_phase_timing.py > "$OUT/mm8w_phase_timing.txt" 2>&1; fi
# the raw traces (rocpd databases, counter CSVs) stay on the box: gpurun brings back 64 MiB at most, and the summaries above are what profiles/ keeps
rm -rf "$OUT"/stats "$OUT"/stats_* "$OUT"/pmc_FETCH_SIZE "$OUT"/pmc_WRITE_SIZE "$OUT"/pmc_SQ_WAVES "$OUT"/c5_pmc_* "$OUT"/c3o_pmc_* "$OUT"/c4_pmc_*
tail -1 "$OUT/bench_default.json" | cut -c1-300
