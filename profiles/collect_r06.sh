#!/bin/bash
# Round-6 artefacts of profiles/ (run through gpurun; outputs under gpurun_out/<tag>/):   profiles/collect_r06.sh r06a
# 1. default bench line  2. kernel-trace stats of the same command  3. PMC passes (separate runs, counters only, --prewarm 0: equal launch counts) for
# the headline workload  4. the other workloads' lines (cfg5-shard, cfg3-omega, cfg2, cfg5, cfg4, cfg4 with 10 uniform erasures, cfg3-p64)
# 5. the protocol path at first sight: the three host flows, the GPU timelines, the host's own timeline  6. the decoder under attack  7. bounded stress runs
set -u
TAG="${1:-r06a}"
OUT="gpurun_out/$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
exec < /dev/null
pmc() {   # pmc <workload> <extra bench args...>: FETCH_SIZE | WRITE_SIZE | SQ passes -> pmc_summary_<w>.txt, traffic_<w>.json
  local W="$1"; shift
  for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
    local name=$(echo "$pass" | cut -d' ' -f1)
    timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$OUT/pmc_${W}_$name" -o p -- python bench.py --workload $W --steps 3 --warmup 1 --prewarm 0 --cpu-sample 0 --no-two-streams-extra "$@" > "$OUT/pmc_${W}_$name.log" 2>&1
  done
  timeout 900 python profiles/summarize_pmc.py "$OUT"/pmc_${W}_* > "$OUT/pmc_summary_$W.txt" 2>&1
  timeout 900 python profiles/make_traffic.py "$OUT/pmc_summary_$W.txt" $W "profiles/r06_pmc_$W.txt (rocprofv3 --pmc, separate passes: FETCH_SIZE | WRITE_SIZE | SQ counters, bench.py --prewarm 0; FETCH x2 gfx950 correction)" > "$OUT/traffic_$W.json" 2> "$OUT/traffic_$W.err"
  rm -rf "$OUT"/pmc_${W}_FETCH_SIZE "$OUT"/pmc_${W}_WRITE_SIZE "$OUT"/pmc_${W}_SQ_WAVES
}
stats() {  # stats <name> <command...>
  local N="$1"; shift
  timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/stats_$N" -o run -- "$@" > "$OUT/stats_$N.log" 2>&1
  timeout 600 python profiles/summarize_rocpd.py "$OUT/stats_$N/run_results.db" > "$OUT/kernel_stats_$N.txt" 2>&1
  rm -rf "$OUT/stats_$N"
}
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
stats cfg3 python bench.py --cpu-sample 0
pmc cfg3
for w in cfg5-shard cfg3-omega cfg2 cfg5; do
  timeout 600 python bench.py --workload $w --steps 30 --warmup 5 --cpu-sample 0 --no-two-streams-extra > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err"
done
timeout 900 python bench.py --workload cfg4 --cpu-sample 0 > "$OUT/bench_cfg4.json" 2> "$OUT/bench_cfg4.err"
timeout 900 python bench.py --workload cfg4 --erasures 10 --cpu-sample 0 > "$OUT/bench_cfg4_erasures10.json" 2> "$OUT/bench_cfg4_erasures10.err"
timeout 900 python bench.py --workload cfg3-p64 --cpu-sample 0 > "$OUT/bench_cfg3-p64.json" 2> "$OUT/bench_cfg3-p64.err"
stats cfg3-p64 python bench.py --workload cfg3-p64 --cpu-sample 0
pmc cfg3-p64
timeout 300 python scratch/time_open_p64.py 100 > "$OUT/open_p64.txt" 2>&1
HB_NO_MFMA=1 timeout 300 python scratch/time_open_p64.py 100 >> "$OUT/open_p64.txt" 2>&1
{
for m in wait "defer --r1-in-order" defer early; do timeout 300 python scratch/first_sight_timeline.py $m; done
for m in wait "defer --r1-in-order" defer early; do timeout 300 python scratch/first_sight_timeline.py $m; done
} > "$OUT/first_sight_flows.txt" 2>&1
for m in defer early; do
  timeout 600 rocprofv3 --kernel-trace -d "$OUT/prof_fs" -o fs -- python scratch/first_sight_timeline.py $m > /dev/null 2>&1
  { echo "# $m"; timeout 300 python scratch/first_sight_timeline.py --analyse "$OUT/prof_fs/fs_results.db"; } >> "$OUT/first_sight_timeline.txt" 2>&1
  rm -rf "$OUT/prof_fs"
done
timeout 300 python scratch/first_sight_host.py > "$OUT/first_sight_host.txt" 2>&1
timeout 300 python scratch/time_open3.py 100 > "$OUT/open3.txt" 2>&1
timeout 300 python scratch/dec21.py > "$OUT/dec21_cfg3.txt" 2>&1
timeout 300 python scratch/dec21.py 256 85 > "$OUT/dec21_cfg5.txt" 2>&1
timeout 300 python scratch/dec21.py 64 21 spread freeze > "$OUT/dec21_cfg3_spread.txt" 2>&1
timeout 300 python scratch/dec21.py 256 85 spread freeze > "$OUT/dec21_cfg5_spread.txt" 2>&1
timeout 300 python scratch/stress_decoder.py 100 61 > "$OUT/stress_decoder.txt" 2>&1
timeout 300 python scratch/stress_gao.py 80 62 > "$OUT/stress_gao.txt" 2>&1
timeout 300 python scratch/stress_open_paths.py 60 63 > "$OUT/stress_open_paths.txt" 2>&1
timeout 300 python scratch/stress_narrow.py 60 64 > "$OUT/stress_narrow.txt" 2>&1
tail -1 "$OUT/bench_default.json" | cut -c1-300
