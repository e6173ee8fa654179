#!/bin/bash
# Reproduces the artefacts of profiles/ on the GPU box (run through gpurun; outputs under gpurun_out/<tag>/).
#   profiles/collect.sh r01c
# 1. the default bench line, 2. kernel-trace stats of the same command, 3.-5. PMC passes (separate runs, counters only).
set -u
TAG="${1:-run}"
OUT="gpurun_out/$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o run -- python bench.py --cpu-sample 0 > "$OUT/stats.log" 2>&1
python profiles/summarize_rocpd.py "$OUT/stats/run_results.db" > "$OUT/kernel_stats.txt" 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  name=$(echo "$pass" | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$OUT/pmc_$name" -o p -- python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-two-streams-extra > "$OUT/pmc_$name.log" 2>&1
done
python profiles/summarize_pmc.py "$OUT"/pmc_* > "$OUT/pmc_summary.txt" 2>&1
tail -1 "$OUT/bench_default.json" | cut -c1-400
head -12 "$OUT/kernel_stats.txt"
cat "$OUT/pmc_summary.txt"
