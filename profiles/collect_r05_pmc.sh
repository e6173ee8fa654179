#!/bin/bash
# Round-5 PMC passes of the headline workload (run through gpurun):   profiles/collect_r05_pmc.sh r05pmc [workload]
# Separate rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ counters), counters only + --kernel-trace, --prewarm 0 so that every pass sees
# the same launches; the summary keys rows on (kernel, grid, dynamic LDS): the R1 and R2 launches of k_mm8f are separate rows.
set -u
TAG="${1:-r05pmc}"
W="${2:-cfg3}"
OUT="gpurun_out/$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
exec < /dev/null
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  name=$(echo "$pass" | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$OUT/pmc_${W}_$name" -o p -- python bench.py --workload $W --steps 3 --warmup 1 --prewarm 0 --cpu-sample 0 --no-two-streams-extra > "$OUT/pmc_${W}_$name.log" 2>&1
done
timeout 900 python profiles/summarize_pmc.py "$OUT"/pmc_${W}_* > "$OUT/pmc_summary_$W.txt" 2>&1
timeout 900 python profiles/make_traffic.py "$OUT/pmc_summary_$W.txt" $W "profiles/r05_pmc_$W.txt (rocprofv3 --pmc, separate passes: FETCH_SIZE | WRITE_SIZE | SQ counters, bench.py --prewarm 0; FETCH x2 gfx950 correction; rows keyed on kernel, grid and LDS size)" > "$OUT/traffic_$W.json" 2> "$OUT/traffic_$W.err"
head -1 "$OUT"/pmc_${W}_FETCH_SIZE/*/*counter_collection.csv 2>/dev/null | head -3
rm -rf "$OUT"/pmc_${W}_FETCH_SIZE "$OUT"/pmc_${W}_WRITE_SIZE "$OUT"/pmc_${W}_SQ_WAVES
cat "$OUT/traffic_$W.json" "$OUT/traffic_$W.err" | head -30
