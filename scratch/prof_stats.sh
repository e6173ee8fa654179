#!/bin/bash
# kernel-trace stats of one command: scratch/prof_stats.sh <tag> <command...>   (through gpurun; output gpurun_out/<tag>_kernel_stats.txt)
TAG="$1"; shift
OUT="gpurun_out/$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o run -- "$@" > "$OUT/run.log" 2>&1 < /dev/null
timeout 120 python profiles/summarize_rocpd.py "$OUT/stats/run_results.db" > "gpurun_out/${TAG}_kernel_stats.txt" 2>&1 < /dev/null
tail -4 "$OUT/run.log" | cut -c1-400
head -30 "gpurun_out/${TAG}_kernel_stats.txt" | cut -c1-200
