#!/bin/bash
# Re-runs the bench lines of collect_r04.sh (same tag: kernel stats and PMC passes of the first pass stay -- the kernels are the same):   profiles/collect_r04_part3.sh r04s
set -u
TAG="${1:-r04s}"
OUT="gpurun_out/$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
exec < /dev/null
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
for w in cfg5-shard cfg3-omega cfg2 cfg5; do
  python bench.py --workload $w --steps 30 --warmup 5 --cpu-sample 0 --no-two-streams-extra > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err"
done
tail -1 "$OUT/bench_default.json" | cut -c1-200
