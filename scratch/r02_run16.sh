#!/bin/bash
O=gpurun_out/r02ai
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in cfg3-omega cfg5-shard; do
  rocprofv3 --kernel-trace --stats -d "$O/stats_$w" -o run -- python bench.py --workload $w --steps 20 --warmup 3 --cpu-sample 0 --no-two-streams-extra > "$O/stats_$w.log" 2>&1
  python profiles/summarize_rocpd.py "$O/stats_$w/run_results.db" > "$O/kernel_stats_$w.txt" 2>&1
done
