#!/bin/bash
# cfg5-shard, R2 launch of k_mm8w: HBM read requests with and without row-group units
OUT="gpurun_out/r03rg"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
exec < /dev/null
for mode in default all; do
  for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
    name=$(echo "$pass" | cut -d' ' -f1)
    if [ "$mode" = all ]; then export HB_MM8W_RQ=all; else unset HB_MM8W_RQ; fi
    timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$OUT/${mode}_$name" -o p -- python bench.py --workload cfg5-shard --steps 3 --warmup 1 --cpu-sample 0 --no-two-streams-extra > "$OUT/${mode}_$name.log" 2>&1
  done
  timeout 120 python profiles/summarize_pmc.py "$OUT"/${mode}_* > "$OUT/summary_$mode.txt" 2>&1
  unset HB_MM8W_RQ
  if [ "$mode" = all ]; then export HB_MM8W_RQ=all; fi
  timeout 300 python bench.py --workload cfg5-shard --steps 30 --warmup 5 --cpu-sample 0 --no-two-streams-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])" > "$OUT/bench_$mode.txt" 2>&1
  unset HB_MM8W_RQ
done
cat "$OUT"/bench_*.txt
grep -h "k_mm8w<true" "$OUT"/summary_default.txt "$OUT"/summary_all.txt | cut -c1-250
head -1 "$OUT"/summary_default.txt | cut -c1-250
