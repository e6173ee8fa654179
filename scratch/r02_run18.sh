#!/bin/bash
O=gpurun_out/r02final11
mkdir -p $O
cd $GRAFT_REPO_ROOT
python bench.py > $O/bench_default.json 2> $O/bench_default.err
for w in cfg5-shard cfg3-omega cfg2 cfg5; do
  python bench.py --workload $w --steps 30 --warmup 5 --cpu-sample 0 --no-two-streams-extra > "$O/bench_$w.json" 2> "$O/bench_$w.err"
done
python bench.py --no-matrix-cores --steps 30 --warmup 5 --cpu-sample 0 --no-two-streams-extra > $O/bench_valu.json 2> $O/bench_valu.err
