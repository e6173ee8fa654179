#!/bin/bash
mkdir -p gpurun_out/r02c
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > gpurun_out/r02c/pytest.txt 2>&1
for w in cfg3 cfg3-omega cfg5-shard cfg2; do
  python bench.py --workload $w --steps 20 --warmup 5 --cpu-sample 0 --no-two-streams-extra > gpurun_out/r02c/bench_$w.json 2> gpurun_out/r02c/bench_$w.err
done
python scratch/bench_robust.py 16384 > gpurun_out/r02c/robust.txt 2>&1
