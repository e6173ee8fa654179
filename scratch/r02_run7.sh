#!/bin/bash
mkdir -p gpurun_out/r02t
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > gpurun_out/r02t/pytest.txt 2>&1
python scratch/time_ntt.py > gpurun_out/r02t/time_ntt.txt 2>&1
python scratch/bench_robust.py 65536 > gpurun_out/r02t/robust.txt 2>&1
for w in cfg3-omega cfg5-shard; do
  python bench.py --workload $w --steps 20 --warmup 5 --cpu-sample 0 --no-two-streams-extra > gpurun_out/r02t/bench_$w.json 2> gpurun_out/r02t/bench_$w.err
done
