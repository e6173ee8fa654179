#!/bin/bash
mkdir -p gpurun_out/r02n
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/r02n/pytest.txt 2>&1
for w in cfg3-omega cfg5-shard cfg2; do
  python bench.py --workload $w --steps 20 --warmup 5 --cpu-sample 0 --no-two-streams-extra > gpurun_out/r02n/bench_$w.json 2> gpurun_out/r02n/bench_$w.err
done
python scratch/time_ntt.py > gpurun_out/r02n/time_ntt.txt 2>&1
