#!/bin/bash
mkdir -p gpurun_out/r02u
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_offline.py -m gpu -q > gpurun_out/r02u/pytest.txt 2>&1
for w in cfg3-omega cfg5-shard cfg2; do
  python bench.py --workload $w --steps 20 --warmup 5 --cpu-sample 0 --no-two-streams-extra > gpurun_out/r02u/bench_$w.json 2> gpurun_out/r02u/bench_$w.err
done
