#!/bin/bash
O=gpurun_out/r02bf
mkdir -p $O
cd $GRAFT_REPO_ROOT
python scratch/test_mm8w.py > $O/mm8w.txt 2>&1; python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_offline.py -m gpu -q -x > $O/pytest.txt 2>&1
python bench.py --steps 30 --warmup 5 --cpu-sample 0 --no-two-streams-extra > $O/bench_cfg3.json 2> $O/bench_cfg3.err
for w in cfg3-omega cfg5-shard cfg2; do
  python bench.py --workload $w --steps 20 --warmup 5 --cpu-sample 0 --no-two-streams-extra > $O/bench_$w.json 2> $O/bench_$w.err
done
timeout 200 python scratch/stress_wide.py 40 33 > $O/stress.txt 2>&1
