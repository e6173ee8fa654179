#!/bin/bash
O=gpurun_out/r02aa
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1
python bench.py --workload cfg3-omega --steps 20 --warmup 5 --cpu-sample 0 > $O/bench_cfg3-omega.json 2> $O/bench_cfg3-omega.err
python scratch/bench_device_decoder.py > $O/decoder.txt 2>&1
