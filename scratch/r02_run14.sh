#!/bin/bash
O=gpurun_out/r02af
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1
python bench.py --steps 20 --warmup 5 --cpu-sample 0 > $O/bench_cfg3.json 2> $O/bench_cfg3.err
