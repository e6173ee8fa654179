#!/bin/bash
O=gpurun_out/r02bd
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1
python scratch/bench_device_decoder.py > $O/decoder.txt 2>&1
python scratch/plan_create_cost.py > $O/plan.txt 2>&1
python bench.py --steps 20 --warmup 5 --cpu-sample 0 > $O/bench_default.json 2> $O/bench_default.err
python scratch/bench_coalescer.py > $O/coalescer.txt 2>&1
