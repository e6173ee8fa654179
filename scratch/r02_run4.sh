#!/bin/bash
mkdir -p gpurun_out/r02f
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > gpurun_out/r02f/pytest.txt 2>&1
python bench.py --steps 50 --warmup 10 --cpu-sample 0 > gpurun_out/r02f/bench_cfg3.json 2> gpurun_out/r02f/bench_cfg3.err
python scratch/stress_open_paths.py 60 > gpurun_out/r02f/stress.txt 2>&1
