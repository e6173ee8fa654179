#!/bin/bash
mkdir -p gpurun_out/r02h
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_offline.py -m gpu -q -x > gpurun_out/r02h/pytest.txt 2>&1
python scratch/bench_device_decoder.py > gpurun_out/r02h/decoder.txt 2>&1
