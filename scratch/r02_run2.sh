#!/bin/bash
mkdir -p gpurun_out/r02d
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_multirank.py tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/r02d/pytest.txt 2>&1
python bench.py --workload cfg5 --steps 10 --warmup 3 --cpu-sample 0 > gpurun_out/r02d/bench_cfg5_n1.json 2> gpurun_out/r02d/bench_cfg5_n1.err
python bench.py --steps 20 --warmup 5 --cpu-sample 65536 --no-two-streams-extra > gpurun_out/r02d/bench_cfg3_cpu.json 2> gpurun_out/r02d/bench_cfg3_cpu.err
