#!/bin/bash
mkdir -p gpurun_out/r02s
cd $GRAFT_REPO_ROOT
python scratch/boundary_rates.py > gpurun_out/r02s/boundary.txt 2>&1
python scratch/test_mm8w.py 2>&1 | tail -3 > gpurun_out/r02s/mm8w_base.txt
HBMPC_HIP_LIB=honeybadgermpc_amd/lib/libhbmpc_hip_ilp.so python scratch/test_mm8w.py 2>&1 | tail -3 > gpurun_out/r02s/mm8w_ilp.txt
python bench.py --steps 50 --warmup 10 --cpu-sample 0 --no-two-streams-extra > gpurun_out/r02s/bench_base.json 2>/dev/null
HBMPC_HIP_LIB=honeybadgermpc_amd/lib/libhbmpc_hip_ilp.so python bench.py --steps 50 --warmup 10 --cpu-sample 0 --no-two-streams-extra > gpurun_out/r02s/bench_ilp.json 2>/dev/null
HBMPC_HIP_LIB=honeybadgermpc_amd/lib/libhbmpc_hip_ilp.so python scratch/time_ntt.py > gpurun_out/r02s/ntt_ilp.txt 2>&1
