#!/bin/bash
O=gpurun_out/r02bl
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1000 python scratch/stress_wide.py 900 211 > $O/stress_wide.txt 2>&1
timeout 700 python scratch/stress_open_paths.py 600 223 > $O/stress_open.txt 2>&1
HB_CACHE_CAP=16 HB_PLAN_CACHE=2 timeout 400 python scratch/stress_wide.py 300 227 > $O/stress_caps.txt 2>&1
