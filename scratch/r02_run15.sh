#!/bin/bash
O=gpurun_out/r02av
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 400 python scratch/stress_wide.py 300 41 > $O/stress_wide.txt 2>&1
timeout 300 python scratch/stress_open_paths.py 200 17 > $O/stress_open.txt 2>&1
HB_CACHE_CAP=16 timeout 200 python scratch/stress_wide.py 90 43 > $O/stress_cap16.txt 2>&1
