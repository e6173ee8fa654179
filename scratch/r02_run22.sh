#!/bin/bash
O=gpurun_out/r02bc
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 700 python scratch/stress_wide.py 600 101 > $O/stress_wide.txt 2>&1
timeout 500 python scratch/stress_open_paths.py 400 103 > $O/stress_open.txt 2>&1
