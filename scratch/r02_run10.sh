#!/bin/bash
O=gpurun_out/r02y
mkdir -p $O
cd $GRAFT_REPO_ROOT
for w in 1 2; do
  echo "WG_PER_CU=$w" >> $O/mm8w.txt
  HB_MM8W_WG_PER_CU=$w python scratch/test_mm8w.py 2>&1 | grep "us per launch" >> $O/mm8w.txt
done
