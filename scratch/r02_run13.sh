#!/bin/bash
O=gpurun_out/r02ad
mkdir -p $O
cd $GRAFT_REPO_ROOT
for v in nored notail norednotail nomfma; do
  echo "== $v" >> $O/timing.txt
  HBMPC_HIP_LIB=honeybadgermpc_amd/lib/libhbmpc_hip_abl_$v.so python scratch/mm8w_phase_timing.py 2>&1 | grep -E "kernel|asm pass" >> $O/timing.txt
done
