#!/bin/bash
O=gpurun_out/r02bk
mkdir -p $O
cd $GRAFT_REPO_ROOT
HB_MM8W_TILE16=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -q > $O/tile16.txt 2>&1
HB_NO_FUSED_VALIDATE=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_offline.py -m gpu -q > $O/nofused.txt 2>&1
HB_NO_MFMA_WIDE=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -q > $O/nowide.txt 2>&1
HB_PLAN_CACHE=0 python -m pytest tests/test_gpu_offline.py -m gpu -q > $O/noplancache.txt 2>&1
