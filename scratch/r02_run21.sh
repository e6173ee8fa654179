#!/bin/bash
O=gpurun_out/r02ax
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1
timeout 200 python scratch/stress_open_paths.py 120 23 > $O/stress_open.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
