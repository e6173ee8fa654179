#!/bin/bash
O=gpurun_out/r02bg
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1
timeout 300 python scratch/stress_wide.py 200 57 > $O/stress_open.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
