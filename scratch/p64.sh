#!/bin/bash
cd "$GRAFT_REPO_ROOT"; exec < /dev/null
python -m pytest tests -m gpu -x -q -k "narrow or moduli or p64" 2>&1 | tail -2
python scratch/time_open_p64.py 2>&1 | tail -1
HB_NO_MFMA=1 python scratch/time_open_p64.py 2>&1 | tail -1
