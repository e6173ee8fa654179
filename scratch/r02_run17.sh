#!/bin/bash
O=gpurun_out/r02ba
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
python bench.py --workload cfg3-omega --steps 20 --warmup 5 --cpu-sample 0 > $O/bench_cfg3-omega.json 2> $O/bench_cfg3-omega.err
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
