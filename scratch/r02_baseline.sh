#!/bin/bash
# round-2 baseline: reference timings of every workload before this round's kernel work
mkdir -p gpurun_out/r02a
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for w in cfg3 cfg3-omega cfg5-shard cfg2; do
  python bench.py --workload $w --steps 20 --warmup 5 --cpu-sample 0 --no-two-streams-extra > gpurun_out/r02a/bench_$w.json 2> gpurun_out/r02a/bench_$w.err
done
python scratch/bench_robust.py 16384 > gpurun_out/r02a/robust.txt 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/r02a/prof_c5 -- python bench.py --workload cfg5-shard --steps 5 --warmup 2 --cpu-sample 0 --no-two-streams-extra > gpurun_out/r02a/prof_c5.log 2>&1
ls gpurun_out/r02a/prof_c5/*/ > gpurun_out/r02a/ls.txt 2>&1
