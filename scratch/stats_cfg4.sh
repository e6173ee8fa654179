#!/bin/bash
# kernel-trace stats of config 4's bench line, with and without two codewords a wave (run through gpurun)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/gao
for mode in pair single; do
  if [ $mode = single ]; then export HB_GAO_PAIR=0; else unset HB_GAO_PAIR; fi
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/gao/stats_$mode -o run -- python bench.py --workload ${W:-cfg4} --steps 5 --cpu-sample 0 > gpurun_out/gao/stats_$mode.log 2>&1
  timeout 300 python profiles/summarize_rocpd.py gpurun_out/gao/stats_$mode/run_results.db > gpurun_out/gao/kernel_stats_$mode.txt 2>&1
  rm -rf gpurun_out/gao/stats_$mode
  head -12 gpurun_out/gao/kernel_stats_$mode.txt
done
