#!/bin/bash
# kernel-trace summary of one bench workload:  scratch/kstats.sh <workload> [bench args]   (run through gpurun)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; exec < /dev/null
W="$1"; shift
rm -rf gpurun_out/ks; mkdir -p gpurun_out/ks
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/ks -o run -- python bench.py --workload $W --steps 30 --warmup 5 --cpu-sample 0 --no-two-streams-extra "$@" > gpurun_out/ks/log.txt 2>&1
python profiles/summarize_rocpd.py gpurun_out/ks/run_results.db | head -16 | cut -c1-170
rm -rf gpurun_out/ks
