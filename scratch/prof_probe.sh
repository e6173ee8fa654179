set -u
OUT=gpurun_out/r04h; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
exec < /dev/null
python scratch/dec21.py 256 85 > $OUT/dec21_cfg5.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/st" -o run -- python scratch/dec21.py 256 85 > "$OUT/st.log" 2>&1
python profiles/summarize_rocpd.py "$OUT/st/run_results.db" > "$OUT/kernel_stats_dec21_cfg5.txt" 2>&1
rm -rf $OUT/st
cat $OUT/dec21_cfg5.txt; head -12 $OUT/kernel_stats_dec21_cfg5.txt
