set -u
OUT=gpurun_out/r04g; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
exec < /dev/null
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/stats_cfg4" -o run -- python bench.py --workload cfg4 --steps 5 --cpu-sample 0 > "$OUT/stats_cfg4.log" 2>&1
python profiles/summarize_rocpd.py "$OUT/stats_cfg4/run_results.db" > "$OUT/kernel_stats_cfg4.txt" 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU --kernel-trace --output-format csv -d "$OUT/c4_pmc_SQ" -o p -- python bench.py --workload cfg4 --steps 2 --warmup 1 --cpu-sample 0 > "$OUT/c4_pmc_SQ.log" 2>&1
timeout 900 python profiles/summarize_pmc.py "$OUT"/c4_pmc_SQ > "$OUT/pmc_sq_cfg4.txt" 2>&1
rm -rf "$OUT"/stats_cfg4 "$OUT"/c4_pmc_SQ
tail -3 "$OUT/stats_cfg4.log"; head -20 "$OUT/kernel_stats_cfg4.txt"; cat "$OUT/pmc_sq_cfg4.txt" | head -30
