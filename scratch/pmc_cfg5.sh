#!/bin/bash
# PMC rows of config 5's shard (k_ntt_lds encode + the two k_mm8w launches): FETCH_SIZE | WRITE_SIZE | SQ counters, separate passes
set -u
OUT="gpurun_out/pmc_cfg5"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; exec < /dev/null
W=cfg5-shard
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INST_CYCLES_VMEM"; do
  name=$(echo "$pass" | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$OUT/pmc_${W}_$name" -o p -- python bench.py --workload $W --steps 3 --warmup 1 --prewarm 0 --cpu-sample 0 --no-two-streams-extra > "$OUT/pmc_${W}_$name.log" 2>&1
done
timeout 900 python profiles/summarize_pmc.py "$OUT"/pmc_${W}_* > "$OUT/pmc_summary_$W.txt" 2>&1
rm -rf "$OUT"/pmc_${W}_FETCH_SIZE "$OUT"/pmc_${W}_WRITE_SIZE "$OUT"/pmc_${W}_SQ_WAVES "$OUT"/pmc_${W}_SQ_INSTS_LDS
grep -E "^kernel|k_ntt_lds|k_mm8w" "$OUT/pmc_summary_$W.txt" | cut -c1-330
