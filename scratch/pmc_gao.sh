#!/bin/bash
# vector instructions, waves and busy cycles of the Euclid kernel, two codewords a wave against one (run through gpurun): W=<workload> bash scratch/pmc_gao.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/gao
for mode in pair single; do
  if [ $mode = single ]; then export HB_GAO_PAIR=0; else unset HB_GAO_PAIR; fi
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/gao/pmc_$mode -o p -- python bench.py --workload ${W:-cfg4-n64} --steps 2 --warmup 1 --prewarm 0 --cpu-sample 0 > gpurun_out/gao/pmc_$mode.log 2>&1
  python - "$mode" <<'PY'
import csv, glob, sys, collections
mode = sys.argv[1]
f = glob.glob(f"gpurun_out/gao/pmc_{mode}/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
    if "gao" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
for k, v in acc.items():
    print(mode, k, "launches", cnt[k], {c: round(x / cnt[k]) for c, x in v.items()})
PY
  rm -rf gpurun_out/gao/pmc_$mode
done
