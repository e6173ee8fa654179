# repeats the 8-rank shared-GPU bench launch (tests/test_gpu_multirank.py) to look for a hang: each try under its own timeout,
# Python stacks of every rank dumped after 40 s (HB_BENCH_DUMP_AFTER)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04j
export HB_BENCH_SHARE_GPU=1 MASTER_ADDR=127.0.0.1 HB_BENCH_DUMP_AFTER=40
for i in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16; do
  if [ $((i % 2)) -eq 0 ]; then mode=direct; else mode=collective; fi
  port=$((20000 + RANDOM % 20000))
  s=$(date +%s)
  timeout 60 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 8 --steps 3 --warmup 1 --cpu-sample 0 --workload cfg5-mini --gather $mode > gpurun_out/r04j/run_${i}_$mode.out 2> gpurun_out/r04j/run_${i}_$mode.err
  rc=$?
  echo "try $i $mode rc=$rc $(( $(date +%s) - s )) s"
  if [ $rc -ne 0 ]; then break; fi
done
