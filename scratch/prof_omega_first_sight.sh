# kernel stats of bench.py --workload cfg3-omega (the first-sight decoders' builder at d = 22 among them)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/om -o run -- python bench.py --workload cfg3-omega --steps 20 --cpu-sample 0 --no-two-streams-extra > /tmp/om.json 2>/tmp/om.err
python profiles/summarize_rocpd.py /tmp/om/run_results.db 2>&1 | grep "k_quick\|k_ntt_lds\|k_mm8w" | head -8
tail -1 /tmp/om.json | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value']/1e9, d['detail']['shares_per_s_per_gpu_first_sight_protocol_path']/1e9, d['detail']['r2_decode_under_attack_first_sight'])"
