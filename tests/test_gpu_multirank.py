"""N > 1 on a GPU box: bench.py's real multi-rank paths under torch.distributed.run, two ranks sharing cuda:0
(HB_BENCH_SHARE_GPU=1: gloo carries the barrier / the gather, the kernels are the product's).  The 8-GPU RCCL run is the
driver's; this covers everything but the transport."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def _run(extra, timeout=900):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HB_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--cpu-sample", "0"] + extra
    res = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=timeout)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_weak_scaling_path():
    out = _run(["--workload", "tiny", "--no-two-streams-extra"])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0 and out["detail"]["bit_exact_vs_secrets"]


@pytest.mark.parametrize("mode", ["direct", "collective"])
def test_two_ranks_sharded_open_with_gather(mode):
    """one open split over two ranks by chunk, opened shares gathered on both: bench.py asserts bit-exactness of every
    rank's slice and of the gathered vector before it prints"""
    out = _run(["--workload", "cfg5-mini", "--gather", mode])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["value"] > 0
    assert out["config"]["shares_total"] == 1 << 16 and out["config"]["shares_this_rank"] <= (1 << 15) + 86
    assert out["detail"]["gather_mode"] == mode and out["detail"]["allgather_ms_per_step_max_over_ranks"] >= 0
    assert out["detail"]["bit_exact_vs_secrets"] and out["detail"]["matrix_core_path"]
