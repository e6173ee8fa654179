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


def _run(extra, timeout=900, nproc=2):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HB_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", str(nproc), "--steps", "3", "--warmup", "1", "--cpu-sample", "0"] + extra
    res = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=timeout)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


def _run_plain(extra, timeout=900):
    """`python bench.py --gpus 2 ...` with NO torchrun on the command line and no RANK / WORLD_SIZE in the environment: the script
    must launch its own ranks (this is how the driver invokes --gpus 1, and the r2 script died on an assert)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    env["HB_BENCH_SHARE_GPU"] = "1"
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--cpu-sample", "0"] + extra
    res = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=timeout)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


def test_self_launch_weak():
    out = _run_plain(["--workload", "tiny", "--no-two-streams-extra"])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["distributed"]["world_size"] == 2 and "self-launch" in out["distributed"]["launcher"]


def test_self_launch_sharded_auto_gather():
    out = _run_plain(["--workload", "cfg5-mini"])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["value"] > 0
    dd = out["distributed"]
    assert dd["world_size"] == 2 and dd["gather"]["requested"] == "auto" and dd["gather_mode"] in ("direct", "collective")
    assert set(dd["gather"]["probe_ms"]) == {"direct", "collective"}
    assert out["detail"]["bit_exact_vs_secrets"]


def test_wrong_world_size_is_reported_not_fatal():
    """torchrun with 2 ranks but --gpus 4: the launcher's world is what runs"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HB_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--workload", "tiny",
           "--no-two-streams-extra"]
    res = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 2 and out["distributed"]["gpus_requested"] == 4


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
    assert out["distributed"]["backend"] == "gloo" and out["distributed"]["world_size"] == 2
    assert out["detail"]["bit_exact_vs_secrets"] and out["detail"]["matrix_core_path"]


# ---- eight ranks (sharing the one GPU of the test box): the driver's N = 8 run must not be the first time rank 7 exists ----------
@pytest.mark.parametrize("mode", ["direct", "collective"])
def test_eight_ranks_sharded_open_with_gather(mode):
    """cfg5-mini split over EIGHT ranks (762 chunks -> 96, 96, 95, ...: uneven like config 5's 48 771), both gathers; bench.py
    asserts every rank's slice and the gathered vector bit for bit before it prints"""
    out = _run(["--workload", "cfg5-mini", "--gather", mode], nproc=8, timeout=1500)
    assert out["n_gpus"] == 8 and out["scaling"] == "strong" and out["value"] > 0
    assert out["config"]["shares_total"] == 1 << 16 and out["config"]["shares_this_rank"] <= (1 << 13) + 86
    assert out["detail"]["gather_mode"] == mode and out["distributed"]["world_size"] == 8
    assert out["detail"]["bit_exact_vs_secrets"]


def test_eight_ranks_weak_scaling_path():
    out = _run(["--workload", "tiny", "--no-two-streams-extra"], nproc=8, timeout=1500)
    assert out["n_gpus"] == 8 and out["scaling"] == "weak" and out["value"] > 0 and out["detail"]["bit_exact_vs_secrets"]
    assert out["distributed"]["world_size"] == 8


# ---- one rank on the RCCL backend: what can be exercised of the multi-GPU path on a one-GPU box ---------------------
@pytest.mark.parametrize("mode", ["direct", "collective", "auto"])
def test_one_rank_on_the_nccl_backend_goes_through_the_collectives(mode):
    """torchrun with ONE rank and no HB_BENCH_SHARE_GPU: backend nccl (RCCL) -- init_process_group with a device id, the barrier, the MAX
    reduction, the gather probes and the timed all-gather (all_gather_into_tensor / batch_isend_irecv with no peers) all run on RCCL;
    per-rank times are in the line"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k != "HB_BENCH_SHARE_GPU"}
    env["MASTER_ADDR"] = "127.0.0.1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", "1", "--steps", "3", "--warmup", "1", "--cpu-sample", "0", "--workload", "cfg5-mini", "--gather", mode]
    res = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    dd = out["distributed"]
    assert dd["backend"] == "nccl" and dd["world_size"] == 1 and out["n_gpus"] == 1
    assert dd["gather_mode"] in ("direct", "collective") and (mode == "auto" or dd["gather_mode"] == mode)
    assert set(k for k, v in dd["gather"]["probe_ms"].items() if v is not None) >= {dd["gather_mode"]}
    assert len(dd["per_rank"]) == 1 and dd["per_rank"][0]["compute_ms_per_step"] > 0
    assert out["detail"]["bit_exact_vs_secrets"]


def test_the_64_bit_prime_workload_line():
    """bench.py --workload cfg3-p64-mini: the open over p = 2^64 - 59 on the 1-limb kernels, checked bit for bit against the secrets inside
    bench.py, with roofline (three segments) and the CPU baseline on the same modulus"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "HB_BENCH_SHARE_GPU")}
    res = subprocess.run([sys.executable, "bench.py", "--workload", "cfg3-p64-mini", "--steps", "3", "--warmup", "1", "--cpu-sample", "8192"], cwd=REPO, env=env,
                         capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    assert out["value"] > 0 and out["dtype"].startswith("u64") and out["detail"]["bit_exact_vs_secrets"]
    assert len(out["roofline"]["segments"]) == 3 and 0 < out["roofline"]["frac"] < 1
    assert out["cpu_baseline"]["value"] > 0 and out["cpu_baseline"]["kind"] == "port"
