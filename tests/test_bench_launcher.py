"""bench.py as its own launcher: `python bench.py --gpus N` with no torchrun environment must start N ranks
under torch.distributed.run.  HB_BENCH_RENDEZVOUS_ONLY=1 stops every rank after the rendezvous (gloo), so this runs without a GPU;
the GPU twins (real opens on 2 ranks) are in tests/test_gpu_multirank.py."""
import json
import os
import subprocess
import sys

from conftest import REPO

_TORCHRUN_VARS = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "GROUP_RANK", "LOCAL_WORLD_SIZE")


def _plain_env():
    env = {k: v for k, v in os.environ.items() if k not in _TORCHRUN_VARS}
    env["HB_BENCH_RENDEZVOUS_ONLY"] = "1"
    return env


def _json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def test_plain_python_invocation_launches_its_ranks():
    res = subprocess.run([sys.executable, "bench.py", "--gpus", "3", "--workload", "tiny"], cwd=REPO, env=_plain_env(),
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-3000:]
    out = _json_line(res.stdout)
    assert out == {"rendezvous_only": True, "world_size_env": 3, "ranks_seen": 3, "gpus_requested": 3, "self_launched": True}
    assert "launching 3 ranks" in res.stderr


def test_single_gpu_invocation_stays_one_process():
    res = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--workload", "tiny"], cwd=REPO, env=_plain_env(),
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-3000:]
    out = _json_line(res.stdout)
    assert out["ranks_seen"] == 1 and not out["self_launched"]


def test_under_torchrun_no_second_launch():
    """the driver's own launch line: the script must NOT re-launch when torchrun already set RANK / WORLD_SIZE"""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", "2", "--workload", "tiny"]
    res = subprocess.run(cmd, cwd=REPO, env=_plain_env(), capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-3000:]
    out = _json_line(res.stdout)
    assert out["ranks_seen"] == 2 and out["world_size_env"] == 2 and not out["self_launched"]


def test_timeout_names_the_rank_that_fell_behind():
    """bench.Progress: every rank notes its phase; the watchdog of a rank that is still waiting after the limit prints which rank never got
    as far and ends the process (an 8-GPU run that wedges must say who)"""
    code = (
        "import os, sys, time\n"
        "os.environ['MASTER_PORT'] = 'test%d' % os.getpid()\n"
        "import bench\n"
        "late = bench.Progress(1, 3, 0)\n"           # rank 1 started and never got past 'start'
        "time.sleep(0.3)\n"
        "p2 = bench.Progress(2, 3, 0); p2.note('timed steps')\n"
        "p0 = bench.Progress(0, 3, 1.0); p0.note('timed steps')\n"    # rank 0's watchdog: one second
        "time.sleep(30)\n"
    )
    res = subprocess.run([sys.executable, "-c", code], cwd=REPO, capture_output=True, text=True, timeout=120)
    assert res.returncode == 3, (res.returncode, res.stderr[-2000:])
    assert "straggler = rank 1" in res.stderr and "last phase 'start'" in res.stderr
    assert "rank 2: 'timed steps'" in res.stderr
