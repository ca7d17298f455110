"""bench.py's own launcher (`python bench.py --gpus N` without torch.distributed.run): the parent starts N fresh
processes before it touches any GPU, relays rank 0's JSON line and fails when a child fails.  Run here on CPU with
`--dry-run` (gloo, one all-reduce of the parameter-gradient buffer); the timed GPU path uses the same launcher."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                          timeout=timeout, env=env, cwd=ROOT)


def test_gpus_flag_spawns_that_many_ranks():
    r = _run(["--gpus", "2", "--dry-run", "--config", "5"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["allreduce_ok"] is True
    assert out["allreduce_bytes"] == (6 * 7829 + 4 + 3) * 4


def test_single_rank_needs_no_launcher():
    r = _run(["--gpus", "1", "--dry-run", "--config", "5"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["rccl_ranks"] == 1


def test_a_failing_rank_fails_the_launch():
    # without --dry-run there is no GPU here: every rank exits non-zero, and so must the parent
    import torch
    if torch.cuda.device_count() > 0:
        import pytest
        pytest.skip("needs a box without GPU")
    r = _run(["--gpus", "2", "--config", "5", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "needs a GPU" in r.stderr or "exited with code" in r.stderr


def test_the_drivers_own_launch_line():
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2`:
    the ranks come with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment, bench.py must not launch anything itself."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--dry-run", "--config", "5"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["allreduce_ok"] is True
