"""The C++ host driver (examples/epsm_host_driver.cpp) drives tangent -> calc_grad -> scatter and the fused
launch through the C ABI alone -- no Python, no torch in the process -- and checks both routes against each
other on the device; this test runs the binary that __graft_entry__.build() produced."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "build", "epsm_host_driver")


def _exe():
    if not os.path.isfile(EXE):
        subprocess.run(["make", "-C", os.path.join(ROOT, "examples"), "-s"], check=True)
    return EXE


@pytest.mark.parametrize("n,K,variant", [(200000, 5, 0), (65536, 3, 1), (1000, 1, 0)])
def test_cxx_host_driver(n, K, variant):
    r = subprocess.run([_exe(), str(n), str(K), str(variant), "20000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().endswith("OK")


def test_cxx_host_driver_rccl_mode_one_rank():
    """`--ranks R`: one process per GPU started by a parent that never touches the GPU, ONE ncclAllReduce(float, sum) of
    the flat parameter-gradient buffer, self-check against the single-GPU sum.  A one-GPU box can only run R = 1 (RCCL
    wants one device per rank): the launcher, the id exchange, the communicator, the collective and the check are the
    code the R = 8 run uses."""
    r = subprocess.run([_exe(), "--ranks", "1", "200000", "5", "0", "20000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "RCCL communicator of 1" in r.stdout and r.stdout.strip().endswith("OK")


def test_cxx_host_driver_rccl_mode_refuses_more_ranks_than_gpus():
    import torch
    n = torch.cuda.device_count()
    r = subprocess.run([_exe(), "--ranks", str(n + 1), "1000", "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "need" in r.stderr


def test_bench_sharded_over_two_ranks_gives_the_same_gradient_image():
    """bench.py --gpus 2 as the driver launches it, except that both ranks share GPU 0 and talk gloo through the host
    (--rehearse-one-gpu; RCCL wants one device per rank): launcher, slab s -> rank s % 2, one all-reduce per step.  The
    gradient image is the one a single rank computes (sum of |parameter gradients| to 1e-5: atomic order only)."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--config", "4", "--res", "512", "--spp", "128", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"]
    lines = []
    for extra in (["--gpus", "1"], ["--gpus", "2", "--rehearse-one-gpu"]):
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common + extra, capture_output=True, text=True,
                           timeout=600, cwd=root, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        lines.append(json.loads(r.stdout.strip().splitlines()[-1]))
    one, two = lines
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["rccl_ranks"] == 2 and "rehearsal" in two
    assert one["config"]["slabs"] == 2 and two["config"]["slabs_per_gpu"] == 1
    assert one["grad_abs_sum"] > 0
    assert abs(one["grad_abs_sum"] - two["grad_abs_sum"]) <= 1e-5 * one["grad_abs_sum"]
