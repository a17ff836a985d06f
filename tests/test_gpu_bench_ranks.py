"""bench.py with more than one rank (-m gpu): the exact launch line the driver uses for the scaling run
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N`), with two ranks on the one GPU of
the test box.  RCCL refuses two ranks on one device, so the collectives go over gloo here (BENCH_DIST_BACKEND / BENCH_FORCE_DEVICE
test hooks); everything else - rendezvous, barriers, max-over-ranks timing, the per-step all_gather of the latents, the extra
instrumented step every rank must enter, rank 0 printing one JSON line, clean exit - is the code path of an N-GPU node."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("scaling,batch", [("weak", 2), ("strong", 3)])
def test_bench_two_ranks(scaling, batch):
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo", BENCH_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--inference-steps", "2", "--batch", str(batch), "--scaling", scaling]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout + r.stderr)[-3000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == scaling and out["value"] > 0
    per_rank = out["config"]["images_per_rank"]
    assert per_rank == ([batch, batch] if scaling == "weak" else [2, 1])       # strong: the reference's batched_seeds split
    assert out["config"]["images_per_step"] == sum(per_rank)
    assert "kernel_classes" in out and out["cpu_baseline"] is None
    # the launch checks itself: world size as the collective backend saw it, one entry per rank with its device
    assert out["config"]["rccl_ranks_seen"] == 2 and out["config"]["dist_backend"] == "gloo"
    assert [d[0] for d in out["config"]["rank_devices"]] == [0, 1]


def test_bench_two_ranks_sdxl_strong_scaling():
    """BASELINE configs[3] in the form the 8-GPU node will run it (`--config sdxl --scaling strong --batch 16 --gpus 8`), shrunk to two
    ranks / three images / 256 px / two steps: one request split with the batched_seeds rule, added conditioning sharded with it."""
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo", BENCH_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--inference-steps", "2", "--batch", "3", "--scaling", "strong", "--config", "sdxl", "--size", "256", "--no-class-table"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout + r.stderr)[-3000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["value"] > 0
    assert out["config"]["images_per_rank"] == [2, 1] and out["config"]["rccl_ranks_seen"] == 2
    assert "SDXL" in out["config"]["workload"]


def test_plain_bench_line_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher (what the driver's 1-GPU line looks like with N = 2) must run TWO ranks."""
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo", BENCH_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--inference-steps", "2",
           "--batch", "2", "--no-class-table"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout + r.stderr)[-3000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["images_per_rank"] == [2, 2]
    assert "latency_request_split_s" in out


def test_plain_bench_line_refuses_more_ranks_than_gpus():
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BENCH_FORCE_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert f"--gpus {n}" in r.stderr and "GPU(s)" in r.stderr
