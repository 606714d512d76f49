"""-m gpu rehearsal of bench.py's N > 1 control flow on the ONE GPU of the test box (verdict, round 2): two
ranks, started by bench.py's own self-launch as a child `torch.distributed.run`, share device 0 over the gloo
backend (RCCL refuses duplicate devices; CBINFER_ALLOW_SHARED_DEVICE=1 lifts bench.py's refusal for exactly this
purpose).  What executes end to end: the GPU count without a HIP call, the self-launch, SequenceShard,
the agreement on the repeat count (agree_max), the launch-form broadcast, the barriers inside timed_loop,
aggregate() and the single JSON line of rank 0.  It is NOT a scaling measurement."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_visible_gpu_count_makes_no_hip_call():
    # in a fresh interpreter: the count is there and torch has not initialised the GPU afterwards
    code = ("import torch; from cbinfer_amd.shard import visible_gpu_count; n = visible_gpu_count(); "
            "print(n, int(torch.cuda.is_initialized()))")
    out = subprocess.run([sys.executable, "-c", code], cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    n, inited = out.stdout.decode().split()[-2:]
    assert int(n) >= 1 and int(inited) == 0


def test_bench_two_ranks_share_the_gpu_over_gloo():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(CBINFER_DIST_BACKEND="gloo", CBINFER_ALLOW_SHARED_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
           "--min-seconds", "0.05", "--mode", "auto"]
    out = subprocess.run(cmd, env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    # rank 0 alone prints: the details object, then -- the FINAL line, what the driver parses -- the compact record
    assert len(lines) == 2 and lines[0].startswith('{"bench_details"') and len(lines[1]) < 4096, lines
    r = json.loads(lines[-1])
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["value"] > 0
    assert r["steps"] == 20                             # `value` is the literal K-step region (round 6)
    assert r["config"]["launch"] in ("graph", "graph4", "eager", "program")      # (rank 0's calibration, for every rank)
    # whole-job value = frames of BOTH ranks over the max elapsed
    assert abs(r["value"] - 2 * r["steps"] / r["timed_region_s"]) <= 1e-6 * r["value"]


def test_bench_still_refuses_by_default():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK",
                                                            "CBINFER_ALLOW_SHARED_DEVICE")}
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "7", "--steps", "1"],
                         env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    # (a 1-GPU box; on an 8-GPU node 7 ranks would simply run -- then there is nothing to refuse)
    from cbinfer_amd.shard import visible_gpu_count
    if visible_gpu_count() < 7:
        assert out.returncode != 0 and b"--gpus 7 requested but only" in out.stderr


def test_bench_control_flow_over_rccl_with_one_rank():
    """The RCCL ("nccl") branch of SequenceShard -- device tensors in agree_max / broadcast_flag / aggregate, barriers
    on the device -- had never run: the builder has one GPU and RCCL refuses two ranks on one device.
    CBINFER_FORCE_DIST=1 makes bench.py open the process group for its single rank as well, so every collective of the
    N > 1 control flow executes over RCCL once (world size 1: no peer, the same calls)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT",
                                                            "CBINFER_DIST_BACKEND")}
    env.update(CBINFER_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
           "--min-seconds", "0.05", "--mode", "auto", "--multi", "0", "--no-variants", "--no-secondary",
           "--no-isolated", "--no-cpu-baseline", "--no-dense", "--no-pipelined", "--no-last-frame"]
    out = subprocess.run(cmd, env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 2 and lines[0].startswith('{"bench_details"') and len(lines[1]) < 4096, lines
    r = json.loads(lines[-1])
    assert r["config"]["dist_backend"] == "nccl" and r["n_gpus"] == 1 and r["value"] > 0
    assert r["steps"] == 20 and r["config"]["launch"] in ("graph", "graph4", "eager", "program")
