"""The N>1 path on CPU: world_size-2 `gloo` run of the sequence-shard helper bench.py uses (one
sequence per rank, different seeds, MAX(elapsed)/SUM(frames) reduction, no data-path collective)."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from cbinfer_amd.shard import SequenceShard
    from cbinfer_amd import workloads
    shard = SequenceShard(backend="gloo")
    assert shard.world == world and shard.rank == rank
    vid = workloads.SyntheticVideo(H=32, W=48, ratio=0.25, block=8, seed=shard.sequence_seed(100),
                                   device="cpu")
    checksum = float(vid.frames(3)[-1].sum())
    shard.barrier()
    elapsed = 0.5 + 0.25 * rank          # rank 1 is the slow one
    frames, t = shard.aggregate(10, elapsed)
    out.put((rank, frames, t, checksum))
    shard.finish()


def test_two_rank_gloo_aggregation():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [20, 20]                 # SUM of frames over ranks
    assert all(abs(r[2] - 0.75) < 1e-12 for r in res)      # MAX of elapsed over ranks
    assert res[0][3] != res[1][3]                          # different sequence per rank


def test_single_process_is_identity():
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    from cbinfer_amd.shard import SequenceShard
    s = SequenceShard()
    assert s.world == 1 and s.aggregate(7, 0.5) == (7, 0.5) and s.sequence_seed(3) == 3
    s.barrier()
    s.finish()


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus N` without a launcher starts its own ranks -- and must refuse, loudly and
    before touching a GPU, when fewer than N are visible (round 1 silently ran a single rank and reported
    n_gpus = 1).  This container has none."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = ""
    env["CUDA_VISIBLE_DEVICES"] = ""
    out = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "1"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert out.returncode != 0
    assert b"--gpus 2 requested but only 0 GPU(s) are visible" in out.stderr
    assert out.stdout.strip() == b""       # no JSON line pretending to be a result


def test_bench_rejects_world_size_mismatch():
    """Launched by a launcher with another world size than --gpus says: refuse instead of mislabelling."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "4", "--steps", "1"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert out.returncode != 0 and b"--gpus 4 but WORLD_SIZE=1" in out.stderr
