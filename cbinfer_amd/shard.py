"""Sequence sharding across the GPUs of one node: one process per GPU, one independent video sequence
(own module state, own replica of the read-only weights) per process, no exchange on the data path.
The only collective is the final reduction of the timing (RCCL `nccl` backend on GPUs; `gloo` on CPU
for tests).  The reference has no multi-GPU code."""
import glob
import os
import subprocess
import sys

import torch


def visible_gpu_count():
    """How many GPUs a process started from here would see -- found WITHOUT a single HIP call in this process,
    so that the caller may still start child processes that use the GPU (a process that has initialised the GPU
    must not exec or be relied on to fork workers): the GPU nodes of the KFD topology in sysfs (simd_count > 0),
    cut down by HIP_/CUDA_/ROCR_VISIBLE_DEVICES when set.  Where sysfs says nothing, a child python process is
    asked for torch.cuda.device_count()."""
    for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        if os.environ.get(var, None) is not None and os.environ[var].strip() == '':
            return 0                       # explicitly hidden
    physical = None
    props = glob.glob('/sys/class/kfd/kfd/topology/nodes/*/properties')
    if props:
        n = 0
        try:
            for f in props:
                for line in open(f):
                    if line.startswith('simd_count') and int(line.split()[1]) > 0:
                        n += 1
            physical = n
        except (OSError, ValueError):
            physical = None
    if physical is None:
        try:
            out = subprocess.run([sys.executable, '-c', 'import torch; print(torch.cuda.device_count())'],
                                 stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300)
            return int(out.stdout.decode().strip().splitlines()[-1])    # (the child applied the env filters)
        except Exception:
            return 0
    count = physical
    for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        if var in os.environ:
            ids = [v for v in os.environ[var].split(',') if v.strip() != '']
            count = min(count, len(ids))
    return count


class SequenceShard(object):
    def __init__(self, backend=None):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        # CBINFER_FORCE_DIST=1: the process group (and every collective of the N > 1 control flow) also for ONE rank --
        # the only way to run the RCCL branch on a one-GPU box (tests/test_gpu_shard.py)
        forced = self.world == 1 and os.environ.get("CBINFER_FORCE_DIST", "0") == "1"
        if forced:
            os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if self.world > 1 or forced:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if backend is None:
                # RCCL ("nccl") on GPUs; CBINFER_DIST_BACKEND=gloo lets the N>1 control flow be rehearsed
                # with several ranks on one GPU, where RCCL refuses duplicate devices
                backend = os.environ.get("CBINFER_DIST_BACKEND") or (
                    "nccl" if torch.cuda.is_available() else "gloo")
            if not dist.is_initialized():
                dist.init_process_group(backend)
            self.dist = dist
        self.backend = backend

    def device(self):
        if torch.cuda.is_available():
            d = self.local_rank % max(1, torch.cuda.device_count())
            torch.cuda.set_device(d)
            return torch.device("cuda", d)
        return torch.device("cpu")

    def sequence_seed(self, base_seed):
        """Every rank processes a different sequence."""
        return int(base_seed) + self.rank

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def aggregate(self, steps, elapsed, device="cpu"):
        """(total frames over all ranks, MAX elapsed over ranks): whole-job throughput is their ratio."""
        if self.dist is None:
            return steps, elapsed
        if self.backend == "gloo":
            device = "cpu"
        t = torch.tensor([float(elapsed)], dtype=torch.float64, device=device)
        n = torch.tensor([float(steps)], dtype=torch.float64, device=device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        self.dist.all_reduce(n, op=self.dist.ReduceOp.SUM)
        return int(round(n.item())), float(t.item())

    def agree_max(self, value):
        """MAX of an integer over the ranks (CPU tensor under gloo, device tensor under RCCL)."""
        if self.dist is None:
            return int(value)
        dev = "cpu" if self.backend == "gloo" else "cuda"
        t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return int(t.item())

    def broadcast_flag(self, flag, src=0):
        """Rank src's boolean for everybody."""
        if self.dist is None:
            return bool(flag)
        dev = "cpu" if self.backend == "gloo" else "cuda"
        t = torch.tensor([1.0 if flag else 0.0], device=dev)
        self.dist.broadcast(t, src)
        return t.item() > 0.5

    def finish(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None
