"""Sequence sharding across the GPUs of one node: one process per GPU, one independent video sequence
(own module state, own replica of the read-only weights) per process, no exchange on the data path.
The only collective is the final reduction of the timing (RCCL `nccl` backend on GPUs; `gloo` on CPU
for tests).  The reference has no multi-GPU code."""
import os

import torch


class SequenceShard(object):
    def __init__(self, backend=None):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        if self.world > 1:
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

    def finish(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None
