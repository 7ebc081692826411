"""Data parallelism for the training step: one process per GPU, weights resident on every rank, each
rank reads its own shard of the batch, ONE all-reduce of the flat trainable-gradient buffer per step
(RCCL over xGMI through torch.distributed's "nccl" backend; "gloo" on CPU for tests).

Replaces the reference's `torch.nn.DataParallel` (sr.py:467-470), which every step scatters the input
from GPU 0, re-broadcasts all weights, gathers the logits and reduces the gradients onto GPU 0
(SURVEY 2a).  The frozen backbones (2 x 58 M parameters) are never communicated; BatchNorm statistics
stay per rank, exactly as under DataParallel (no SyncBN).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun).
    Returns (rank, world_size, local_rank).  World size 1 needs no process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("SR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class GradBucket:
    """Flat fp32 bucket over the trainable parameters.  `reduce()` copies the gradients in, runs one
    all-reduce(sum) and scatters `sum / world` back, so `clip_grad_norm_` afterwards sees the gradient of the
    mean loss over the GLOBAL batch (the reference clips after DataParallel's reduction, sr.py:79-81)."""

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, device=dev, dtype=torch.float32)
        self.views, o = [], 0
        for p in self.params:
            self.views.append(self.flat[o:o + p.numel()].view_as(p))
            o += p.numel()

    @property
    def nbytes(self):
        return self.flat.numel() * 4

    def reduce(self, weight=1.0):
        """weight: this rank's share of the global mean (default 1 -> plain average over ranks)."""
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        with torch.no_grad():
            for v, p in zip(self.views, self.params):
                if p.grad is None:
                    v.zero_()
                else:
                    v.copy_(p.grad)
            if world > 1:
                if weight != 1.0:
                    self.flat.mul_(weight)
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
                self.flat.div_(world)
                for v, p in zip(self.views, self.params):
                    if p.grad is None:
                        p.grad = v.clone()
                    else:
                        p.grad.copy_(v)


def shard_range(total, rank, world):
    """Contiguous shard [lo, hi) of `total` samples for `rank` (sizes differ by at most 1)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def barrier():
    if dist.is_initialized():
        dist.barrier()
