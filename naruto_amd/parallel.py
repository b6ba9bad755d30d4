"""Data-parallel mapping over the GPUs of one node: ray-batch sharding, one process per GPU.

The reference is single-process / single-GPU (reference scripts/naruto/run_replica.sh:14); the only
parallelism the path offers is over independent rays.  Every rank holds a full
replica of the parameters (hash table 6.5 MB, MLPs, uncertainty grid) and its own Adam state; one mapping
iteration exchanges exactly two things over RCCL (torch.distributed backend "nccl" on ROCm):

  1. the 16-slot fp64 vector of loss sums, BETWEEN forward and backward -- the mapping losses are
     normalised by GLOBAL counts (n_fs, n_sdf, n_valid, N: Co-SLAM get_masks, scene_rep.py:246-285), so
     each rank must scale its cotangents by the global denominators;
  2. the gradients, as ONE flat fp32 buffer (1 729 400 floats for office0), summed.

With (1) in place the sum over ranks of the per-rank gradients equals the single-process gradient of the
whole batch, so every rank then takes the identical Adam step (no parameter broadcast).

The functions here are backend-agnostic (they work on CPU tensors over gloo as well), which is what the
world_size-2 CPU tests exercise.
"""

from __future__ import annotations

from typing import Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

LOSS_SLOT_MINUNCERT = 9


def world_size(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def is_distributed(group=None) -> bool:
    """True when a process group exists (even with a single rank: the collective code path is then exercised)."""
    return group is not None and dist.is_available() and dist.is_initialized()


def rank(group=None) -> int:
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def shard_bounds(n_items: int, rank_: int, world: int) -> Tuple[int, int]:
    """Contiguous split of n_items over ranks; the first (n_items % world) ranks get one extra item."""
    base, extra = divmod(n_items, world)
    lo = rank_ * base + min(rank_, extra)
    return lo, lo + base + (1 if rank_ < extra else 0)


def shard_rays(tensors: Sequence[torch.Tensor], rank_: int, world: int) -> List[torch.Tensor]:
    lo, hi = shard_bounds(tensors[0].shape[0], rank_, world)
    return [t[lo:hi] for t in tensors]


class _Done:
    def wait(self):
        return True


def all_reduce_sum(t: torch.Tensor, group=None, async_op: bool = False):
    """SUM all-reduce of ``t`` in place.  RCCL ("nccl") takes device tensors directly; the gloo backend (CPU tests, and the
    two-ranks-on-one-GPU test) gets device tensors staged through the host.  ``async_op``: returns a handle whose ``wait()`` makes
    the current stream wait for the result (RCCL runs the collective on its own stream: kernels launched meanwhile overlap it)."""
    if not (dist.is_available() and dist.is_initialized()):
        return _Done()
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.detach().cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
        return _Done()
    w = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return w if async_op else _Done()


def reduce_scatter_sum(src: torch.Tensor, out: torch.Tensor, group=None) -> None:
    """out = this rank's 1/world slice of the SUM over ranks of ``src`` (src.numel() == world * out.numel()): the first half of a
    ring all-reduce -- (world-1)/world of src per GPU over the wire.  gloo gets device tensors staged through the host."""
    if not (dist.is_available() and dist.is_initialized()):
        out.copy_(src)
        return
    assert src.numel() == world_size(group) * out.numel()
    if src.is_cuda and dist.get_backend(group) == "gloo":
        h, o = src.detach().cpu(), torch.empty(out.numel(), dtype=out.dtype)
        dist.reduce_scatter_tensor(o, h, op=dist.ReduceOp.SUM, group=group)
        out.copy_(o)
        return
    dist.reduce_scatter_tensor(out, src, op=dist.ReduceOp.SUM, group=group)


def all_gather_into(dst: torch.Tensor, piece: torch.Tensor, group=None) -> None:
    """dst = concatenation over ranks of ``piece`` (dst.numel() == world * piece.numel()): the second half of a ring all-reduce."""
    if not (dist.is_available() and dist.is_initialized()):
        dst.copy_(piece)
        return
    assert dst.numel() == world_size(group) * piece.numel()
    if dst.is_cuda and dist.get_backend(group) == "gloo":
        h = torch.empty(dst.numel(), dtype=dst.dtype)
        dist.all_gather_into_tensor(h, piece.detach().cpu(), group=group)
        dst.copy_(h)
        return
    dist.all_gather_into_tensor(dst, piece, group=group)


def allreduce_loss_sums(sums: torch.Tensor, group=None) -> torch.Tensor:
    """In-place all-reduce (SUM) of the nine additive loss-sum slots.  Slot 9, min(uncert_map), stays this rank's own
    minimum: it only feeds the reference's ``assert uncert_map.min() > 0``, which every rank can check for its own
    rays -- not worth a second collective per iteration."""
    if not (dist.is_available() and dist.is_initialized()):
        return sums
    all_reduce_sum(sums[:LOSS_SLOT_MINUNCERT], group)
    return sums


def allreduce_grads(params: Iterable[torch.nn.Parameter], group=None, flat: Optional[torch.Tensor] = None) -> None:
    """Sum the .grad of all parameters over ranks through one flat bucket (one collective per step)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    ps = [p for p in params if p.grad is not None]
    if not ps:
        return
    # fast path: the gradients are views that tile one flat buffer (what the fused training node returns)
    bases = {id(p.grad._base) for p in ps if p.grad._base is not None}
    if len(bases) == 1 and all(p.grad._base is not None for p in ps):
        base = ps[0].grad._base
        if base.is_contiguous() and base.numel() == sum(p.grad.numel() for p in ps):
            all_reduce_sum(base, group)
            return
    n = sum(p.grad.numel() for p in ps)
    if flat is None or flat.numel() != n or flat.device != ps[0].grad.device:
        flat = torch.empty(n, dtype=ps[0].grad.dtype, device=ps[0].grad.device)
    off = 0
    for p in ps:
        k = p.grad.numel()
        flat[off:off + k].copy_(p.grad.reshape(-1))
        off += k
    all_reduce_sum(flat, group)
    off = 0
    for p in ps:
        k = p.grad.numel()
        p.grad.copy_(flat[off:off + k].view_as(p.grad))
        off += k


def init_from_env(backend: Optional[str] = None):
    """torchrun-style rendezvous (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT)."""
    import os
    # NARUTO_FORCE_DIST=1 builds the process group even for a single rank (exercises the collective path on one GPU)
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1 and os.environ.get("NARUTO_FORCE_DIST", "0") != "1":
        return None
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("LOCAL_RANK", "0")
    os.environ.setdefault("MASTER_PORT", "29511")
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(backend=backend)
    return dist.group.WORLD
