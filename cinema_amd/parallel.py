"""Multi-GPU plumbing of the hot path: one process per GPU, subjects sharded, MLP replicated.

The reference is single-device.  Its data-parallel restatement: a subject -- latent grid, rigid/affine
row, resident volumes, foreground list, AdamW moments -- lives on exactly ONE rank; every rank runs the reference's
step (build_atlas.py:69-90) on `n_samples` coordinates of ITS subjects; the only data-path exchange is one
all-reduce(average) of the flat SIREN gradient per step.  Everything here is backend-agnostic (RCCL on the GPUs,
gloo in the CPU tests).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import os

import torch
import torch.distributed as dist


def world_info(process_group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(process_group), dist.get_world_size(process_group)
    return 0, 1


def shard_subjects(n_subjects: int, rank: int, world: int) -> List[int]:
    """Dataframe rows owned by `rank`: a balanced contiguous block (sizes differ by at most one), so that a
    checkpoint's latent / transformation tables are the rank-order concatenation of the shards."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, rem = divmod(n_subjects, world)
    lo = rank * base + min(rank, rem)
    return list(range(lo, lo + base + (1 if rank < rem else 0)))


def owner_of(subject: int, n_subjects: int, world: int) -> int:
    base, rem = divmod(n_subjects, world)
    edge = rem * (base + 1)
    return subject // (base + 1) if subject < edge else rem + (subject - edge) // max(base, 1)


def agree_steps(local_steps: int, process_group=None, device="cpu") -> int:
    """Steps per epoch every rank will run (the minimum over ranks: a rank whose foreground list is longer drops
    its tail for this epoch; the list is reshuffled every epoch so no voxel is excluded systematically).
    Without it a rank with more steps would wait forever in the gradient all-reduce."""
    _, world = world_info(process_group)
    if world == 1:
        return local_steps
    if dist.get_backend(process_group) != "nccl":
        device = "cpu"                      # gloo: host tensors only (see allreduce_mean_)
    t = torch.tensor([local_steps], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=process_group)
    return int(t.item())


def allreduce_mean_(flat_grad: torch.Tensor, process_group=None, async_op: bool = False):
    """Average the replicated MLP's flat gradient over ranks in place.  With the per-rank losses normalised by the union
    batch's counts (share_batch_normaliser_) this is the gradient of the same loss on the union batch
    (tests/test_dp_gloo.py).  RCCL has a native AVG; gloo has not, so sum and scale there."""
    _, world = world_info(process_group)
    if world == 1:
        return None
    if os.environ.get("CINEMA_DP_SKIP_ALLREDUCE") == "1":
        return None      # MEASUREMENT ONLY (bench.py allreduce_exposed_ms): the replicas drift apart from here on
    backend = dist.get_backend(process_group)
    if backend == "nccl":
        return dist.all_reduce(flat_grad, op=dist.ReduceOp.AVG, group=process_group, async_op=async_op)
    if flat_grad.is_cuda:
        # gloo is a CPU transport.  Handed a device tensor it stages through its own streams and host callbacks, and two
        # processes sharing one GPU have been seen to sit in that path for minutes (both ranks inside all_reduce).  Keep
        # the device out of it: finish this rank's queue, copy to a pinned host buffer, reduce on the host, copy back.
        torch.cuda.current_stream(flat_grad.device).synchronize()
        host = _host_stage(flat_grad)
        host.copy_(flat_grad)
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=process_group)
        host.div_(world)
        flat_grad.copy_(host)
        return None
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=process_group, async_op=False)
    flat_grad.div_(world)
    return None


class _UnionNormaliser:
    """Handle of share_batch_normaliser_: wait() leaves norm = union-batch sums / world on the current stream."""

    def __init__(self, norm, work, world):
        self.norm, self.work, self.world = norm, work, world

    def wait(self):
        if self.work is not None:
            self.work.wait()
        self.norm.mul_(1.0 / self.world)
        return self.norm


def share_batch_normaliser_(norm: torch.Tensor, process_group=None):
    """norm = fp32[2] {samples, sum of class weights} of THIS rank's batch; after .wait() it holds the union batch's sums
    divided by the number of ranks.  A rank that normalises its mean losses by these (instead of its own counts) turns
    the rank-AVERAGE of the gradients into the union batch's gradient also when class weights make the cross-entropy
    denominators differ between ranks or when a rank runs a shorter last chunk (utils.py:57-79 on the union batch).
    RCCL: started asynchronously (8 bytes; it hides under the forward pass); gloo: host tensors only, synchronous."""
    _, world = world_info(process_group)
    if world == 1:
        return _UnionNormaliser(norm, None, 1)
    if dist.get_backend(process_group) == "nccl":
        return _UnionNormaliser(norm, dist.all_reduce(norm, op=dist.ReduceOp.SUM, group=process_group, async_op=True), world)
    if norm.is_cuda:
        host = norm.detach().cpu()          # synchronises this rank's queue; gloo never sees the device
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=process_group)
        norm.copy_(host)
    else:
        dist.all_reduce(norm, op=dist.ReduceOp.SUM, group=process_group)
    return _UnionNormaliser(norm, None, world)


def allreduce_sum(t: torch.Tensor, process_group=None) -> torch.Tensor:
    """Sum a small tensor over ranks (the partial weighted latent sums of the atlas, 27.6 KB); returns the result on
    t's device.  gloo sees host tensors only (see allreduce_mean_)."""
    _, world = world_info(process_group)
    if world == 1:
        return t
    if t.is_cuda and dist.get_backend(process_group) != "nccl":
        host = t.detach().cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=process_group)
        return host.to(t.device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=process_group)
    return t


_HOST_STAGE: Dict[tuple, torch.Tensor] = {}


def _host_stage(t: torch.Tensor) -> torch.Tensor:
    """Reusable pinned host buffer of t's size and dtype (the gloo rehearsal path on a GPU box only)."""
    key = (t.numel(), t.dtype)
    buf = _HOST_STAGE.get(key)
    if buf is None:
        buf = torch.empty(t.numel(), dtype=t.dtype, pin_memory=True)
        _HOST_STAGE[key] = buf
    return buf.view(t.shape)      # the cache holds the flat buffer: tensors of equal size and different shape share it


def broadcast_(tensor: torch.Tensor, src: int = 0, process_group=None):
    """Make the replicated parameters identical at start-up (each rank seeds its own latents but must share the MLP)."""
    _, world = world_info(process_group)
    if world > 1:
        if tensor.is_cuda and dist.get_backend(process_group) != "nccl":
            host = tensor.detach().cpu()    # gloo: host tensors only (see allreduce_mean_)
            dist.broadcast(host, src=src, group=process_group)
            tensor.copy_(host)
        else:
            dist.broadcast(tensor, src=src, group=process_group)
    return tensor


def gather_rows(local: torch.Tensor, process_group=None) -> torch.Tensor:
    """Concatenate per-rank tables of different lengths in rank order (latents / transformations for a checkpoint,
    condition values for the atlas weights).  Every rank gets the full table."""
    rank, world = world_info(process_group)
    if world == 1:
        return local
    if local.is_cuda and dist.get_backend(process_group) != "nccl":
        return gather_rows(local.cpu(), process_group).to(local.device)    # gloo: host tensors only
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=process_group)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes)
    pad = local.new_zeros((m, *local.shape[1:]))
    pad[:local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=process_group)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0)


def shard_jobs(n_jobs: int, rank: int, world: int) -> List[int]:
    """Round-robin split of independent jobs ((time point x condition) atlas volumes) over ranks."""
    return list(range(rank, n_jobs, world))


def collect_jobs(mine: Dict[int, torch.Tensor], n_jobs: int, process_group=None, device=None, dst: int = 0):
    """Bring the results of round-robin jobs (`shard_jobs`) to rank `dst`, in job order: the atlas volumes that
    utils.save_atlas stacks over time points (the reference holds the whole atlas on one host, build_atlas.py:181-187).
    Point-to-point, one volume in flight: rank `dst` receives job after job from its owner while every other rank sends
    its own jobs in the same order, so the exchange cannot deadlock.  Tensors travel through `device` (the GPU under
    RCCL, the CPU under gloo).  Returns the list of CPU tensors on `dst`, None elsewhere."""
    rank, world = world_info(process_group)
    if world == 1:
        return [mine[j] for j in range(n_jobs)]
    if device is None:
        device = "cpu" if dist.get_backend(process_group) == "gloo" else torch.device("cuda", torch.cuda.current_device())
    out = [] if rank == dst else None
    for j in range(n_jobs):
        owner = j % world
        if rank == dst:
            if owner == dst:
                out.append(mine[j])
                continue
            meta = torch.zeros(8, dtype=torch.int64, device=device)
            dist.recv(meta, src=owner, group=process_group)
            nd = int(meta[0])
            buf = torch.empty([int(x) for x in meta[1:1 + nd]], dtype=torch.float32, device=device)
            dist.recv(buf, src=owner, group=process_group)
            out.append(buf.cpu())
        elif owner == rank:
            v = mine[j].to(device=device, dtype=torch.float32).contiguous()
            meta = torch.zeros(8, dtype=torch.int64, device=device)
            meta[0] = v.dim()
            meta[1:1 + v.dim()] = torch.tensor(v.shape, dtype=torch.int64)
            dist.send(meta, dst=dst, group=process_group)
            dist.send(v, dst=dst, group=process_group)
    return out
