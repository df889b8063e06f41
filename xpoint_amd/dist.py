"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

The hot path shards by image pair (independent units): there is NO data-path collective.
The only collective is the one-time broadcast of the packed weight blob (81.6 MB fp32) from rank 0 —
the reference has nothing to mirror here (it only knows nn.DataParallel, train.py:149-151)."""
from __future__ import annotations

from typing import Tuple

import torch


def shard_pairs(total_pairs: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block of pair indices [first, first + count) owned by `rank`.  Remainders go to the low ranks."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, rem = divmod(int(total_pairs), world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def broadcast_weights(net, state_dict_fn=None, src: int = 0, device=None, group=None):
    """Rank `src` loads the reference-format state dict (state_dict_fn() -> dict) and packs it into the
    device-format blob; every other rank allocates an empty blob of the same size; one broadcast; every rank
    adopts the blob.  Works with gloo on CPU tensors (tests) and nccl/RCCL on GPU tensors (bench.py)."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    device = torch.device(device) if device is not None else torch.device("cpu")
    if rank == src:
        if state_dict_fn is not None:
            net.load_state_dict(state_dict_fn(), strict=True)
        blob = net.pack_weights().to(device)
    else:
        blob = torch.empty(net.weights_numel(), dtype=torch.float32, device=device)
    if world > 1:
        dist.broadcast(blob, src, group=group)
    if blob.is_cuda:
        net.set_weight_blob(blob)
    return blob
