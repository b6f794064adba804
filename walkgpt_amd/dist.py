"""Data-parallel plumbing of the grounded-segmentation path: one process per GPU, images sharded by rank, ONE exchange
step -- an all-gather of the mask logits (RCCL over xGMI through torch.distributed backend "nccl"; gloo on CPU in tests).

The reference shards evaluation images with a DistributedSampler (evaluation_walkgpt.py:396-402) and exchanges only
tiny IoU counters; the mask-logit all-gather is this build's addition (BASELINE.json north_star).  Masks are ragged
(T_i [SEG] tokens per image, per-image original sizes), so ranks first exchange a small int64 header and then gather
fixed-size flat payloads padded to the largest rank.
"""
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [start, end) slice of `n_items` for `rank`; the first n_items % world ranks get one extra item."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world %d/%d" % (rank, world))
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def _collective_device(group=None) -> torch.device:
    """Where a rank without any local tensor must put its collective buffers: RCCL (backend "nccl") only takes tensors of the
    rank's GPU, gloo takes CPU tensors."""
    if dist.get_backend(group) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def _to_wire(t: torch.Tensor, wire_dtype) -> torch.Tensor:
    """The payload in the dtype it travels in.  SURVEY.md 8d/8e count the exchange at 2 bytes per logit (bf16): thresholding at 0 and
    IoU read the SIGN of a logit, which round-to-nearest keeps (and zero stays zero), so bf16 on the wire halves the one collective of
    the path at no cost to the masks.  fp32 -> bf16 on the GPU goes through the library's cast kernel, anything else through torch."""
    if wire_dtype is None or t.dtype == wire_dtype:
        return t
    if t.is_cuda and t.dtype == torch.float32 and wire_dtype == torch.bfloat16:
        from . import ops
        return ops.cast_bf16(t.contiguous())
    return t.to(wire_dtype)


def all_gather_masks(masks: Sequence[torch.Tensor], group=None, device=None, dtype=None, wire_dtype=None) -> List[List[torch.Tensor]]:
    """masks[i]: [T_i, H_i, W_i] float tensors of this rank's images.  Returns, on every rank, a list over ranks of
    lists over that rank's images.  Collectives: one all_gather of the int64 header, one all_gather of the payload.
    A rank may hold NO masks (fewer images than ranks, or a shard without any [SEG]): it still has to join both collectives with
    buffers on the right device and of the payload dtype the other ranks use -- pass `device` / `dtype` (default: the backend's
    device, float32, which is what Sam.postprocess_masks returns).  wire_dtype (e.g. torch.bfloat16): the payload is cast to it before
    the collective and the gathered masks come back in it (every rank must pass the same)."""
    world = dist.get_world_size(group)
    dev = masks[0].device if len(masks) else (torch.device(device) if device is not None else _collective_device(group))
    dtype = wire_dtype or (masks[0].dtype if len(masks) else (dtype or torch.float32))
    masks = [_to_wire(m, wire_dtype) for m in masks]
    shapes = torch.tensor([list(m.shape) for m in masks], dtype=torch.int64, device=dev).reshape(-1, 3)
    # header 1: images per rank and payload elements per rank
    local = torch.tensor([shapes.shape[0], int(sum(m.numel() for m in masks))], dtype=torch.int64, device=dev)
    counts = torch.empty(world * 2, dtype=torch.int64, device=dev)  # flat in/out: accepted by both RCCL and gloo
    dist.all_gather_into_tensor(counts, local, group=group)
    counts_h = counts.view(world, 2).cpu()
    max_imgs, max_elems = int(counts_h[:, 0].max()), int(counts_h[:, 1].max())
    # header 2: per-image shapes, padded to the largest image count
    shp_pad = torch.zeros(max(max_imgs, 1), 3, dtype=torch.int64, device=dev)
    shp_pad[: shapes.shape[0]] = shapes
    all_shapes = torch.empty(world * shp_pad.numel(), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(all_shapes, shp_pad.reshape(-1), group=group)
    all_shapes_h = all_shapes.view(world, -1, 3).cpu()
    # payload
    flat = torch.zeros(max(max_elems, 1), dtype=dtype, device=dev)
    if len(masks):
        torch.cat([m.reshape(-1) for m in masks], out=flat[: int(local[1])])
    gathered = torch.empty(world * flat.numel(), dtype=dtype, device=dev)
    dist.all_gather_into_tensor(gathered, flat, group=group)
    gathered = gathered.view(world, -1)
    out = []
    for r in range(world):
        off, per_rank = 0, []
        for i in range(int(counts_h[r, 0])):
            t, h, w = (int(v) for v in all_shapes_h[r, i])
            per_rank.append(gathered[r, off: off + t * h * w].view(t, h, w))
            off += t * h * w
        out.append(per_rank)
    return out


def all_gather_masks_uniform(stacked: torch.Tensor, out: torch.Tensor = None, group=None, wire_dtype=None) -> torch.Tensor:
    """Fast path when every rank holds the same [N, H, W] block (bench config): a single collective, no header.  wire_dtype: see
    all_gather_masks; `out` (if given) must have it.  Rank r's block lands at out[r * N : (r + 1) * N]."""
    world = dist.get_world_size(group)
    stacked = _to_wire(stacked, wire_dtype)
    if out is not None and (out.dtype != stacked.dtype or out.numel() != world * stacked.numel()):
        raise ValueError("all_gather_masks_uniform: out is %s with %d elements, the payload %s with %d per rank x %d ranks"
                         % (out.dtype, out.numel(), stacked.dtype, stacked.numel(), world))
    if out is None:
        out = torch.empty((world * stacked.shape[0],) + tuple(stacked.shape[1:]), dtype=stacked.dtype, device=stacked.device)
    dist.all_gather_into_tensor(out.view(-1), stacked.contiguous().view(-1), group=group)
    return out
