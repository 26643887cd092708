"""Multi-GPU join: one process per GPU, one camera tile per rank, all-gatherv of the
per-rank clouds over RCCL (torch.distributed backend "nccl" on ROCm) into the fused cloud.

What it replaces: the reference fuses tiles in ONE process by folding cwipc_join
pairwise (reference python/cwipc/net/source_synchronizer.py:175-188,
python/cwipc/util.py:1330-1332), copying O(n_tile^2) bytes.  Here every rank filters
its own tile on its own GPU and the only exchange step is the concatenation itself.

The exchange is written once, device-agnostic, on torch tensors holding cwipc_point
records as int32[n, 4] (16 bytes per row):
  * counts: all_gather of one int64 per rank;
  * payload: all_gather of rank-local shards padded to the largest count (RCCL has no
    all-gatherv; shards are equal-sized up to a few percent for camera tiles, and after
    voxelisation they are a few hundred KB, i.e. latency-bound on xGMI either way);
  * result order = rank order = tile order, which is the reference's fold order;
  * timestamp = min, cellsize = min over the contributing clouds
    (reference src/cwipc_filters.cpp:411-414); a rank may contribute zero points
    (the synchroniser drops late tiles, source_synchronizer.py:163-171).
The same function runs on CPU tensors over gloo, which is how the tests cover N > 1.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

__all__ = ["tiles_of_rank", "all_gatherv_points", "join_across_ranks"]


def tiles_of_rank(ntiles: int, rank: int, world: int) -> List[int]:
    """Tile t lives on rank t mod world: round-robin, ascending."""
    return [t for t in range(ntiles) if t % world == rank]


def all_gatherv_points(points: torch.Tensor, timestamp: int, cellsize: float, has_cloud: bool = True,
                       group: Optional[dist.ProcessGroup] = None) -> Tuple[torch.Tensor, int, float, List[int]]:
    """Concatenate the per-rank point records in rank order on every rank.

    points: int32[n_local, 4] on the device the process group communicates on.
    Returns (fused int32[n_total, 4], min timestamp, min cellsize, per-rank counts).
    Ranks with has_cloud=False contribute no points and do not take part in the min.
    """
    assert points.dtype == torch.int32 and points.dim() == 2 and points.shape[1] == 4
    world = dist.get_world_size(group)
    dev = points.device
    n_local = points.shape[0]

    # one small all_gather carries count, timestamp, cellsize and the participation flag
    meta = torch.tensor([float(n_local), float(timestamp), float(cellsize), 1.0 if has_cloud else 0.0], dtype=torch.float64, device=dev)
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    table = torch.stack(metas).cpu()
    counts = [int(v) for v in table[:, 0].tolist()]
    part = table[:, 3] > 0
    if bool(part.any()):
        ts = int(table[part, 1].min().item())
        cs = float(table[part, 2].min().item())
    else:
        ts, cs = 0, 0.0
    total, biggest = sum(counts), max(counts)
    if total == 0:
        return torch.empty((0, 4), dtype=torch.int32, device=dev), ts, cs, counts

    padded = torch.zeros((biggest, 4), dtype=torch.int32, device=dev)
    padded[:n_local] = points
    gathered = torch.empty((world, biggest, 4), dtype=torch.int32, device=dev)
    if dev.type == "cuda":
        dist.all_gather_into_tensor(gathered.view(world * biggest, 4), padded, group=group)
    else:   # gloo has no all_gather_into_tensor for every build: list form, same bytes
        parts = [gathered[r] for r in range(world)]
        dist.all_gather(parts, padded, group=group)
    if all(c == biggest for c in counts):
        fused = gathered.view(world * biggest, 4)
    else:
        fused = torch.cat([gathered[r, :counts[r]] for r in range(world)], dim=0)
    return fused, ts, cs, counts


def join_across_ranks(pc, group: Optional[dist.ProcessGroup] = None):
    """All ranks call this with their (device-resident) cloud, or None for "no tile this frame";
    every rank gets the fused cloud as a new cwipc_pointcloud_wrapper.  GPU only."""
    from . import util
    dev = torch.device("cuda", torch.cuda.current_device())
    if pc is None:
        local = torch.empty((0, 4), dtype=torch.int32, device=dev)
        ts, cs, has = 0, 0.0, False
    else:
        n = pc.count()
        local = torch.empty((n, 4), dtype=torch.int32, device=dev)
        if n:
            util.cwipc_hip_copy_device_aos(pc, local.data_ptr(), n * 16)
        ts, cs, has = pc.timestamp(), pc.cellsize(), True
    torch.cuda.current_stream().synchronize()
    if dist.get_backend(group) == "nccl":
        fused, ts, cs, _counts = all_gatherv_points(local, ts, cs, has, group)
    else:
        # a process group without device collectives (gloo: rehearsals on one GPU): the exchange runs on host copies
        fused, ts, cs, _counts = all_gatherv_points(local.cpu(), ts, cs, has, group)
        fused = fused.to(dev)
    torch.cuda.current_stream().synchronize()
    return util.cwipc_hip_from_device_aos(fused.data_ptr() if fused.shape[0] else 0, fused.shape[0], ts, cs)
