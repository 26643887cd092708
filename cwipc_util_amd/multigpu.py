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

__all__ = ["tiles_of_rank", "all_gatherv_points", "SlotExchange", "join_across_ranks", "JoinPipeline", "library_comm", "free_library_comms"]


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


class SlotExchange:
    """The same exchange for a stream of frames: one collective per frame instead of two.

    Every rank owns a slot of `cap + 2` rows in a persistent buffer: two header rows
    [count, has_cloud, cellsize bits, 0] [timestamp lo, timestamp hi, 0, 0] and its points.  One
    all_gather of whole slots carries data and metadata together; the capacity is agreed once (a count
    exchange on the first frame) and grown, by all ranks alike, when some rank's count outgrows it
    (every rank sees every header, so all of them take the same decision and repeat the gather).
    """

    HEADER_ROWS = 2

    def __init__(self, device: torch.device, group: Optional[dist.ProcessGroup] = None):
        self.device = device
        self.group = group
        self.world = dist.get_world_size(group)
        self.cap = 0
        self.send: Optional[torch.Tensor] = None
        self.recv: Optional[torch.Tensor] = None
        self.collectives = 0   # for tests: how many collectives the last call needed
        self._overflow: Optional[torch.Tensor] = None
        self._work = None        # the collective in flight (async launch)
        self._n_local = 0
        # the two header rows are written on the host; page-locked when they have to travel to a GPU
        self._header = torch.zeros((self.HEADER_ROWS, 4), dtype=torch.int32, pin_memory=(device.type == "cuda"))
        self._header_np = self._header.numpy()   # same memory: filled with one numpy assignment per frame

    def _resize(self, need: int) -> None:
        # room for 1/16 more than the largest cloud seen (frames of a stream are alike; every byte of a slot crosses
        # the links, used or not), rounded up to 1024 rows
        cap = ((need + need // 16 + 1023) // 1024) * 1024
        self.cap = max(cap, 4096)
        rows = self.cap + self.HEADER_ROWS
        self.send = torch.zeros((rows, 4), dtype=torch.int32, device=self.device)
        self.recv = torch.zeros((self.world, rows, 4), dtype=torch.int32, device=self.device)
        if self.device.type == "cuda":
            torch.cuda.current_stream().synchronize()   # the library's streams write into / read from these buffers

    def _agree_on_capacity(self, n_local: int) -> None:
        mine = torch.tensor([n_local], dtype=torch.int64, device=self.device)
        counts = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(counts, mine, group=self.group)
        self.collectives += 1
        self._resize(max(int(c.item()) for c in counts))

    def slot_points(self, n_local: int) -> torch.Tensor:
        """Where this rank's points go (int32[n_local, 4]): its slot, or a buffer of its own while the slots
        are too small (no collective here: ranks must never communicate on their own).  Fill it, then call gather()."""
        if n_local <= self.cap:
            self._overflow = None
            return self.send[self.HEADER_ROWS:self.HEADER_ROWS + n_local]
        self._overflow = torch.empty((n_local, 4), dtype=torch.int32, device=self.device)
        return self._overflow

    def gather(self, n_local: int, timestamp: int, cellsize: float, has_cloud: bool = True) -> Tuple[torch.Tensor, int, float, List[int]]:
        """After the points are in slot_points(n_local): returns (fused int32[n_total, 4], min timestamp, min cellsize, counts).
        Collective: every rank of the group calls it, once per frame."""
        ts_min, cs_min, counts = self.gather_slots(n_local, timestamp, cellsize, has_cloud)
        if sum(counts) == 0:
            return torch.empty((0, 4), dtype=torch.int32, device=self.device), ts_min, cs_min, counts
        fused = torch.cat([self.recv[r, self.HEADER_ROWS:self.HEADER_ROWS + counts[r]] for r in range(self.world) if counts[r]], dim=0)
        return fused, ts_min, cs_min, counts

    def gather_slots(self, n_local: int, timestamp: int, cellsize: float, has_cloud: bool = True) -> Tuple[int, float, List[int]]:
        """The collective itself: afterwards self.recv[r, HEADER_ROWS : HEADER_ROWS + counts[r]] holds rank r's points.
        Returns (min timestamp, min cellsize, counts); the library turns the slots into a cloud in one pass
        (cwipc_hip_from_device_slots), gather() concatenates them into a tensor."""
        self.launch(n_local, timestamp, cellsize, has_cloud)
        return self.finish()

    def launch(self, n_local: int, timestamp: int, cellsize: float, has_cloud: bool = True, async_op: bool = False) -> None:
        """First half of gather_slots(): header and collective go out; with async_op the call does not wait for the
        collective (finish() does), so that the caller can pack the next frame into ANOTHER SlotExchange meanwhile."""
        import struct
        self.collectives = 0
        if self.cap == 0:
            self._agree_on_capacity(n_local)   # first frame, all ranks alike
            self._adopt_overflow(n_local)
        ts = int(timestamp) & 0xffffffffffffffff
        as_i32 = lambda v: v - (1 << 32) if v >= (1 << 31) else v
        cs_bits = struct.unpack("<i", struct.pack("<f", float(cellsize)))[0]
        self._header_np[:] = ((n_local, 1 if has_cloud else 0, cs_bits, 0), (as_i32(ts & 0xffffffff), as_i32(ts >> 32), 0, 0))
        self._n_local = n_local
        self._send_once(async_op)

    def _send_once(self, async_op: bool) -> None:
        self.send[:self.HEADER_ROWS].copy_(self._header, non_blocking=True)   # (the read-back of the headers in finish() waits for it)
        rows = self.cap + self.HEADER_ROWS
        if self.device.type == "cuda":
            self._work = dist.all_gather_into_tensor(self.recv.view(self.world * rows, 4), self.send, group=self.group, async_op=async_op)
        else:   # gloo: list form, same bytes
            self._work = dist.all_gather([self.recv[r] for r in range(self.world)], self.send, group=self.group, async_op=async_op)
        self.collectives += 1

    def finish(self) -> Tuple[int, float, List[int]]:
        """Second half of gather_slots(): waits for the collective, reads the headers, and gathers again (all ranks
        alike) while some rank's cloud does not fit the slots."""
        import struct
        n_local = self._n_local
        while True:
            if self._work is not None:
                self._work.wait()
                self._work = None
            heads = self.recv[:, :self.HEADER_ROWS, :].cpu().numpy()   # (world, 2, 4) int32; waits for the collective
            counts = [int(v) for v in heads[:, 0, 0]]
            if max(counts) <= self.cap:
                break
            # somebody's cloud outgrew the slots (its points were not sent): every rank sees that in the
            # headers, all grow alike and send again
            keep = None if self._overflow is not None else self.send[self.HEADER_ROWS:self.HEADER_ROWS + n_local].clone()
            self._resize(max(counts))
            if keep is not None:
                self.send[self.HEADER_ROWS:self.HEADER_ROWS + n_local] = keep
            self._adopt_overflow(n_local)
            self._send_once(False)
        # min timestamp and min cellsize over the ranks that had a cloud (plain Python on 2 x world numbers)
        stamps, sizes = [], []
        for r in range(self.world):
            if heads[r, 0, 1] > 0:
                stamps.append(((int(heads[r, 1, 1]) & 0xffffffff) << 32) | (int(heads[r, 1, 0]) & 0xffffffff))
                sizes.append(struct.unpack("<f", struct.pack("<i", int(heads[r, 0, 2])))[0])
        ts_min, cs_min = (min(stamps), min(sizes)) if stamps else (0, 0.0)
        return ts_min, cs_min, counts

    def _adopt_overflow(self, n_local: int) -> None:
        if self._overflow is not None and n_local <= self.cap:
            self.send[self.HEADER_ROWS:self.HEADER_ROWS + n_local] = self._overflow
            self._overflow = None


_exchanges = {}


def _torch_stream() -> int:
    """torch's current stream on the current device as a hipStream_t (an integer).  The library's pack and unpack kernels
    run as steps of it, where the collectives are ordered too, so that neither needs a wait on the host.  Looked up on
    every call: the caller may be inside `with torch.cuda.stream(s)` or on another thread than last time, and a stale
    handle would put the two kernels on a stream the collective is not ordered against."""
    return torch.cuda.current_stream().cuda_stream


def _pack(ex: SlotExchange, pc, dev: torch.device, staged: bool) -> Tuple[int, int, float, bool]:
    """This rank's points into its send slot of `ex` (or its overflow buffer); returns (n, timestamp, cellsize, has_cloud)."""
    from . import util
    if pc is None:
        n, ts, cs, has = 0, 0, 0.0, False
    else:
        n, ts, cs, has = pc.count(), pc.timestamp(), pc.cellsize(), True
    slot = ex.slot_points(n)
    if n:
        if staged:
            tmp = torch.empty((n, 4), dtype=torch.int32, device=dev)
            util.cwipc_hip_copy_device_aos(pc, tmp.data_ptr(), n * 16)
            slot.copy_(tmp)
        else:
            # a step of torch's stream: behind that stream's last use of the slot, in front of the collective that sends it
            util.cwipc_hip_copy_device_aos(pc, slot.data_ptr(), n * 16, stream=_torch_stream())
    return n, ts, cs, has


def _unpack(ex: SlotExchange, counts: List[int], ts: int, cs: float, dev: torch.device, staged: bool):
    """The receive buffer of `ex` as the fused cloud: slots -> planes in one pass of the library."""
    from . import util
    if staged:
        recv = ex.recv.to(dev)   # (world, cap + HEADER_ROWS, 4) int32
        torch.cuda.current_stream().synchronize()
        return util.cwipc_hip_from_device_slots(recv.data_ptr(), recv.shape[1], ex.HEADER_ROWS, counts, ts, cs)
    # device path: a step of torch's stream, behind the collective that filled recv (its headers have been read back) and
    # in front of the next collective into the same buffer; the fused cloud carries an event, nobody waits here
    return util.cwipc_hip_from_device_slots(ex.recv.data_ptr(), ex.recv.shape[1], ex.HEADER_ROWS, counts, ts, cs, stream=_torch_stream())


_library_comms = {}


def library_comm(group: Optional[dist.ProcessGroup] = None):
    """This rank's end of the exchange INSIDE the library (util.cwipc_hip_comm: RCCL linked into libcwipc_util.so, one C call
    per frame, nothing of torch on the per-frame path).  torch.distributed only carries the 128-byte id from rank 0 to the
    others, once; any backend will do for that.  Collective on first use per (group, device)."""
    from . import util
    key = (id(group), util.cwipc_util_dll_load().cwipc_hip_get_device())
    comm = _library_comms.get(key)
    if comm is None:
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [util.cwipc_hip_comm_unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        comm = _library_comms[key] = util.cwipc_hip_comm(box[0], rank, world)
    return comm


def free_library_comms() -> None:
    """Destroy the library-side communicators made by library_comm() (every rank, before the process group goes)."""
    for comm in _library_comms.values():
        comm.free()
    _library_comms.clear()


def join_across_ranks(pc, group: Optional[dist.ProcessGroup] = None, exchange: Optional[str] = None):
    """All ranks call this with their (device-resident) cloud, or None for "no tile this frame";
    every rank gets the fused cloud as a new cwipc_pointcloud_wrapper.  GPU only.

    exchange = "library" (the default on an RCCL group): the library's own exchange, see library_comm().
    exchange = "torch" (the default on any other backend, e.g. gloo rehearsals): the protocol written out on torch.distributed --
    frame after frame one collective (SlotExchange); the library's SoA -> AoS kernel writes straight into the send slot."""
    if exchange is None:
        exchange = "library" if dist.get_backend(group) == "nccl" else "torch"
    if exchange == "library":
        return library_comm(group).join(pc)
    dev = torch.device("cuda", torch.cuda.current_device())
    staged = dist.get_backend(group) != "nccl"   # no device collectives (gloo: rehearsals on one GPU): slots live on the host
    key = (id(group), dev.index, staged)
    ex = _exchanges.get(key)
    if ex is None:
        ex = _exchanges[key] = SlotExchange(torch.device("cpu") if staged else dev, group)
    n, ts, cs, has = _pack(ex, pc, dev, staged)
    ts, cs, counts = ex.gather_slots(n, ts, cs, has)
    return _unpack(ex, counts, ts, cs, dev, staged)


class JoinPipeline:
    """join_across_ranks for a stream of frames, one frame deep: submit(frame i + 1) packs that frame and sends its
    collective off BEFORE it waits for the collective of frame i, so the time on the wire lies behind the packing of
    the next frame (and behind whatever the caller does between two submits).  Two SlotExchange objects take the frames
    in turn.  Every rank calls submit() with its cloud of the same frame (or None), in the same order; collectives stay
    in step because each rank issues them in that one order, the repeats after an outgrown slot included."""

    def __init__(self, group: Optional[dist.ProcessGroup] = None):
        self.dev = torch.device("cuda", torch.cuda.current_device())
        self.staged = dist.get_backend(group) != "nccl"
        where = torch.device("cpu") if self.staged else self.dev
        self.ex = [SlotExchange(where, group), SlotExchange(where, group)]
        self.pending: Optional[int] = None   # index of the exchange whose collective is in flight
        self.frames = 0

    def submit(self, pc):
        """Returns the fused cloud of the PREVIOUS frame (None for the first frame)."""
        k = self.frames % 2
        self.frames += 1
        ex = self.ex[k]
        n, ts, cs, has = _pack(ex, pc, self.dev, self.staged)
        ex.launch(n, ts, cs, has, async_op=True)
        out = self._finish(self.pending) if self.pending is not None else None
        self.pending = k
        return out

    def flush(self):
        """The fused cloud of the last frame submitted (None if there is none outstanding)."""
        out = self._finish(self.pending) if self.pending is not None else None
        self.pending = None
        return out

    def _finish(self, k: int):
        ex = self.ex[k]
        ts, cs, counts = ex.finish()
        return _unpack(ex, counts, ts, cs, self.dev, self.staged)


