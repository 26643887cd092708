"""N > 1 on CPU: the all-gatherv join of cwipc_util_amd.multigpu over gloo, world sizes 2 and 3.

The exchange function is device-agnostic (torch tensors of cwipc_point records as int32[n, 4]);
on the GPU box the same code runs over RCCL.  The expected result is the reference's fold of
cwipc_join over the tiles in tile order (reference python/cwipc/net/source_synchronizer.py:175-188),
computed with the oracle.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _tile_of_rank(rank: int, case: str):
    """The cloud a rank contributes: (points, timestamp, cellsize, has_cloud)."""
    from oracle import oracle
    pts, cs = oracle.synthetic(20000 + 7000 * rank, 0.3 * rank)
    if case == "tiles":
        # every rank filters "its camera" out of its own capture
        out = oracle.tilefilter(pts, 1 + (rank % 2))
        out = out.copy()
        out['tile'] = 1 << rank
        return out, 1000 - rank, cs, True
    if case == "voxelized":
        out, ocs = oracle.downsample(pts, cs, 0.02)
        return out, 500 + rank, ocs, True
    if case == "ragged":
        # rank 0 has no cloud this frame, rank 1 an empty one, later ranks real ones
        if rank == 0:
            return oracle.empty(0), 0, 0.0, False
        if rank == 1:
            return oracle.empty(0), 77, 0.5, True
        return pts[: 1000 * rank + 3], 90 + rank, cs, True
    raise ValueError(case)


def _expected(world: int, case: str):
    from oracle import oracle
    fused, ts, cs = None, None, None
    for r in range(world):
        pts, t, c, has = _tile_of_rank(r, case)
        if not has:
            continue
        if fused is None:
            fused, ts, cs = pts, t, c
        else:
            fused, ts, cs = oracle.join(fused, pts), min(ts, t), min(cs, c)
    if fused is None:
        fused, ts, cs = oracle.empty(0), 0, 0.0
    return fused, ts, cs


def _worker(rank: int, world: int, port: int, case: str, result_dir: str):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        from cwipc_util_amd.multigpu import all_gatherv_points, tiles_of_rank
        pts, ts, cs, has = _tile_of_rank(rank, case)
        local = torch.from_numpy(np.ascontiguousarray(pts).view(np.int32).reshape(-1, 4).copy())
        fused, fts, fcs, counts = all_gatherv_points(local, ts, cs, has)
        exp, ets, ecs = _expected(world, case)
        got = fused.numpy().reshape(-1).view(exp.dtype) if fused.shape[0] else exp[:0]
        assert counts == [len(_tile_of_rank(r, case)[0]) for r in range(world)], counts
        assert len(got) == len(exp), (len(got), len(exp))
        assert got.tobytes() == exp.tobytes(), "fused cloud differs from the cwipc_join fold"
        assert fts == ets, (fts, ets)
        assert fcs == pytest.approx(ecs, rel=0, abs=0), (fcs, ecs)
        # tile -> rank mapping: round robin, every tile exactly once
        owned = [tiles_of_rank(8, r, world) for r in range(world)]
        assert sorted(t for o in owned for t in o) == list(range(8))
        assert all(t % world == rank for t in owned[rank])
        open(os.path.join(result_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,case", [(2, "tiles"), (2, "voxelized"), (3, "ragged"), (2, "ragged")])
def test_all_gatherv_join_gloo(tmp_path, world, case):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, case, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def _stream_worker(rank: int, world: int, port: int, result_dir: str):
    """A stream of frames through SlotExchange: one collective per frame once the slots are sized, regrowth when a
    tile outgrows them, empty and missing tiles in between."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        from cwipc_util_amd.multigpu import SlotExchange
        from oracle import oracle
        ex = SlotExchange(torch.device("cpu"))
        # (points of rank r in frame f, has_cloud)
        def frame_tile(f, r):
            n = [3000, 3500, 0, 3200, 20000, 100, 0, 21000][f] + 137 * r
            if f == 2:
                return oracle.empty(0), r != 0          # an empty frame; rank 0 has no cloud at all
            if f == 6 and r == 1:
                return oracle.empty(0), False
            pts, _ = oracle.synthetic(2 * n + 1000, 0.1 * f + r)
            return pts[:n].copy(), True
        collectives = []
        for f in range(8):
            pts, has = frame_tile(f, rank)
            slot = ex.slot_points(len(pts))
            if len(pts):
                slot.copy_(torch.from_numpy(np.ascontiguousarray(pts).view(np.int32).reshape(-1, 4).copy()))
            fused, ts, cs, counts = ex.gather(len(pts), 1000 + 10 * f - rank, 0.01 * (rank + 1), has)
            collectives.append(ex.collectives)
            exp, ets, ecs = None, None, None
            for r in range(world):
                p, h = frame_tile(f, r)
                if not h:
                    continue
                exp = p if exp is None else oracle.join(exp, p)
                ets = 1000 + 10 * f - r if ets is None else min(ets, 1000 + 10 * f - r)
                ecs = np.float32(0.01 * (r + 1)) if ecs is None else min(ecs, np.float32(0.01 * (r + 1)))
            if exp is None:
                exp, ets, ecs = oracle.empty(0), 0, 0.0
            got = fused.numpy().reshape(-1).view(exp.dtype) if fused.shape[0] else exp[:0]
            assert counts == [len(frame_tile(f, r)[0]) for r in range(world)], (f, counts)
            assert got.tobytes() == exp.tobytes(), f"frame {f}: fused cloud differs from the cwipc_join fold"
            assert ts == ets and cs == pytest.approx(float(ecs), rel=0, abs=0), (f, ts, ets, cs, ecs)
        # frame 0: count exchange + gather; frame 4 and 7 outgrow the slots: one extra gather; all others: one collective
        assert collectives[0] == 2 and collectives[1] == 1 and collectives[2] == 1 and collectives[3] == 1, collectives
        assert collectives[4] == 2 and collectives[5] == 1 and collectives[6] == 1, collectives
        open(os.path.join(result_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_slot_exchange_stream_gloo(tmp_path, world):
    port = _free_port()
    mp.spawn(_stream_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def _pipeline_worker(rank: int, world: int, port: int, result_dir: str):
    """The exchange as JoinPipeline drives it: two SlotExchange objects take the frames in turn, the collective of
    frame f + 1 is launched (async) before the one of frame f is waited for -- including frames that outgrow the slots,
    whose repeat gathers are issued while the next frame's collective is in flight."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        from cwipc_util_amd.multigpu import SlotExchange
        from oracle import oracle
        exs = [SlotExchange(torch.device("cpu")), SlotExchange(torch.device("cpu"))]
        sizes = [3000, 3500, 0, 3200, 20000, 100, 22000, 0, 21000, 50, 30000]

        def frame_tile(f, r):
            n = sizes[f] + 137 * r
            if sizes[f] == 0:
                return oracle.empty(0), r != 0
            pts, _ = oracle.synthetic(2 * n + 1000, 0.1 * f + r)
            return pts[:n].copy(), True

        def check(f, ex, ts, cs, counts):
            exp, ets, ecs = None, None, None
            for r in range(world):
                p, h = frame_tile(f, r)
                if not h:
                    continue
                exp = p if exp is None else oracle.join(exp, p)
                ets = 1000 + 10 * f - r if ets is None else min(ets, 1000 + 10 * f - r)
                ecs = np.float32(0.01 * (r + 1)) if ecs is None else min(ecs, np.float32(0.01 * (r + 1)))
            if exp is None:
                exp, ets, ecs = oracle.empty(0), 0, 0.0
            parts = [ex.recv[r, ex.HEADER_ROWS:ex.HEADER_ROWS + counts[r]] for r in range(world) if counts[r]]
            got = torch.cat(parts, dim=0).numpy().reshape(-1).view(exp.dtype) if parts else exp[:0]
            assert counts == [len(frame_tile(f, r)[0]) for r in range(world)], (f, counts)
            assert got.tobytes() == exp.tobytes(), f"frame {f}: fused cloud differs from the cwipc_join fold"
            assert ts == ets and cs == pytest.approx(float(ecs), rel=0, abs=0), (f, ts, ets, cs, ecs)

        pending = None
        for f in range(len(sizes)):
            ex = exs[f % 2]
            pts, has = frame_tile(f, rank)
            slot = ex.slot_points(len(pts))
            if len(pts):
                slot.copy_(torch.from_numpy(np.ascontiguousarray(pts).view(np.int32).reshape(-1, 4).copy()))
            ex.launch(len(pts), 1000 + 10 * f - rank, 0.01 * (rank + 1), has, async_op=True)
            if pending is not None:
                pex = exs[pending % 2]
                check(pending, pex, *pex.finish())
            pending = f
        pex = exs[pending % 2]
        check(pending, pex, *pex.finish())
        open(os.path.join(result_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_pipelined_exchange_gloo(tmp_path, world):
    port = _free_port()
    mp.spawn(_pipeline_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))
