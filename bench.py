#!/usr/bin/env python3
"""Benchmark of the hot path: voxel downsample of a 10 M-point synthetic cloud.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): cwipc_synthetic(npoints=10 000 000) -> N = 9 998 244 points
(159 971 904 B), device-resident; one step = one cwipc_downsample(pc, +0.01) through the C-ABI
(octree-split path, the reference's default).  Steps rotate over 4 distinct device copies
(640 MB > the 256 MB Infinity Cache) so every step reads cold HBM.  In a stream of calls the library hands a
result out while its kernels still run (it settles on first use); the timed region ends with a wait for everything.

N > 1 (weak scaling): every rank holds one camera tile of the same size (tile mask 1 << rank),
runs the same step on its own GPU, and the per-rank results are fused by the all-gatherv join -- the path's one
exchange step: RCCL inside the library, pipelined inside the library (cwipc_hip_comm_submit: one C call per frame that
returns at once; a thread of the communicator does the waiting, so the join of frame i overlaps the downsample of frame
i + 1; all joins are complete when the timed region ends).  torch.distributed runs on gloo and only carries the
communicator's id, the fences and the sum of the timing: a torch NCCL process group in the process costs every step ~20 us
whether it is used or not.  CWIPC_BENCH_EXCHANGE=torch selects the same protocol on torch.distributed
(cwipc_util_amd.multigpu, with a Python worker thread and an RCCL process group made for it), which is also what the
bench falls back to, on all ranks together, should the library's exchange fail its preflight against it -- a comparison
on a real frame, under a time limit.  For N > 1 the library is told to leave 24 compute units out of the voxel kernel's
persistent grid (CWIPC_SPARE_CUS, unless already set; the default is one per XCD), so that the exchange's kernels find
room.  Rehearsal knobs: more ranks than GPUs (several ranks share a GPU, torch exchange staged through the host),
CWIPC_BENCH_FORCE_JOIN=1 (N = 1 with a one-rank communicator: the whole N > 1 step but the wire),
CWIPC_BENCH_PIPELINE=0 (join and downsample one after the other), CWIPC_BENCH_JOIN_ASYNC=0 (torch exchange: the worker
waits for every frame's collective before it takes the next frame).

One JSON line on rank 0, the only thing written to stdout.  `value` = points filtered by all ranks / wall time of the K
timed steps (max over ranks), inputs resident in HBM.  `roofline` = algorithmic bytes of the dominant kernel (16 B per
input point + 16 B per output point, all N points in one launch) / its mean launch duration, measured with hipEvents on
the library's stream during a second pass over the same steps (each call waited for, so the kernel runs alone);
`traffic` = HBM bytes per launch from the newest committed PMC passes (profiles/rNN_traffic.json).  `cpu_baseline` = the
CPU oracle (a single-threaded C restatement of the PCL algorithm, NOT PCL) on the same cloud, rank 0, N = 1 only: one
core, and (`all_cores`) every core of the host running one cloud each.  Sub-records in the same line: `config4` =
BASELINE configs[3], the 8 x 2 M-tile capture (tilefilter -> downsample per tile, n-ary join, join across ranks), tile t
on rank t mod N, strong scaling; `config3` = BASELINE configs[2], outlier removal of the 10 M cloud (N = 1 only), with its
own roofline fraction over all kernels of a call; `config5` = BASELINE configs[4], the live-size stream (8 x 300 k-point tiles per
frame through colorize -> downsample -> outliers -> join, >= 300 frames: frames/s, p50 / p99 frame latency, what PCIe adds).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np

NPOINTS_ARG = 10_000_000
CELLSIZE = 0.01
NCOPIES = 4
HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec


def make_input(cwipc, npoints_arg: int, angle: float):
    """The synthetic cloud through the product's own source (amd-fixangle pins the colours)."""
    import struct
    src = cwipc.cwipc_synthetic(0, npoints_arg)
    src.start()
    out = bytearray(4)
    assert src.auxiliary_operation("amd-fixangle", struct.pack("f", angle), out)
    pc = src.get()
    src.stop()
    return pc


def measured_traffic(n_points: int, kernel: str):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC passes
    (profiles/rNN_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs of this
    bench, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  None if the committed
    numbers are for another workload or kernel."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload_points") == n_points and d.get("kernel") == kernel:
            return d.get("hbm_bytes_per_launch"), os.path.basename(f)
    return None, None


def cpu_baseline(points: np.ndarray, pc_cellsize: float, budget_s: float = 10.0):
    """Time the oracle's downsample on the host.  Test infrastructure used as the CPU baseline.  Two legs: one thread
    (the like-for-like figure: the reference filter is single-threaded, no OpenMP or threads in src/cwipc_filters.cpp),
    and every core of this host running the same single-threaded job on a cloud each (a stream of frames is what
    parallelises on the CPU side too; `cores` says how many)."""
    from oracle import oracle
    try:
        lib = oracle.load(native=True)   # -march=native copy built on this host
        kind_note = "gcc -O3 -march=native"
    except Exception:
        lib = oracle.load()
        kind_note = "gcc -O3"
    import ctypes

    def one_run(out, ocs):
        t0 = time.perf_counter()
        m = lib.oracle_downsample(points.ctypes.data, len(points), pc_cellsize, CELLSIZE, out.ctypes.data, len(out),
                                  ctypes.addressof(ocs), None, None)
        assert m > 0
        return time.perf_counter() - t0, int(m)

    out = np.zeros(len(points), dtype=oracle.POINT_DTYPE)
    ocs = ctypes.c_float(0)
    times = []
    t_end = time.perf_counter() + budget_s
    while len(times) < 3 or (time.perf_counter() < t_end and len(times) < 20):
        dt, m = one_run(out, ocs)
        times.append(dt)
    best = float(np.median(times))
    result = {
        "value": len(points) / best / 1e6,
        "unit": "Mpoints/s",
        "cores": 1,
        "kind": "port",
        "sample": f"full workload ({len(points)} points), median of {len(times)} runs, {kind_note}; "
                  "C restatement of pcl::VoxelGrid + octree split, not PCL",
        "outputs": int(m),
    }
    # all cores: one single-threaded job per core at the same time (ctypes releases the GIL), three rounds
    from concurrent.futures import ThreadPoolExecutor
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 64))
    bufs = [(np.zeros(len(points), dtype=oracle.POINT_DTYPE), ctypes.c_float(0)) for _ in range(cores)]
    rounds = []
    with ThreadPoolExecutor(max_workers=cores) as pool:
        for _ in range(3):
            t0 = time.perf_counter()
            list(pool.map(lambda b: one_run(*b), bufs))
            rounds.append(time.perf_counter() - t0)
    result["all_cores"] = {
        "value": cores * len(points) / float(np.median(rounds)) / 1e6,
        "unit": "Mpoints/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{cores} concurrent single-threaded runs of the full workload (one cloud per core), median of 3 rounds: "
                  "aggregate throughput of a stream of frames, not the latency of one",
    }
    return result


def parity_record(points: np.ndarray, pc_cellsize: float, got: np.ndarray):
    """The HIP result of the timed workload against the oracle's, on the full cloud (the checker, not the product; rank 0, N = 1
    only, next to the cpu_baseline leg).  north_star's bar is 1e-5 on the centroids: it is asserted by the tests for voxels of at
    most 300 points; above, the (restated) fp32 running sums of pcl::VoxelGrid drift by themselves, which the two distances from
    the float64 mean of each voxel's points show.  Voxel set, order, colours and tiles must be identical."""
    from oracle import oracle
    exp, _cs, mean64, count = oracle.downsample_audit(points, pc_cellsize, CELLSIZE)
    rec = {"outputs_hip": int(len(got)), "outputs_oracle": int(len(exp)), "same_count": bool(len(got) == len(exp))}
    if len(got) != len(exp) or not len(exp):
        return rec
    pop = count.astype(np.int64)
    small = pop <= 300
    err = np.zeros(len(exp)); e64 = np.zeros(len(exp)); o64 = np.zeros(len(exp))
    for i, f in enumerate(("x", "y", "z")):
        g = got[f].astype(np.float64)
        err = np.maximum(err, np.abs(g - exp[f].astype(np.float64)))
        e64 = np.maximum(e64, np.abs(g - mean64[:, i]))
        o64 = np.maximum(o64, np.abs(exp[f].astype(np.float64) - mean64[:, i]))
    rec.update({
        "xyz_max_pop_le_300": float(err[small].max()) if small.any() else 0.0,
        "xyz_max_pop_gt_300": float(err[~small].max()) if (~small).any() else 0.0,
        "outputs_gt_300": int((~small).sum()),
        "largest_population": int(pop.max()),
        "hip_vs_f64_max": float(e64.max()),
        "oracle_vs_f64_max": float(o64.max()),
        "rgb_tile_exact": bool(all((got[f] == exp[f]).all() for f in ("r", "g", "b", "tile"))),
        "tolerance": "north_star: centroids within 1e-5 of the reference; holds against the oracle for voxels of <= 300 points, "
                     "above that the oracle's own fp32 running sum is what moves (oracle_vs_f64_max), not the HIP path (hip_vs_f64_max)",
        "oracle": "CPU restatement of pcl::VoxelGrid + octree split (oracle/cwipc_oracle.c), parity unpinned: PCL is not in the image",
    })
    return rec


def call_then_count(cwipc, clouds, cellsize: float, runs: int = 40):
    """One call followed by count(), as the reference's caller does (python/cwipc/filters/voxelize.py:28-37): the latency of a
    single downsample with its result settled, median over `runs` calls on rotating (cold) inputs, in microseconds."""
    lib = cwipc.util.cwipc_util_dll_load()
    for i in range(6):
        cwipc.cwipc_downsample(clouds[i % len(clouds)], cellsize).count()
    lib.cwipc_hip_synchronize()
    times = []
    for i in range(runs):
        t0 = time.perf_counter()
        cwipc.cwipc_downsample(clouds[i % len(clouds)], cellsize).count()
        times.append(time.perf_counter() - t0)
    lib.cwipc_hip_synchronize()
    return float(np.median(times)) * 1e6


def bench_config4(cwipc, rank: int, world: int, steps: int, warmup: int, fence, join_across_ranks):
    """BASELINE configs[3]: the 8-tile capture (8 x synthetic(2 000 000), camera mask 1 << i, rotated i x 45 degrees),
    tile t on rank t mod world; per tile tilefilter(1 << t) -> downsample(0.01); the local results joined (n-ary join),
    then the all-gatherv join across ranks.  Strong scaling: the job is the same 8 tiles whatever N is."""
    from cwipc_util_amd.capture import capture_tile, per_tile_chain
    from cwipc_util_amd.multigpu import tiles_of_rank
    NT, NP = 8, 2_000_000
    mine = tiles_of_rank(NT, rank, world)
    tiles = []
    for t in mine:
        pc = capture_tile(NP, t, NT, timestamp=1000 + t)
        cwipc.cwipc_hip_upload(pc, drop_host_copy=True)
        tiles.append((t, pc))
    n_tile = tiles[0][1].count() if tiles else 0

    # The tiles of a frame are filtered on a few threads of this rank, as the reference's per-tile decoders are threads
    # (net/source_synchronizer.py:128-149); every thread has its own streams and workspaces in the library, so the small kernels
    # of different tiles overlap.  CWIPC_BENCH_TILE_THREADS=1 filters them one after the other.
    nthreads = max(1, min(int(os.environ.get("CWIPC_BENCH_TILE_THREADS", "4")), len(tiles)))
    pool = None
    if nthreads > 1:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=nthreads)

    def some_tiles(k):
        return [(i, per_tile_chain(pc, t, CELLSIZE)) for i, (t, pc) in enumerate(tiles) if i % nthreads == k]

    def frame():
        if pool is None:
            outs = [per_tile_chain(pc, t, CELLSIZE) for t, pc in tiles]
        else:
            done = sorted((item for part in pool.map(some_tiles, range(nthreads)) for item in part), key=lambda item: item[0])
            outs = [o for _, o in done]   # tile order = the reference's fold order
        local = cwipc.cwipc_join_multi(outs) if outs else None
        return join_across_ranks(local) if join_across_ranks is not None else local

    fused = None
    for _ in range(warmup):
        fused = frame()
    if fused is not None:
        fused.count()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        fused = frame()
    if fused is not None:
        fused.count()   # (a submitted join settles here: this frame's exchange, and every earlier one's, has been issued)
    fence()
    elapsed = time.perf_counter() - t0
    if pool is not None:
        pool.shutdown()
    return elapsed, n_tile, (fused.count() if fused is not None else 0), len(mine), nthreads


def bench_config5(cwipc, rank: int, world: int, frames: int, warmup: int, fence, join_across_ranks):
    """BASELINE configs[4]: the live-size stream.  Per frame 8 tiles x synthetic(300 000) (the capture of config 4 at a tenth
    of the size), tile t on rank t mod world; per tile the full chain colorize(0.8, "camera") -> downsample(0.01) ->
    remove_outliers(16, 1.0, false) (reference loop: python/cwipc/scripts/_scriptsupport.py:346-390 feeding
    registration/util.py:91-96 and :170-182); the tiles of a rank joined (n-ary join), then the join across ranks.  A frame
    is complete when its fused cloud's points exist (count()): that wall time is the frame's latency; frames are NOT
    overlapped here, so fps = 1 / mean latency is the conservative figure.  Inputs resident in HBM; a second, shorter pass
    hands host arrays in and takes the fused cloud back as a host array (what PCIe adds)."""
    from cwipc_util_amd.capture import capture_tile
    from cwipc_util_amd.filters.colorize import ColorizeFilter
    from cwipc_util_amd.multigpu import tiles_of_rank
    NT, NP = 8, 300_000
    mine = tiles_of_rank(NT, rank, world)
    tiles = []
    for t in mine:
        pc = capture_tile(NP, t, NT, timestamp=2000 + t)
        cwipc.cwipc_hip_upload(pc, drop_host_copy=True)
        tiles.append(pc)
    n_tile = tiles[0].count() if tiles else 0
    flt = ColorizeFilter(0.8, "camera")
    nthreads = max(1, min(int(os.environ.get("CWIPC_BENCH_TILE_THREADS", "4")), len(tiles)))
    pool = None
    if nthreads > 1:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=nthreads)

    def chain(pc):
        pc = flt.filter(pc)
        pc = cwipc.cwipc_downsample(pc, CELLSIZE)
        return cwipc.cwipc_remove_outliers(pc, 16, 1.0, False)

    def frame(inputs):
        outs = list(pool.map(chain, inputs)) if pool is not None else [chain(pc) for pc in inputs]
        local = cwipc.cwipc_join_multi(outs) if outs else None
        fused = join_across_ranks(local) if join_across_ranks is not None else local
        return fused

    for _ in range(warmup):
        f = frame(tiles)
        if f is not None:
            f.count()
    fence()
    lat = []
    fused_points = 0
    t_all = time.perf_counter()
    for _ in range(frames):
        t0 = time.perf_counter()
        f = frame(tiles)
        fused_points = f.count() if f is not None else 0
        lat.append(time.perf_counter() - t0)
    fence()
    elapsed = time.perf_counter() - t_all
    # what PCIe adds: host arrays in, host array out (rank-local part of the chain only).  Two ways: ordinary numpy arrays through
    # cwipc_from_numpy_array on the calling thread and get_numpy_array() at the end (what a caller of the reference does), and
    # (r4) arrays in page-locked memory (cwipc_hip_pinned_points), every tile's cloud made by the thread that filters it -- so
    # that tile t's upload runs while tile t - 1 computes -- and the fused cloud written into a page-locked array (copy_into)
    host_ms = host_pinned_ms = upload_gbps = None
    if tiles and world == 1:
        arrays = [(pc.get_numpy_array().copy(), pc.cellsize(), pc.timestamp()) for pc in tiles]
        times = []
        for i in range(20 + 3):
            t0 = time.perf_counter()
            ins = []
            for a, cs, ts in arrays:
                pc = cwipc.cwipc_from_numpy_array(a, ts)
                pc._set_cellsize(cs)
                ins.append(pc)
            f = frame(ins)
            f.get_numpy_array()
            if i >= 3:
                times.append(time.perf_counter() - t0)
        host_ms = float(np.median(times)) * 1e3
        pinned = []
        for a, cs, ts in arrays:
            p = cwipc.cwipc_hip_pinned_points(len(a))
            p[:] = a
            pinned.append((p, cs, ts))
        out_buf = cwipc.cwipc_hip_pinned_points(sum(len(a) for a, _, _ in arrays))

        def chain_from_host(item):
            a, cs, ts = item
            pc = cwipc.cwipc_from_numpy_array(a, ts)
            pc._set_cellsize(cs)
            return chain(pc)

        times, up = [], []
        for i in range(20 + 3):
            t0 = time.perf_counter()
            outs = list(pool.map(chain_from_host, pinned)) if pool is not None else [chain_from_host(it) for it in pinned]
            f = cwipc.cwipc_join_multi(outs)
            f.copy_into(out_buf[:f.count()])
            if i >= 3:
                times.append(time.perf_counter() - t0)
        host_pinned_ms = float(np.median(times)) * 1e3
        for i in range(10):   # the upload alone: eight tiles from page-locked arrays, one after the other on this thread
            t0 = time.perf_counter()
            keep = [cwipc.cwipc_from_numpy_array(a, ts) for a, _, ts in pinned]
            up.append(time.perf_counter() - t0)
            del keep
        upload_gbps = sum(a.nbytes for a, _, _ in pinned) / float(np.median(up)) / 1e9
    if pool is not None:
        pool.shutdown()
    lat_ms = np.array(lat) * 1e3
    return {"elapsed": elapsed, "n_tile": n_tile, "fused_points": fused_points, "tiles_here": len(mine), "threads": nthreads,
            "p50_ms": float(np.percentile(lat_ms, 50)), "p99_ms": float(np.percentile(lat_ms, 99)), "host_io_ms_per_frame": host_ms,
            "host_pinned_io_ms_per_frame": host_pinned_ms, "pinned_upload_gbps": upload_gbps}


def bench_config3(cwipc, pc, n: int, runs: int = 5):
    """BASELINE configs[2]: cwipc_remove_outliers(k = 16, sigma = 1.0, perTile = false) on the 10 M-point cloud.
    Algorithmic bytes: 16 N + 8 N (the d_i round trip) + 16 N_keep."""
    sync = cwipc.util.cwipc_util_dll_load().cwipc_hip_synchronize
    kept = cwipc.cwipc_remove_outliers(pc, 16, 1.0, False).count()   # warm-up: grid workspaces
    sync()
    times = []
    for _ in range(runs):
        t0 = time.perf_counter()
        cwipc.cwipc_remove_outliers(pc, 16, 1.0, False)
        sync()
        times.append(time.perf_counter() - t0)
    with cwipc.cwipc_hip_profile() as prof:
        cwipc.cwipc_remove_outliers(pc, 16, 1.0, False)
    kernel_ms = sum(v[0] for v in prof.kernels.values())
    ms = float(np.median(times)) * 1e3
    alg = 16 * n + 8 * n + 16 * kept
    return {
        "workload": "cwipc_synthetic(10000000) -> cwipc_remove_outliers(16, 1.0, false) [BASELINE configs[2]]",
        "ms_per_call": ms, "value": n / ms / 1e3, "unit": "Mpoints/s", "kept": kept, "runs": runs,
        "roofline": {"bound": "hbm", "algorithmic_bytes_per_call": alg, "all_kernels_ms_per_call": kernel_ms,
                     "achieved": alg / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else None, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": alg / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if kernel_ms > 0 else None,
                     "note": "no target (SURVEY section 8d): the k-NN pass is latency / LDS bound, not HBM bound"},
        "kernels": {k: {"ms_total": v[0], "launches": v[1]} for k, v in prof.kernels.items()},
    }


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--npoints", type=int, default=NPOINTS_ARG)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config4", action="store_true", help="skip the 8-tile capture sub-record")
    ap.add_argument("--no-config3", action="store_true", help="skip the outlier-removal sub-record (N = 1 only)")
    ap.add_argument("--config4-steps", type=int, default=30)
    ap.add_argument("--no-config5", action="store_true", help="skip the live-size stream sub-record")
    ap.add_argument("--config5-frames", type=int, default=300)
    args = ap.parse_args()

    # Exactly ONE line goes to stdout: the JSON line.  Libraries that print there on their own (RCCL writes a five-line
    # version banner to stdout when a communicator is made) are sent to stderr: descriptor 1 points at stderr until the
    # line is printed.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    if world > 1:
        # RCCL hands device buffers from process to process with HIP IPC, and this image's host driver supports only the dmabuf
        # flavour: without this setting the library refuses to make a communicator between processes (and torch's RCCL group
        # would die in hipIpcGetMemHandle).  Set here, before torch or the library is imported -- nothing has touched the
        # GPU yet, so the process needs no restart.
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world > 1 or os.environ.get("CWIPC_BENCH_FORCE_JOIN") == "1":
        # the join's small kernels (pack, collective, unpack) run next to the next frame's downsample: leave them a few
        # compute units (the library reads this when it is first used)
        os.environ.setdefault("CWIPC_SPARE_CUS", "24")
    import torch
    import cwipc_util_amd as cwipc

    if cwipc.cwipc_hip_device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the filter path has no CPU fallback")
    # One rank per GPU.  torch.distributed only bootstraps (it carries the library's RCCL id), fences and sums the timing: its
    # process group is gloo.  A torch NCCL process group in the process costs every step ~20 us whether it is used or not
    # (measured on one GPU: 84 instead of 62 us per step with the join -- its watchdog threads and the HIP runtime), so one
    # is made only if the bench has to fall back to the torch.distributed exchange.  When there are more ranks than GPUs
    # (a rehearsal on a one-GPU box: RCCL refuses two ranks on one device) the ranks share GPUs and the exchange is the
    # torch one, staged through the host.
    backend = os.environ.get("CWIPC_BENCH_BACKEND", "gloo")
    own_gpu = world <= torch.cuda.device_count()
    device_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    cwipc.cwipc_hip_set_device(device_index)
    dist = None
    # CWIPC_BENCH_FORCE_JOIN=1: rehearsal of the N > 1 step (downsample + join over RCCL, pipelined) on ONE GPU,
    # with a one-rank process group: everything of the multi-GPU path but the wire
    force_join = world == 1 and os.environ.get("CWIPC_BENCH_FORCE_JOIN") == "1"
    if world > 1 or force_join:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29555")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)

    # ---- input: 4 device-resident copies of this rank's tile ----
    base = make_input(cwipc, args.npoints, angle=0.25 * rank)
    pts = base.get_numpy_array().copy()
    if world > 1:
        pts['tile'] = 1 << (rank % 8)     # camera mask of this rank's tile
    pc_cellsize = base.cellsize()
    n = len(pts)
    clouds = []
    for _ in range(NCOPIES):
        pc = cwipc.cwipc_from_numpy_array(pts, 1000 + rank)
        pc._set_cellsize(pc_cellsize)
        cwipc.cwipc_hip_upload(pc, drop_host_copy=True)
        clouds.append(pc)
    del base

    # N > 1: the join of frame i (a worker thread, as in the reference: net/source_synchronizer.py:17,184 joins on
    # a worker thread too) overlaps the downsample of frame i + 1; at most two frames are in flight, and every
    # frame's join has finished when the timed region ends.  CWIPC_BENCH_PIPELINE=0 runs them one after the other.
    joining = world > 1 or force_join
    pipelined = joining and os.environ.get("CWIPC_BENCH_PIPELINE", "1") != "0"
    joiner = None
    exchange, exchange_note = None, None
    if joining:
        from cwipc_util_amd import multigpu
        from cwipc_util_amd.multigpu import JoinPipeline
        # "library": RCCL inside libcwipc_util.so, one C call per frame (cwipc_hip_comm_join); "torch": the same protocol on
        # torch.distributed (multigpu.SlotExchange).  The library's exchange is checked against the other on a real frame first
        # and all ranks fall back together if any of them disagrees.
        exchange = os.environ.get("CWIPC_BENCH_EXCHANGE", "library" if own_gpu else "torch")
        abandon_at_exit = False
        torch_group = [None]   # the group the torch exchange runs on: the default one, or an RCCL group made for it

        def use_rccl_for_torch_exchange():
            if backend != "nccl" and own_gpu and torch_group[0] is None:
                torch_group[0] = dist.new_group(backend="nccl")   # collective: every rank comes here together
        if exchange == "library":
            # the preflight runs on a thread of its own with a time limit: two ranks on RCCL have never run where this was
            # written (one GPU), and a bench that hangs measures nothing.  A rank whose preflight does not come back leaves
            # that thread behind, all ranks fall back to the torch exchange together, and the process ends with os._exit.
            import threading
            verdict = {}

            def preflight():
                try:
                    torch.cuda.set_device(device_index)
                    probe = cwipc.cwipc_downsample(clouds[0], CELLSIZE)
                    a = multigpu.join_across_ranks(probe, exchange="library")
                    b = multigpu.join_across_ranks(probe, exchange="torch")
                    good = a.count() == b.count() and a.timestamp() == b.timestamp() and a.cellsize() == b.cellsize() \
                        and bool(np.array_equal(a.get_numpy_array(), b.get_numpy_array()))
                    verdict["good"], verdict["why"] = good, "" if good else "results differ from the torch.distributed exchange"
                except Exception as e:   # noqa: BLE001 -- whatever it is, the bench goes on with the other exchange
                    verdict["good"], verdict["why"] = False, f"{type(e).__name__}: {e}"

            th = threading.Thread(target=preflight, daemon=True)
            th.start()
            th.join(float(os.environ.get("CWIPC_BENCH_PREFLIGHT_S", "120")))
            if th.is_alive():
                verdict.setdefault("good", False)
                verdict.setdefault("why", "no answer within the time limit")
                abandon_at_exit = True
            good, why = verdict.get("good", False), verdict.get("why", "")
            # (second word: does any rank leave a stuck thread behind?  Then every rank ends with os._exit, none waits for another)
            flag = torch.tensor([1 if good else 0, 0 if abandon_at_exit else 1], dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            abandon_at_exit = int(flag[1].item()) == 0
            if int(flag[0].item()) == 0:
                exchange, exchange_note = "torch", "library exchange failed its preflight on some rank" + (f" (here: {why})" if why else "")
                print(f"[rank {rank}] {exchange_note}", file=sys.stderr)
        if exchange == "torch":
            use_rccl_for_torch_exchange()

        def join_across_ranks(pc):
            return multigpu.join_across_ranks(pc, group=torch_group[0], exchange=exchange) if exchange == "torch" else multigpu.join_across_ranks(pc, exchange=exchange)
    # the library's exchange pipelines inside the library (cwipc_hip_comm_submit: a thread of the communicator does the waiting,
    # no Python thread, no interpreter lock on the per-frame path); the torch exchange needs the worker thread below
    lib_comm = multigpu.library_comm() if (joining and pipelined and exchange == "library") else None
    last_fused = [None]
    if pipelined and lib_comm is None:
        import queue
        import threading

        FLUSH = object()
        # CWIPC_BENCH_JOIN_ASYNC=0: the worker waits for every frame's collective before it takes the next frame;
        # default: the collective of frame i is on the wire while frame i + 1 is packed (multigpu.JoinPipeline)
        join_async = os.environ.get("CWIPC_BENCH_JOIN_ASYNC", "1") != "0" and exchange == "torch"   # (the library's call never waits for the payload)

        class Joiner:
            def __init__(self):
                self.todo = queue.Queue(maxsize=2)
                self.last = None
                self.error = None
                self.thread = threading.Thread(target=self.run, daemon=True)
                self.thread.start()

            def run(self):
                torch.cuda.set_device(device_index)
                pipe = JoinPipeline(torch_group[0]) if join_async else None
                self.idle = self.busy = 0.0
                while True:
                    t_wait = time.perf_counter()
                    item = self.todo.get()
                    t_got = time.perf_counter()
                    self.idle += t_got - t_wait
                    try:
                        if item is None:
                            return
                        if self.error is not None:
                            continue
                        if item is FLUSH:
                            fused = pipe.flush() if pipe is not None else None
                        elif pipe is not None:
                            fused = pipe.submit(item)      # the fused cloud of the frame before
                        else:
                            fused = join_across_ranks(item)
                        if fused is not None:
                            self.last = fused
                    except BaseException as e:   # keep draining: the main thread must not block on a full queue
                        self.error = e
                    finally:
                        self.busy += time.perf_counter() - t_got
                        self.todo.task_done()

            def drain(self):
                self.todo.put(FLUSH)
                self.todo.join()
                if self.error is not None:
                    raise self.error
                return self.last

        joiner = Joiner()

    def step(i: int):
        out = cwipc.cwipc_downsample(clouds[i % NCOPIES], CELLSIZE)
        if lib_comm is not None:
            last_fused[0] = lib_comm.submit(out)   # returns at once; frames are exchanged in this order
            return out
        if joiner is not None:
            t_put = time.perf_counter()
            joiner.todo.put(out)
            joiner.blocked = getattr(joiner, "blocked", 0.0) + time.perf_counter() - t_put
            return out
        if joining:
            out = join_across_ranks(out)
        return out

    def fence():
        if lib_comm is not None and last_fused[0] is not None:
            last_fused[0].count()   # settles: this frame's exchange, and every earlier one's, has been issued
        if joiner is not None:
            joiner.drain()
        cwipc.util.cwipc_util_dll_load().cwipc_hip_synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # Setup, whatever W is (r4): a thread whose downsample calls come back to back takes a second and a third voxel workspace (0.3 GB
    # of leaf grids each, ~10 ms of allocation), and it takes them when a call finds the workspace whose turn it is still at work --
    # somewhere in the first ten or twenty calls of a stream.  With a short warm-up that was inside the timed region (W = 10, K = 50:
    # 177 us per step instead of 47).  A burst of calls nobody waits for, before the warm-up steps: allocation is setup, not a step.
    for burst in range(3):
        held = [cwipc.cwipc_downsample(clouds[i % NCOPIES], CELLSIZE) for i in range(12)]
        held[-1].count()
        del held
    for i in range(args.warmup):
        step(i)
    # output points of this rank's own tile; four calls, so that the library's per-thread workspaces know the size of their results
    # whatever W is
    for _ in range(4):
        n_out = cwipc.cwipc_downsample(clouds[0], CELLSIZE).count()

    # ---- timed region ----
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(i)
    fence()
    elapsed = time.perf_counter() - t0
    if joiner is not None and os.environ.get("CWIPC_BENCH_JOIN_STATS") == "1":
        print(f"[rank {rank}] since start: main thread blocked on the join queue {joiner.blocked * 1e3:.2f} ms, "
              f"worker busy {joiner.busy * 1e3:.2f} ms, idle {joiner.idle * 1e3:.2f} ms; timed region {elapsed * 1e3:.2f} ms", file=sys.stderr)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    fused_points = (last_fused[0] if lib_comm is not None else joiner.drain() if joiner is not None else out).count()

    # ---- second pass over the same steps with per-kernel hipEvent timing ----
    with cwipc.cwipc_hip_profile() as prof:
        for i in range(args.steps):
            cwipc.cwipc_downsample(clouds[i % NCOPIES], CELLSIZE)
    kernels = {k: {"ms_total": v[0], "launches": v[1], "ms_avg": v[0] / max(v[1], 1)} for k, v in prof.kernels.items()}
    dominant = max(kernels, key=lambda k: kernels[k]["ms_total"])
    dom_ms = kernels[dominant]["ms_avg"]
    all_ms = sum(v["ms_total"] for v in kernels.values()) / args.steps
    algorithmic_bytes = 16 * n + 16 * n_out
    achieved = algorithmic_bytes / (dom_ms * 1e-3) / 1e9

    traffic, traffic_src = measured_traffic(n, dominant)

    # ---- sub-records: BASELINE configs[3] (8-tile capture, strong scaling over ranks) and configs[2] (outlier removal) ----
    config4 = None
    if not args.no_config4:
        if joiner is not None:
            joiner.drain()
        # (N > 1 with the library's exchange: cwipc_hip_comm_submit, so that the join of frame i overlaps the tiles of frame i + 1;
        # the last frame's fused cloud is settled -- every exchange has then been issued -- before the clock stops)
        c4_join = (lib_comm.submit if lib_comm is not None else join_across_ranks) if joining else None
        c4_elapsed, c4_ntile, c4_fused, c4_mine, c4_threads = bench_config4(cwipc, rank, world, args.config4_steps, 3, fence, c4_join)
        if dist is not None:
            t = torch.tensor([c4_elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            c4_elapsed = float(t.item())
        config4 = {
            "workload": "8 x cwipc_synthetic(2000000, angle = i pi/4), mask 1 << i, rotated i x 45 deg about Y; tile t on rank t mod N: "
                        "cwipc_tilefilter(1 << t) -> cwipc_downsample(0.01), n-ary join"
                        + (f", all-gatherv join over {world} ranks" if joining else "") + " [BASELINE configs[3]]",
            "value": 8 * c4_ntile * args.config4_steps / c4_elapsed / 1e6 if c4_ntile else None,
            "unit": "Mpoints/s", "scaling": "strong", "n_gpus": world, "steps": args.config4_steps,
            "ms_per_frame": c4_elapsed / args.config4_steps * 1e3, "points_per_tile": c4_ntile, "tiles_on_rank0": c4_mine,
            "fused_points": c4_fused, "inputs_resident_in_hbm": True, "tile_threads_per_rank": c4_threads,
        }
    config5 = None
    if not args.no_config5:
        c5_join = (lib_comm.submit if lib_comm is not None else join_across_ranks) if joining else None
        c5 = bench_config5(cwipc, rank, world, args.config5_frames, 10, fence, c5_join)
        c5_elapsed = c5["elapsed"]
        if dist is not None:
            t = torch.tensor([c5_elapsed, c5["p50_ms"], c5["p99_ms"]], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            c5_elapsed, c5["p50_ms"], c5["p99_ms"] = float(t[0].item()), float(t[1].item()), float(t[2].item())
        ms_frame = c5_elapsed / args.config5_frames * 1e3
        config5 = {
            "workload": "per frame 8 x cwipc_synthetic(300000, angle = i pi/4), mask 1 << i, rotated i x 45 deg about Y; tile t on rank t mod N: "
                        "colorize(0.8, camera) -> cwipc_downsample(0.01) -> cwipc_remove_outliers(16, 1.0, false), n-ary join"
                        + (f", all-gatherv join over {world} ranks" if joining else "") + " [BASELINE configs[4]]",
            "value": args.config5_frames / c5_elapsed, "unit": "frames/s", "target_frames_per_s": 30.0, "scaling": "strong", "n_gpus": world,
            "frames": args.config5_frames, "ms_per_frame": ms_frame, "p50_ms": c5["p50_ms"], "p99_ms": c5["p99_ms"],
            "points_per_tile": c5["n_tile"], "tiles_on_rank0": c5["tiles_here"], "fused_points": c5["fused_points"],
            "inputs_resident_in_hbm": True, "tile_threads_per_rank": c5["threads"],
            "frames_overlapped": False,
            "host_arrays_in_and_out_ms_per_frame": c5["host_io_ms_per_frame"],
            "page_locked_arrays_in_and_out_ms_per_frame": c5["host_pinned_io_ms_per_frame"],
            "page_locked_upload_GBps": c5["pinned_upload_gbps"],
            "h2d_d2h_share_of_that": (1.0 - ms_frame / c5["host_io_ms_per_frame"]) if c5["host_io_ms_per_frame"] else None,
        }
    config3 = None
    if world == 1 and not args.no_config3 and args.npoints == NPOINTS_ARG:
        config3 = bench_config3(cwipc, clouds[0], n)

    if rank == 0:
        result = {
            "metric": "Mpoints/s filtered (voxel downsample, 10M-pt synthetic) + achieved HBM GB/s",
            "value": world * n * args.steps / elapsed / 1e6,
            "unit": "Mpoints/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 coordinates in, i64 fixed-point sums, u8 colours",
            "data": "synthetic",
            "config": {
                "workload": f"cwipc_synthetic({args.npoints}) -> cwipc_downsample(+{CELLSIZE}) [BASELINE configs[1]]"
                            + (f" per rank, tile masks 1<<rank, + all-gatherv join over {world} ranks"
                               + (" (join of frame i overlaps the downsample of frame i+1)" if pipelined else "") if joining else ""),
                "points_per_gpu": n,
                "bytes_per_gpu": 16 * n,
                "outputs_per_gpu": n_out,
                "fused_points": fused_points,
                "input_copies_rotated": NCOPIES,
                "inputs_resident_in_hbm": True,
                **({"voxel_kernel_spare_cus": int(os.environ.get("CWIPC_SPARE_CUS", "0")), "exchange": exchange} if joining else {}),
                **({"exchange_fallback": True, "exchange_note": exchange_note} if exchange_note else {}),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": dominant,
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": algorithmic_bytes,
                "kernel_ms_avg": dom_ms,
                "all_kernels_ms_per_step": all_ms,
                # in a stream of calls consecutive accumulate kernels overlap (each leaves one CU per XCD free for the next):
                # what the chip sustains per call is the algorithmic traffic over the wall time of a step
                "achieved_pipelined": algorithmic_bytes / (elapsed / args.steps) / 1e9,
                "frac_pipelined": algorithmic_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBPS,
                "achieved_all_kernels": algorithmic_bytes / (all_ms * 1e-3) / 1e9,
                "frac_all_kernels": algorithmic_bytes / (all_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "note": "achieved / frac / kernel_ms_avg: hipEvents around the dominant kernel in a second pass over the same steps, every call "
                        "waited for, so the kernel runs ALONE (the events add ~3 us to what a kernel trace gives it).  In the timed region a "
                        "thread's calls rotate over three workspaces and streams and up to three accumulate kernels are in flight at once: a "
                        "call completes every ms_per_step (achieved_pipelined), while a kernel's own duration in a trace of that region is "
                        "longer than alone -- profiles/rNN_bench_rocprofv3_summary.txt gives both parts and the run with one workspace",
            },
            "kernels": kernels,
        }
        if config4 is not None:
            result["config4"] = config4
        if config3 is not None:
            result["config3"] = config3
        if config5 is not None:
            result["config5"] = config5
        if world == 1:
            # a single call with its result settled, both signs of the cell size (the stream figure above never settles one)
            result["call_then_count_us"] = {"+0.01": call_then_count(cwipc, clouds, CELLSIZE), "-0.01": call_then_count(cwipc, clouds, -CELLSIZE),
                                            "note": "one cwipc_downsample followed by count(), median of 40 calls on rotating inputs, wall"}
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(pts, pc_cellsize)
            result["parity"] = parity_record(pts, pc_cellsize, cwipc.cwipc_downsample(clouds[0], CELLSIZE).get_numpy_array())
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(result), flush=True)   # (flushed here: nothing that happens at interpreter exit may cost the line)
        os.dup2(2, 1)

    if joining and abandon_at_exit:
        # a thread of some rank is still inside a collective that never completed: a GPU-side hang.  The line is out (with
        # `exchange_fallback`); the status says that this run is NOT a clean record of the library's exchange.
        print(f"[rank {rank}] ending with status 3: the preflight of the library's exchange hung on some rank", file=sys.stderr)
        sys.stderr.flush()
        os._exit(3)
    if joiner is not None:
        joiner.todo.put(None)
        joiner.thread.join()
    if dist is not None:
        dist.barrier()
        if joining:
            multigpu.free_library_comms()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
