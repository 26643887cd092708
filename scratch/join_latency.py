"""Host-side cost of one multi-GPU join (join_across_ranks) on the real device path, with a one-rank RCCL group
(everything but the wire): per-frame latency for a downsampled 10 M-point tile.  Not the bench contract."""
import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
import cwipc_util_amd as cw
from cwipc_util_amd.multigpu import join_across_ranks
from bench import make_input
torch.cuda.set_device(0); cw.cwipc_hip_set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
pc = cw.cwipc_downsample(make_input(cw, 10_000_000, 0.0), 0.01)
sync = cw.util.cwipc_util_dll_load().cwipc_hip_synchronize
for ex in ("library", "torch"):
    for _ in range(20): out = join_across_ranks(pc, exchange=ex)
    sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): out = join_across_ranks(pc, exchange=ex)
    sync(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
    print("join_across_ranks(exchange=%s), 1 rank, %d points: %.1f us per frame" % (ex, pc.count(), dt * 1e6), "fused", out.count())
# the library's exchange with this rank's own part sent through RCCL too (record all-gather + host read-back + send/recv group)
from cwipc_util_amd.multigpu import library_comm
comm = library_comm()
for _ in range(20): out = comm.join(pc, loopback=True)
sync(); t0 = time.perf_counter()
for _ in range(200): out = comm.join(pc, loopback=True)
t_host = (time.perf_counter() - t0) / 200
sync(); dt = (time.perf_counter() - t0) / 200
print("cwipc_hip_comm_join, loopback: %.1f us per frame on the host (%.1f us with the last frame's payload done)" % (t_host * 1e6, dt * 1e6), "fused", out.count())
from cwipc_util_amd.multigpu import JoinPipeline
pipe = JoinPipeline()
for _ in range(20): pipe.submit(pc)
sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): out2 = pipe.submit(pc)
sync(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
print("JoinPipeline.submit, 1 rank: %.1f us per frame" % (dt * 1e6), "fused", out2.count())
if os.environ.get("JOIN_PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): out2 = pipe.submit(pc)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(12)
if os.environ.get("JOIN_PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): out = join_across_ranks(pc, exchange="torch")
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
dist.destroy_process_group()
