"""Child process of tests/test_gpu_exchange_ranks.py: W ranks of the library's multi-GPU join as W threads of this process, all on
GPU 0, against tests/standin/lib/libcwipc_util.so (the library with an in-process stand-in for RCCL, CWIPC_LIBRARY_DIR).  Every
rank runs the same list of frames through cwipc_hip_comm_join or cwipc_hip_comm_submit and leaves what it got in <out>/rank<r>.npz.

usage: ranks_child.py <world> <mode: join|submit> <scenario> <out dir>
"""
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cwipc_util_amd as cw                                  # noqa: E402
from exchange_frames import frames_of                       # noqa: E402  (tests/exchange_frames.py: what every rank holds, frame by frame)


def main():
    world, mode, scenario, out_dir = int(sys.argv[1]), sys.argv[2], sys.argv[3], sys.argv[4]
    assert "standin" in cw.util.cwipc_util_dll_load()._name, "this process must load the stand-in build"
    uid = cw.cwipc_hip_comm_unique_id()
    frames = frames_of(scenario, world)
    errors = []
    start = threading.Barrier(world)

    def rank_main(rank):
        try:
            cw.cwipc_hip_set_device(0)
            comm = cw.cwipc_hip_comm(uid, rank, world)       # collective
            start.wait(timeout=60)
            held = []
            results = {}
            for f, per_rank in enumerate(frames):
                pts, ts, cs, has = per_rank[rank]
                pc = None
                if has:
                    pc = cw.cwipc_from_numpy_array(pts, ts)
                    pc._set_cellsize(cs)
                if mode == "submit":
                    held.append((f, comm.submit(pc)))
                    if pc is not None:
                        pc.free()                           # the input may go as soon as the call has returned
                else:
                    try:
                        held.append((f, comm.join(pc)))
                    except cw.CwipcError as e:              # a rank that is left without a result this frame
                        held.append((f, None))
            for f, fused in held:
                if fused is None:
                    results["f%d_failed" % f] = np.array([1])
                    continue
                try:
                    arr = fused.get_numpy_array()
                    results["f%d_points" % f] = arr
                    results["f%d_meta" % f] = np.array([fused.timestamp(), fused.cellsize()], dtype=np.float64)
                except cw.CwipcError:
                    results["f%d_failed" % f] = np.array([1])
            comm.free()
            np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **results)
        except Exception as e:                               # pragma: no cover
            import traceback
            errors.append((rank, traceback.format_exc()))

    threads = [threading.Thread(target=rank_main, args=(r,), name="rank%d" % r) for r in range(world)]
    for t in threads: t.start()
    for t in threads: t.join()
    if errors:
        for r, tb in errors:
            print("rank", r, "failed:\n", tb, file=sys.stderr)
        sys.exit(1)
    print("ranks done", world, mode, scenario)


if __name__ == "__main__":
    main()
