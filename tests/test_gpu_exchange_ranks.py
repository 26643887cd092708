"""The library's multi-GPU join with more than one rank, on ONE GPU (round 4).

csrc/exchange.cpp -- the record all-gather, the second gather round, the send / receive group at prefix-sum displacements, the
submit thread -- had only ever run with one rank: RCCL refuses two ranks on one device and the builder has one GPU.  Here the
ranks are threads of a child process that loads tests/standin/lib/libcwipc_util.so: the library's own objects linked against an
in-process stand-in for the nine RCCL entry points (tests/standin/rccl_standin.cpp: the wire is a device-to-device copy behind a
rendezvous per pair of ranks; a receive without its send, a send without its receive, a rank missing from a collective end the
call with an error after 20 s instead of passing).  Every rank's fused cloud of every frame is compared with the fold of
cwipc_join over the tiles in rank order (reference src/cwipc_filters.cpp:388-418, folded by
python/cwipc/net/source_synchronizer.py:175-188; tiles that are missing are simply not part of the frame, :163-171).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from exchange_frames import expected, frames_of

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STANDIN_DIR = os.path.join(ROOT, "tests", "standin", "lib")
CHILD = os.path.join(ROOT, "tests", "standin", "ranks_child.py")


def ensure_standin():
    """The stand-in build is made of the product's objects: one that is older than the product (or missing) is rebuilt here
    (tests/standin/build_standin.py; hipcc is on the GPU box too)."""
    lib = os.path.join(STANDIN_DIR, "libcwipc_util.so")
    product = os.path.join(ROOT, "cwipc_util_amd", "lib", "libcwipc_util.so")
    if os.path.exists(lib) and os.path.exists(product) and os.path.getmtime(lib) >= os.path.getmtime(product):
        return
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "standin", "build_standin.py")], capture_output=True, text=True, timeout=900)
    if proc.returncode != 0 or not os.path.exists(lib):
        pytest.fail("tests/standin/lib/libcwipc_util.so could not be built:\n" + proc.stdout[-1000:] + proc.stderr[-3000:])


def run_ranks(tmp_path, world, mode, scenario, faults=None, timeout=300):
    ensure_standin()
    env = dict(os.environ, CWIPC_LIBRARY_DIR=STANDIN_DIR, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("CWIPC_TEST_EXCHANGE_FAULTS", None)
    if faults:
        env["CWIPC_TEST_EXCHANGE_FAULTS"] = faults
    proc = subprocess.run([sys.executable, CHILD, str(world), mode, scenario, str(tmp_path)], env=env, timeout=timeout, capture_output=True, text=True)
    assert proc.returncode == 0, "ranks did not finish:\n" + proc.stdout[-2000:] + "\n" + proc.stderr[-4000:]
    return [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)], proc.stderr


@pytest.mark.parametrize("mode", ["join", "submit"])
@pytest.mark.parametrize("world", [2, 3, 8])
def test_stream_of_frames_on_several_ranks(tmp_path, world, mode):
    """Steady frames, regrowth (a second gather round in mid-stream), shrinking, ragged, empty and missing tiles, all points on
    one rank (the frame round 2's exchange deadlocked on), nothing anywhere: every rank gets the fold, frame after frame."""
    got, _ = run_ranks(tmp_path, world, mode, "stream")
    frames = frames_of("stream", world)
    exp = expected(frames, world)
    for r in range(world):
        for f, (pts, ts, cs, any_cloud) in enumerate(exp):
            assert "f%d_failed" % f not in got[r].files, (r, f)
            arr = got[r]["f%d_points" % f]
            assert len(arr) == len(pts) and arr.tobytes() == pts.tobytes(), (world, mode, r, f, len(arr), len(pts))
            if any_cloud:
                assert got[r]["f%d_meta" % f].tolist() == [float(ts), float(np.float32(cs))], (r, f)


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_in_trouble_are_left_out_by_everybody_alike(tmp_path, world):
    """The branches of join_frame that need a rank in trouble (test-build hook CWIPC_TEST_EXCHANGE_FAULTS, rank:frame:bits):
    frame 2: rank 1 has no device context (ST_ABSENT: it sends nothing, receives nothing, the others fuse a frame without its
    tile, its own call fails); frame 4: rank 0 holds no room and cannot allocate between the gathers (ST_NO_RECV: it still SENDS
    its tile, nobody sends to it, its own call fails); frame 6: the last rank holds no room before the gather and allocates in the
    second round while the others reuse theirs.  No rank is left waiting in any of them, and the frames after each are whole."""
    last = world - 1
    got, err = run_ranks(tmp_path, world, "join", "faults", faults="1:2:1,0:4:6,%d:6:4" % last)
    frames = frames_of("faults", world)
    exp = expected(frames, world, absent={(2, 1)})
    for r in range(world):
        for f, (pts, ts, cs, _any) in enumerate(exp):
            failed = "f%d_failed" % f in got[r].files
            if (f, r) in {(2, 1), (4, 0)}:
                assert failed, (r, f)
                continue
            assert not failed, (r, f, err[-2000:])
            arr = got[r]["f%d_points" % f]
            assert len(arr) == len(pts) and arr.tobytes() == pts.tobytes(), (world, r, f)
            assert got[r]["f%d_meta" % f].tolist() == [float(ts), float(np.float32(cs))], (r, f)


def test_the_stand_in_notices_an_unmatched_rank(tmp_path):
    """The checker checks: a rank that is struck out of a frame on ITS side only (a fault the others cannot see would be exactly
    round 2's bug) must not pass.  Here: world 2, but only one of the two ranks is started -- creation is collective, the rank
    gives up its rendezvous and the child ends with an error instead of hanging or succeeding."""
    ensure_standin()
    env = dict(os.environ, CWIPC_LIBRARY_DIR=STANDIN_DIR, HSA_ENABLE_IPC_MODE_LEGACY="0")
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import cwipc_util_amd as cw\n"
        "uid = cw.cwipc_hip_comm_unique_id()\n"
        "try:\n"
        "    cw.cwipc_hip_comm(uid, 0, 2)\n"
        "except cw.CwipcError as e:\n"
        "    print('refused:', e); sys.exit(7)\n"
        "sys.exit(0)\n"
    ) % ROOT
    proc = subprocess.run([sys.executable, "-c", code], env=env, timeout=120, capture_output=True, text=True)
    assert proc.returncode == 7, (proc.returncode, proc.stdout[-500:], proc.stderr[-1500:])
    assert "gave up" in proc.stderr
