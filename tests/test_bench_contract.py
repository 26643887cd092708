"""bench.py's pieces that need no GPU: the committed PMC traffic record is the one the bench line will quote, and the argument
defaults are the contract's (N = 1, a K and W that finish within minutes)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_traffic_record_matches_the_headline_workload():
    sys.path.insert(0, ROOT)
    import bench
    traffic, source = bench.measured_traffic(9998244, "voxel_accumulate")
    assert source is not None and source.startswith("r"), "no committed profiles/rNN_traffic.json for the headline workload and kernel"
    # algorithmic bytes of a launch: 16 B per input point + 16 B per output point (39 548 at +0.01); measured traffic stays near it
    algorithmic = 16 * 9998244 + 16 * 39548
    assert 0.9 * algorithmic < traffic < 1.5 * algorithmic, (traffic, algorithmic)
    newest = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json"))[-1]
    assert source == newest
    record = json.load(open(os.path.join(ROOT, "profiles", newest)))
    assert record["kernel"] == "voxel_accumulate" and record["hbm_bytes_per_launch"] == traffic


def test_defaults_are_the_contracts():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for needle in ('"--gpus", type=int, default=1', '"--steps", type=int, default=200', '"--warmup", type=int, default=20'):
        assert needle in src, needle
    assert 'HBM_PEAK_GBPS = 8000.0' in src


def test_committed_profile_files_are_from_one_run():
    """The newest round's kernel-stats csv is the one its summary text quotes (the two disagreed): the
    accumulate kernel's AverageNs in the csv equals the figure printed in the summary, and the summary's bench line is there."""
    import csv
    import re
    prof = os.path.join(ROOT, "profiles")
    rounds = sorted({f.split("_")[0] for f in os.listdir(prof) if re.match(r"r\d\d_bench_rocprofv3_summary\.txt", f)})
    newest = rounds[-1]
    if int(newest[1:]) < 3:
        import pytest
        pytest.skip("round 2's files predate the single-run script")
    rows = list(csv.DictReader(open(os.path.join(prof, newest + "_bench_kernel_stats.csv"))))
    # (the headline's accumulate kernel is the octree variant, template argument MODE = 1; since round 4 the run also holds the plain-grid
    # variant <0, ...> of the call-then-count figure for a negative cell size)
    k1 = [r for r in rows if "voxel_accumulate" in r["Name"] and "general" not in r["Name"] and "kernel<0" not in r["Name"]]
    assert len(k1) == 1, [r["Name"] for r in k1]
    avg_csv = float(k1[0]["AverageNs"])
    summary = open(os.path.join(prof, newest + "_bench_rocprofv3_summary.txt")).read()
    # (the kernel's name as the csv has it -- "void cwipc_amd::(anonymous namespace)::voxel_accumulate_fast_kernel<1>..." -- holds blanks)
    m = re.search(r"^[^\n]*voxel_accumulate(?!_general)(?!_fast_kernel<0)[^\n]*?\s+(\d+)\s+(\d+)\s+(\d+)\s+\d+\s+\d+\s+[\d.]+\s*$", summary, re.M)
    assert m, "no accumulate-kernel row in the summary"
    assert int(m.group(1)) == int(k1[0]["Calls"]) and abs(int(m.group(3)) - avg_csv) <= 1.0, (m.groups(), avg_csv)
    assert "# bench line under the profiler:" in summary
    assert os.path.exists(os.path.join(prof, newest + "_bench_line.json")) and os.path.exists(os.path.join(prof, newest + "_traffic.json"))
