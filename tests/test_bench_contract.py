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
