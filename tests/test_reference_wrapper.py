"""Drop-in check of the boundary: the REFERENCE's own Python wrapper and the
reference's own test cases (python/test_cwipc_util.py, run where they lie under /root/reference, unmodified)
against this repository's libcwipc_util.so.  Only here: the reference does not travel to the GPU box, so the
cases that run a filter (they need a GPU) are not in the list, nor are those that need the reference's missing PLY fixture."""
import os
import subprocess
import sys

import pytest

REFERENCE = "/root/reference"
HERE = os.path.dirname(__file__)
LIBDIR = os.path.join(os.path.dirname(HERE), "cwipc_util_amd", "lib")

# every case of the reference's TestApi that touches neither a filter nor the missing fixture file nor the network
CASES = [
    "test_point", "test_pointarray", "test_pointarray_filled", "test_cwipc", "test_cwipc_source",
    "test_cwipc_from_points_empty", "test_cwipc_from_points", "test_cwipc_numpy_array", "test_cwipc_numpy_matrix",
    "test_cwipc_timestamp_cellsize", "test_cwipc_read_nonexistent", "test_cwipc_write_nonexistent", "test_cwipc_write", "test_cwipc_write_binary",
    "test_cwipc_write_debugdump", "test_cwipc_write_debugdump_nonexistent", "test_cwipc_packet", "test_cwipc_logger",
    "test_cwipc_synthetic", "test_cwipc_synthetic_available_false", "test_cwipc_synthetic_nonexistent_metadata",
    "test_cwipc_synthetic_metadata", "test_cwipc_synthetic_nonexistent_auxiliary_operation",
    "test_cwipc_synthetic_auxiliary_operation", "test_cwipc_synthetic_args", "test_cwipc_synthetic_tiled",
    "test_cwipc_synthetic_config", "test_cwipc_capturer_nonexistent", "test_metadata_empty",
]


@pytest.mark.skipif(not os.path.exists(os.path.join(REFERENCE, "python", "test_cwipc_util.py")), reason="reference tree not present")
def test_reference_test_cases_pass_against_this_library(tmp_path):
    cmd = [sys.executable, os.path.join(HERE, "helpers", "run_reference_tests.py"), REFERENCE, LIBDIR] + CASES
    proc = subprocess.run(cmd, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    tail = (proc.stdout + proc.stderr)[-4000:]
    assert proc.returncode == 0, tail
    assert f"RAN {len(CASES)} FAILED 0 ERRORS 0 SKIPPED 0" in proc.stdout, tail
