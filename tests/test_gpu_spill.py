"""The HBM spill columns of the traversal stacks, forced into use on a small tree by a build of the library with 2
instead of 16 node entries per lane in LDS (raytracer_amd/librtx_hip_qn2.so, made by `make all`), with an axis-aligned
camera whose dx == 0 pixel column walks the reference topology through the same buffer (tests/spill_check.py)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_spill_columns_of_clean_and_axis_parallel_lanes_never_overlap():
    lib = os.path.join(ROOT, "raytracer_amd", "librtx_hip_qn2.so")
    if not os.path.exists(lib):
        pytest.fail("raytracer_amd/librtx_hip_qn2.so is not built (make -C raytracer_amd/csrc all)")
    env = dict(os.environ, RTX_HIP_LIB="librtx_hip_qn2.so")
    out = subprocess.run([sys.executable, "-u", "-X", "faulthandler", os.path.join(ROOT, "tests", "spill_check.py")], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "spill check OK" in out.stdout
