import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE = "/root/reference"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def psnr(a, b):
    mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
    return float("inf") if mse == 0.0 else 10.0 * np.log10(1.0 / mse)


def have_reference():
    return os.path.isdir(os.path.join(REFERENCE, "src"))


@pytest.fixture(scope="session")
def libs():
    """Builds (if needed) and returns nothing; makes the shared libraries available."""
    import subprocess
    if not os.path.exists(os.path.join(ROOT, "raytracer_amd", "librtx_host.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "raytracer_amd", "csrc"), "host", "-s"])
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    return True


@pytest.fixture(scope="session")
def scene1(libs):
    from raytracer_amd.scene import HostScene
    return HostScene.load_blob(os.path.join(GOLDEN, "scene1.blob"))


@pytest.fixture(scope="session")
def scene2(libs):
    from raytracer_amd.scene import HostScene
    return HostScene.load_blob(os.path.join(GOLDEN, "scene2.blob"))


@pytest.fixture(scope="session")
def synth_small(libs):
    from raytracer_amd.scene import synthetic_scene
    return synthetic_scene(50_000, 320, 200, seed=7, tex_size=64, nthreads=4)


def golden(name):
    return np.load(os.path.join(GOLDEN, name))
