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


MAX_ABS = 2e-3
MIN_PSNR = 50.0


def check_fb(got, want, nan_ok=False):
    """Framebuffer parity with the oracle. The CONTRACT (BASELINE.json north_star) is max-abs error <= 2e-3 and
    PSNR >= 50 dB at equal spp — the double-precision libm calls of the shading terms are not pinned bit for bit between
    glibc and the device libm. What the HIP path DELIVERS on every scene of this suite is the oracle's image bit for bit,
    and that is what is asserted: a lost bit fails the test with the number of pixels that differ. Setting
    RTX_TEST_TOLERANCE_ONLY=1 (a different libm, say) falls back to the contract, and still prints that number.
    nan_ok: the reference's shading yields NaN for some grazing / back-side configurations; the same pixels must be NaN."""
    nan = np.isnan(want)
    if not nan_ok:
        assert not np.isnan(got).any()
    assert (np.isnan(got) == nan).all()
    g, w = np.where(nan, np.float32(0), got), np.where(nan, np.float32(0), want)
    differing = int((g.view(np.uint32) != w.view(np.uint32)).any(axis=-1).sum())
    if differing == 0:
        return
    err = float(np.abs(g - w).max())
    msg = (f"bit-identity with the oracle lost: {differing} of {g.shape[0] * g.shape[1]} pixels differ in bits "
           f"(max abs {err:g}, PSNR {psnr(g, w):.1f} dB; contract {MAX_ABS:g} / {MIN_PSNR:g} dB)")
    if os.environ.get("RTX_TEST_TOLERANCE_ONLY") != "1":
        pytest.fail(msg)
    print("WARNING:", msg)
    assert err <= MAX_ABS and psnr(g, w) >= MIN_PSNR, msg


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
