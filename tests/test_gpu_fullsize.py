"""Full-size checks (BASELINE.json configs[3] shape: 10 M triangles, 3840x2176)
through size-independent properties plus an oracle comparison on a band of rows."""
import numpy as np
import pytest

import oracle
from conftest import psnr
from raytracer_amd import _capi
from raytracer_amd.renderer import HIPShader

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big_scene(libs):
    from raytracer_amd.scene import synthetic_scene
    return synthetic_scene(10_000_000, 3840, 2160, seed=12345, textured=True, nthreads=16)


def test_culled_equals_exact_on_every_pixel_and_matches_oracle_band(big_scene):
    s = big_scene
    d, cam = s.desc(), s.camera
    assert (d.width, d.height) == (3840, 2176)
    culled = HIPShader(s, traversal=_capi.TRAVERSE_CULLED)
    ids_c = culled.primary_hits(0)
    fb_c = np.zeros((d.height, d.width, 4), np.float32)
    for f in range(2):
        culled.frame_counter = f
        culled.Render(fb_c)
    cc = culled.counters()
    culled.close()
    exact = HIPShader(s, traversal=_capi.TRAVERSE_EXACT)
    ids_e = exact.primary_hits(0)
    fb_e = np.zeros_like(fb_c)
    for f in range(2):
        exact.frame_counter = f
        exact.Render(fb_e)
    ce = exact.counters()
    exact.close()
    # distance culling must never change a closest hit: ids and framebuffers identical, same ray counts
    assert (ids_c == ids_e).all()
    assert (fb_c.view(np.uint32) == fb_e.view(np.uint32)).all()
    assert list(cc.raysPerBounce) == list(ce.raysPerBounce) and list(cc.hitsPerBounce) == list(ce.hitsPerBounce)
    assert cc.raysPerBounce[0] == 2 * d.width * d.height
    # oracle on a band of rows (the whole frame would take minutes on the CPU)
    y0, y1 = 1000, 1048
    want = np.zeros_like(fb_c)
    for f in range(2):
        oracle.render(d, cam, f, want, 16, rows=(y0, y1))
    assert float(np.abs(fb_c[y0:y1] - want[y0:y1]).max()) <= 2e-3
    assert psnr(fb_c[y0:y1], want[y0:y1]) >= 50.0
    ids_o = oracle.primary_hits(d, cam, 0, nthreads=16)
    assert (ids_c == ids_o).all()
