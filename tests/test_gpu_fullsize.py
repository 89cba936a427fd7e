"""Full-size checks (BASELINE.json configs[3] shape: 10 M triangles, 3840x2176)
through size-independent properties plus an oracle comparison on a band of rows."""
import numpy as np
import pytest

import oracle
from conftest import check_fb
from raytracer_amd import _capi
from raytracer_amd.renderer import HIPShader

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big_scene(libs):
    from raytracer_amd.scene import synthetic_scene
    return synthetic_scene(10_000_000, 3840, 2160, seed=12345, textured=True, nthreads=16)


def test_wide_culled_and_exact_agree_on_every_pixel_and_match_the_oracle(big_scene):
    """cfg4 in the three traversal modes — RTX_TRAVERSE_WIDE first: it is the library default and the mode bench.py
    times on this very configuration. Primary ids on all 8.36 M pixels against the oracle; the three framebuffers (two
    accumulated frames) identical in bits, and a band of rows identical in bits to the oracle's; same ray and hit counts."""
    s = big_scene
    d, cam = s.desc(), s.camera
    assert (d.width, d.height) == (3840, 2176)
    ids_o = oracle.primary_hits(d, cam, 0, nthreads=16)
    out = {}
    for name, trav in (("wide", _capi.TRAVERSE_WIDE), ("culled", _capi.TRAVERSE_CULLED), ("exact", _capi.TRAVERSE_EXACT)):
        r = HIPShader(s, traversal=trav)
        assert r.info().traversal == trav
        ids = r.primary_hits(0)
        assert (ids == ids_o).all(), name
        del ids
        fb = np.zeros((d.height, d.width, 4), np.float32)
        for f in range(2):
            r.frame_counter = f
            r.Render(fb)
        c = r.counters()
        r.close()
        out[name] = (fb, list(c.raysPerBounce), list(c.hitsPerBounce))
    # no walk may ever change a closest hit: framebuffers identical, same ray counts
    for name in ("culled", "exact"):
        assert (out[name][0].view(np.uint32) == out["wide"][0].view(np.uint32)).all(), name
        assert out[name][1:] == out["wide"][1:], name
    assert out["wide"][1][0] == 2 * d.width * d.height
    # oracle on a band of rows (the whole frame would take minutes on the CPU)
    y0, y1 = 1000, 1048
    want = np.zeros_like(out["wide"][0])
    for f in range(2):
        oracle.render(d, cam, f, want, 16, rows=(y0, y1))
    check_fb(out["wide"][0][y0:y1], want[y0:y1])
