"""Randomised parity: small synthetic scenes, random cameras, rays built to sit on the edges of the culled
path's assumptions (signed zeros, denormal and huge reciprocals, origins on box planes). Bars as in
test_gpu_parity.py: ids bit-exact, hit points and normals bit-exact, framebuffer bit-identical (conftest.check_fb)."""
import numpy as np
import pytest

import oracle
from conftest import check_fb
from raytracer_amd import _capi
from raytracer_amd.renderer import HIPShader
from raytracer_amd.scene import orbit, synthetic_scene

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(6))
def test_random_scene_random_camera(libs, seed):
    rng = np.random.default_rng(100 + seed)
    ntri = int(rng.integers(300, 30000))
    s = synthetic_scene(ntri, 160, 96, seed=1000 + seed, tex_size=32, nthreads=2)
    d, cam = s.desc(), s.camera
    orbit(cam, float(rng.uniform(0, 360)), float(rng.uniform(-30, 30)), radius=float(rng.uniform(1200, 4000)))
    r = HIPShader(s)
    for fc in (0, 5):
        assert (r.primary_hits(fc, cam) == oracle.primary_hits(d, cam, fc)).all()
    px = np.zeros((d.height, d.width, 4), np.float32); want = np.zeros_like(px)
    for f in range(3):
        r.frame_counter = f
        r.Render(None, cam)                      # queued: frames overlap on the device
        oracle.render(d, cam, f, want, 4)
    px = r.download()
    # the reference's shading yields NaN for some grazing / back-side configurations (they then stay in the
    # running mean): the same pixels must be NaN here, and every other pixel within the tolerance
    check_fb(px, want, nan_ok=True)
    r.close()


def test_edge_case_rays(synth_small):
    s = synth_small
    d = s.desc()
    boxes = np.array(s.boxes, copy=False)
    rng = np.random.default_rng(77)
    n = 60_000
    o = np.zeros((n, 4), np.float32); dirs = np.zeros((n, 4), np.float32)
    o[:, :3] = rng.uniform([-400, 0, -600], [1400, 1200, 700], size=(n, 3))
    v = rng.normal(size=(n, 3)); dirs[:, :3] = v / np.linalg.norm(v, axis=1, keepdims=True)
    k = 5000
    dirs[0 * k:1 * k, 0] = -0.0                                   # signed zero component
    dirs[1 * k:2 * k, 1] = np.float32(1e-42)                      # denormal: reciprocal overflows to inf
    dirs[2 * k:3 * k, 2] = np.float32(-3e-39)                     # denormal, negative
    dirs[3 * k:4 * k, 0] = np.float32(1e-30)                      # tiny but normal: reciprocal 1e30, products overflow
    dirs[4 * k:5 * k, 1] = np.float32(-1e-30)
    # origins exactly on box planes (0 * inv and inv * 0 cases), directions axis-parallel for half of them
    leaf = boxes[boxes["objectID"] != -1]
    pick = leaf[rng.integers(0, len(leaf), size=2 * k)]
    o[5 * k:7 * k, 0] = pick["min"][:, 0]
    o[5 * k:7 * k, 1] = pick["max"][:, 1]
    dirs[5 * k:6 * k, :3] = (0.0, 0.0, 1.0)
    # origins inside the geometry's bounds looking along each axis
    dirs[7 * k:8 * k, :3] = (1.0, 0.0, 0.0)
    dirs[8 * k:9 * k, :3] = (0.0, -1.0, 0.0)
    want_ids, want_pts, want_nrm = oracle.trace_rays(d, o, dirs)
    for trav in (_capi.TRAVERSE_CULLED, _capi.TRAVERSE_EXACT, _capi.TRAVERSE_WIDE):
        r = HIPShader(s, traversal=trav)
        ids, pts, nrm = r.trace_rays(o, dirs)
        assert (ids == want_ids).all()
        hit = ids >= 0
        assert hit.sum() > 1000
        assert (pts[hit, :3].view(np.uint32) == want_pts[hit, :3].view(np.uint32)).all()
        assert (nrm[hit, :3].view(np.uint32) == want_nrm[hit, :3].view(np.uint32)).all()
        r.close()


def test_sliver_scene_is_not_culled_by_distance(libs):
    """A triangle's computed t can undercut its own box's computed entry distance when the ray grazes it, by an amount
    that grows with the triangle's aspect ratio; distance culling would then drop a hit the reference takes. This
    height field (38 845 triangles on a 256x128 image: 900-unit slivers) made 48 of 24 M bounce rays differ with the
    fixed 2^-8 slack of earlier builds. rtx_create now measures the triangles' shapes and does not cull such a scene
    at all: culled and exact traversal must agree on every ray."""
    s = synthetic_scene(38845, 256, 128, seed=7038, tex_size=32, nthreads=4)
    d, cam = s.desc(), s.camera
    rng = np.random.default_rng(5038)
    o0, d0 = oracle.primary_rays(d, cam, 0)
    rc = HIPShader(s, traversal=_capi.TRAVERSE_CULLED)
    assert np.isinf(rc.info().cullSlack) and rc.info().sliverFraction > 1e-3 and rc.info().traversal == _capi.TRAVERSE_WIDE
    re = HIPShader(s, traversal=_capi.TRAVERSE_EXACT)
    ids, pts, nrm = re.trace_rays(o0, d0)
    hit = ids >= 0
    per = 3_000_000 // max(1, int(hit.sum()))
    p = np.repeat(pts[hit], per, axis=0); n = np.repeat(nrm[hit], per, axis=0)
    v = rng.normal(size=(len(p), 3)).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    dd = np.zeros_like(p); dd[:, :3] = v + n[:, :3]
    dd[:, :3] /= np.linalg.norm(dd[:, :3], axis=1, keepdims=True) + 1e-20
    p[:, 3] = 0
    ic, pc, nc = rc.trace_rays(p, dd)
    ie, pe, ne = re.trace_rays(p, dd)
    assert (ic == ie).all() and (pc.view(np.uint32) == pe.view(np.uint32)).all()
    rc.close(); re.close()
