"""rtx_trace_rays_noalloc (include/rtx.h): the entry point that allocates nothing — device pointers in and out, its
intermediate buffers in a fixed file-scope device array — against rtx_trace_rays, which allocates per call, on the same
rays: ids, hit points and normals bit for bit, on two calls in a row (the second smaller, so that stale contents of the
shared array would show), in all traversal modes, plus the size limit. Collected last on purpose (file name)."""
import ctypes as C

import numpy as np
import pytest

from raytracer_amd import _capi
from raytracer_amd.renderer import HIPShader

pytestmark = pytest.mark.gpu
MAX_RAYS = 1 << 19
_CALLS = 0


def _rays(n, seed):
    rng = np.random.default_rng(seed)
    o = np.zeros((n, 4), np.float32); d = np.zeros((n, 4), np.float32)
    o[:, :3] = rng.uniform([-400, 0, -600], [1400, 1200, 700], size=(n, 3))
    v = rng.normal(size=(n, 3)); d[:, :3] = v / np.linalg.norm(v, axis=1, keepdims=True)
    d[: n // 50, :3] = 0; d[: n // 50, 2] = 1            # axis-parallel rays: the side list is part of the scratch
    d[:, 3] = 7.0                                         # the 4th lane is "ignored" by contract
    return o, d


def _noalloc(r, o, d):
    import torch
    L = _capi.hip_lib()
    n = len(o)
    to, td = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
    ids = torch.full((n,), -77, dtype=torch.int32, device="cuda")
    pts = torch.full((n, 4), -1.0, dtype=torch.float32, device="cuda")
    nrm = torch.full((n, 4), -1.0, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]                  # hipMemGetInfo: free device memory
    rc = L.rtx_trace_rays_noalloc(r.handle, to.data_ptr(), td.data_ptr(), n, ids.data_ptr(), pts.data_ptr(), nrm.data_ptr())
    free1 = torch.cuda.mem_get_info()[0]
    assert rc == 0, L.rtx_last_error().decode()
    # the call allocates nothing on the device (the very first call of a process may still see the runtime's own lazy set-up)
    global _CALLS
    _CALLS += 1
    assert free1 == free0 or _CALLS == 1, (free0, free1)
    return ids.cpu().numpy(), pts.cpu().numpy(), nrm.cpu().numpy()


@pytest.mark.parametrize("traversal", [_capi.TRAVERSE_WIDE, _capi.TRAVERSE_CULLED, _capi.TRAVERSE_EXACT])
def test_noalloc_entry_point_equals_the_allocating_one(synth_small, traversal):
    r = HIPShader(synth_small, traversal=traversal)
    for n, seed in ((300_000, 1), (40_001, 2)):           # two calls in a row, the second with a smaller input
        o, d = _rays(n, seed)
        want_ids, want_pts, want_nrm = r.trace_rays(o, d)
        c0 = r.counters()
        ids, pts, nrm = _noalloc(r, o, d)
        c1 = r.counters()
        assert c1.frames == c0.frames and list(c1.hitsPerBounce) == list(c0.hitsPerBounce)      # a probe, not a frame
        assert (want_ids >= 0).sum() > n // 20
        assert (ids == want_ids).all()
        assert (pts.view(np.uint32) == want_pts.view(np.uint32)).all()
        assert (nrm.view(np.uint32) == want_nrm.view(np.uint32)).all()
    # the frame pipeline still works on the same renderer afterwards (the call borrowed its launch state)
    px = np.zeros((r.height, r.width, 4), np.float32)
    r.Render(px)
    assert np.isfinite(px).all() and px.any()
    r.close()


def test_noalloc_rejects_more_rays_than_its_scratch_holds(scene1):
    import torch
    L = _capi.hip_lib()
    r = HIPShader(scene1)
    n = MAX_RAYS + 1
    buf = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    rc = L.rtx_trace_rays_noalloc(r.handle, buf.data_ptr(), buf.data_ptr(), n, None, None, None)
    assert rc != 0 and b"exceed" in L.rtx_last_error()
    # exactly the maximum is accepted, with outputs left out
    o, d = _rays(MAX_RAYS, 3)
    to, td = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
    ids = torch.zeros(MAX_RAYS, dtype=torch.int32, device="cuda")
    assert L.rtx_trace_rays_noalloc(r.handle, to.data_ptr(), td.data_ptr(), MAX_RAYS, ids.data_ptr(), None, None) == 0
    want_ids, _, _ = r.trace_rays(o, d)
    assert (ids.cpu().numpy() == want_ids).all()
    r.close()
