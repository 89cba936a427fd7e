"""Run by tests/test_gpu_spill.py in a process of its own with RTX_HIP_LIB=librtx_hip_qn2.so (2 node entries per lane
in LDS instead of 16): every traversal deeper than two pending node entries goes through the HBM spill columns. The
camera is axis-aligned, so a whole pixel column (dx == 0) and a whole pixel row (dy == 0) are rays that are not "clean"
and walk the reference topology with THEIR stack in the same spill buffer, at the same time as the clean lanes of other
waves spill theirs: the two layouts must never share a word (rtx_kernels.h closest_hit)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from raytracer_amd import _capi  # noqa: E402
from raytracer_amd.renderer import HIPShader  # noqa: E402
from raytracer_amd.scene import synthetic_scene  # noqa: E402


def note(*a):
    """progress on stderr, unbuffered: if the process dies, the last line says where"""
    print("[spill_check]", *a, file=sys.stderr, flush=True)


def main():
    note("start")
    assert os.environ.get("RTX_HIP_LIB", "").endswith("qn2.so") or os.environ.get("RTX_SPILL_ANY_LIB") == "1"
    s = synthetic_scene(200_000, 640, 416, seed=3, tex_size=64, nthreads=8)
    d, cam = s.desc(), s.camera
    note("scene built")
    assert cam.front.x == 0.0 and cam.front.y == 0.0 and cam.right.y == 0.0 and cam.up.x == 0.0      # axis-aligned
    want_ids = oracle.primary_hits(d, cam, 0, nthreads=8)
    want = np.zeros((d.height, d.width, 4), np.float32)
    for f in range(3):
        oracle.render(d, cam, f, want, 8)
    note("oracle frames done")
    o, dr = oracle.primary_rays(d, cam, 0)
    dirs = dr.reshape(d.height, d.width, 4)
    assert (dirs[..., 0] == 0.0).sum() > d.height // 2, "no axis-parallel primary rays"
    bad = 0
    for rep in range(3):                                  # the overlap is a race between waves: a few rounds
        for trav in (_capi.TRAVERSE_WIDE, _capi.TRAVERSE_EXACT):
            note("round", rep, "traversal", trav, "create")
            r = HIPShader(s, traversal=trav)
            note("created")
            if trav == _capi.TRAVERSE_WIDE:
                assert r.info().wideTreeDepth >= 4, r.info().wideTreeDepth
            ids = r.primary_hits(0)
            note("primary hits done")
            px = np.zeros_like(want)
            for f in range(3):
                r.frame_counter = f
                r.Render(px)
                note("frame", f, "done")
            r.close()
            note("closed")
            nid = int((ids != want_ids).sum())
            npx = int((px.view(np.uint32) != want.view(np.uint32)).any(axis=2).sum())
            if nid or npx:
                print(f"round {rep} traversal {trav}: {nid} ids differ, {npx} pixels differ")
                bad += 1
    if bad:
        sys.exit(1)
    print("spill check OK")


if __name__ == "__main__":
    main()
