"""Quick GPU sanity run: HIP backend vs the oracle on a small synthetic scene."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd.scene import synthetic_scene
from raytracer_amd.renderer import HIPShader
from raytracer_amd import _capi
import oracle

def psnr(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 999.0 if mse == 0 else 10 * np.log10(1.0 / mse)

ntri = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
w = int(sys.argv[2]) if len(sys.argv) > 2 else 256
h = int(sys.argv[3]) if len(sys.argv) > 3 else 256
s = synthetic_scene(ntri, w, h, tex_size=64)
d = s.desc(); cam = s.camera
print("scene", d.width, d.height, d.numObjects, d.numBoxes, flush=True)
for trav in (_capi.TRAVERSE_CULLED, _capi.TRAVERSE_EXACT):
    r = HIPShader(s, traversal=trav, count_stats=True)
    t = time.time(); ids = r.primary_hits(0); print("gpu primary s", time.time() - t, flush=True)
    ref = oracle.primary_hits(d, cam, 0)
    print("trav", trav, "hits", int((ref >= 0).sum()), "mismatch", int((ids != ref).sum()), flush=True)
    px = np.zeros((d.height, d.width, 4), np.float32)
    opx = np.zeros_like(px)
    for f in range(3):
        r.frame_counter = f
        r.Render(px)
        oracle.render(d, cam, f, opx, 8)
        diff = np.abs(px - opx)
        print("frame", f, "maxabs", float(diff.max()), "psnr", psnr(px, opx), "exact_px", float((diff.max(axis=2) == 0).mean()), flush=True)
    c = r.counters()
    print("rays", c.rays, list(c.raysPerBounce), list(c.hitsPerBounce), "inner", list(c.innerVisits), "leaf", list(c.leafTests), "kms", [round(x,3) for x in c.kernelMs], "ms", c.lastFrameMs, c.lastTraceMs, c.lastShadeMs, flush=True)
    r.close()
