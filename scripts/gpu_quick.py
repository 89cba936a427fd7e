"""Quick timing of the 4K 10M-triangle workload (no CPU baseline, no counting pass)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RTX_VERBOSE", "1")
import numpy as np
from raytracer_amd.scene import synthetic_scene
from raytracer_amd.renderer import HIPShader
ntri = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
W_, H_ = [int(v) for v in os.environ.get("RTX_QUICK_RES", "3840x2160").split("x")]
s = synthetic_scene(ntri, W_, H_, nthreads=16)
world = int(os.environ.get("RTX_QUICK_WORLD", "1"))
t = time.time(); r = HIPShader(s, rank=0, world_size=world, frames_in_flight=int(os.environ.get("RTX_QUICK_FIF", "0"))); print("create s", round(time.time() - t, 2), flush=True)
for f in range(2):
    r.frame_counter = f; r.Render(None)
r.reset_counters()
K = 12
t = time.perf_counter()
for f in range(K):
    r.frame_counter = 2 + f; r.Render(None)
c = r.counters()
print("wall ms/frame", round((time.perf_counter() - t) / K * 1e3, 3), "Mrays/s (wall)", round(c.rays / (time.perf_counter() - t) / 1e6, 1))
km = [round(x / K, 3) for x in c.kernelMs]
print("ms/frame", round(c.totalMs / K, 3), "Mrays/s", round(c.rays / (c.totalMs * 1e-3) / 1e6, 1), "kernels", km, flush=True)

if os.environ.get("RTX_QUICK_COUNTS"):
    rc = HIPShader(s, count_stats=True)
    rc.frame_counter = 2; rc.Render(None)
    cc = rc.counters()
    rays = list(cc.raysPerBounce)
    print("rays/bounce", rays, "hits/bounce", list(cc.hitsPerBounce))
    print("node steps/ray", [round(cc.innerVisits[b] / max(rays[b], 1), 2) for b in range(4)],
          "leaf tests/ray", [round(cc.leafTests[b] / max(rays[b], 1), 2) for b in range(4)],
          "lane utilisation", [round((cc.innerVisits[b] + cc.leafTests[b]) / max(64 * cc.wavePasses[b], 1), 3) for b in range(4)],
          "node-step share", [round(cc.innerVisits[b] / max(cc.innerVisits[b] + cc.leafTests[b], 1), 3) for b in range(4)], flush=True)
