"""Experiment: how much does binning bounce rays by direction octant speed up the trace launch?
Builds diffuse bounce rays from primary hit points on the host and times rtx_trace_rays' device part."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd.scene import synthetic_scene
from raytracer_amd.renderer import HIPShader

W, H = 3840, 2176
s = synthetic_scene(10_000_000, 3840, 2160, nthreads=16)
r = HIPShader(s)
cam = s.camera
v = lambda a: np.array([a.x, a.y, a.z], dtype=np.float64)
pos, front, up, right = v(cam.position), v(cam.front), v(cam.up), v(cam.right)
# pixels in 8x8-block order
by, bx, ly, lx = np.meshgrid(np.arange(H // 8), np.arange(W // 8), np.arange(8), np.arange(8), indexing="ij")
px = (bx * 8 + lx).ravel().astype(np.float64); py = (by * 8 + ly).ravel().astype(np.float64)
th = np.tan(np.radians(cam.fov) / 2)
nx = (2 * (px + 0.5) / W - 1) * th * cam.aspectRatio; ny = (1 - 2 * (py + 0.5) / H) * th
d = front[None, :] + nx[:, None] * right[None, :] + ny[:, None] * up[None, :]
d /= np.linalg.norm(d, axis=1, keepdims=True)
n = len(d)
O = np.zeros((n, 4), np.float32); O[:, :3] = pos
D = np.zeros((n, 4), np.float32); D[:, :3] = d
ids, pts, nrm = r.trace_rays(O, D)
print("primary: hits", int((ids >= 0).sum()), "of", n, "device ms", round(r.counters().lastTraceMs, 3), flush=True)
hit = ids >= 0
P = pts[hit, :3].astype(np.float64); N = nrm[hit, :3].astype(np.float64); Din = d[hit]
N *= np.where((N * Din).sum(1, keepdims=True) > 0, -1.0, 1.0)
rng = np.random.default_rng(1)
u = rng.normal(size=P.shape); u /= np.linalg.norm(u, axis=1, keepdims=True)
B = N + u; B /= np.maximum(np.linalg.norm(B, axis=1, keepdims=True), 1e-9)
m = len(P)
O2 = np.zeros((m, 4), np.float32); O2[:, :3] = P + 0.05 * N
D2 = np.zeros((m, 4), np.float32); D2[:, :3] = B
octant = (B[:, 0] < 0).astype(np.int64) | ((B[:, 1] < 0).astype(np.int64) << 1) | ((B[:, 2] < 0).astype(np.int64) << 2)
print("octant histogram", np.bincount(octant, minlength=8).tolist(), flush=True)

def run(tag, order):
    o, dd = (O2, D2) if order is None else (O2[order], D2[order])
    best = 1e9
    for _ in range(3):
        r.trace_rays(o, dd)
        best = min(best, r.counters().lastTraceMs)
    print(f"{tag:34s} {m} rays, device ms {best:.3f}  -> {m / best / 1e3:.1f} Mrays/s", flush=True)

run("bounce rays, queue order", None)
run("stable sort by octant (global)", np.argsort(octant, kind="stable"))
for g in (1 << 20, 1 << 17, 1 << 14, 4096):
    key = (np.arange(m) // g) * 8 + octant
    run(f"sorted by octant within {g}", np.argsort(key, kind="stable"))
run("random permutation", rng.permutation(m))
