"""Experiment: two renderers (two streams) on one GPU rendering alternate frames — how much do the
tails of one frame's launches overlap with the other's work?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd.scene import synthetic_scene
from raytracer_amd.renderer import HIPShader
world = int(os.environ.get("RTX_QUICK_WORLD", "1"))
s = synthetic_scene(10_000_000, 3840, 2160, nthreads=16)
rs = [HIPShader(s, rank=0, world_size=world) for _ in range(4)]
def run(n_inst, frames):
    for r in rs: r.counters()
    t = time.perf_counter()
    for f in range(frames):
        r = rs[f % n_inst]; r.frame_counter = f; r.Render(None)
    for r in rs: r.counters()
    return (time.perf_counter() - t) / frames * 1e3
for n in (1, 2, 3, 4):
    run(n, 4)
    print(f"world {world}: {n} stream(s): {run(n, 24):.3f} ms/frame wall", flush=True)
