import sys, os, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.getcwd())
import numpy as np
from raytracer_amd.scene import synthetic_scene
from raytracer_amd.renderer import HIPShader
s = synthetic_scene(100000, 3840, 2160, nthreads=16)
r = HIPShader(s)
px = np.zeros((r.height, r.width, 4), np.float32)
r.Render(px)
for k in range(3):
    t = time.perf_counter(); r.frame_counter = k + 1; r.Render(px); print("Render(px) ms", round((time.perf_counter() - t) * 1e3, 2))
t = time.perf_counter(); r.frame_counter = 9; r.Render(None); r.counters(); print("Render(None)+sync ms", round((time.perf_counter() - t) * 1e3, 2))
