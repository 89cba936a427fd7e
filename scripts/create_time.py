#!/usr/bin/env python3
"""rtx_create phase times (RTX_VERBOSE) for a synthetic scene: python scripts/create_time.py [triangles] [grid]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RTX_VERBOSE"] = "1"
from raytracer_amd.renderer import HIPShader
from raytracer_amd.scene import synthetic_scene
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
grid = (10, 10) if len(sys.argv) > 2 else None
t = time.time(); s = synthetic_scene(n, 3840, 2160, nthreads=16, grid=grid); print("scene + reference BVH s", round(time.time() - t, 2), flush=True)
for k in range(2):
    t = time.time(); r = HIPShader(s); dt = time.time() - t
    i = r.info()
    print(f"rtx_create {dt:.2f} s  phases {[round(x, 3) for x in i.createSeconds]}  device GB {i.deviceBytes / 1e9:.2f}  wide nodes {i.wideNodes} depth {i.wideTreeDepth}", flush=True)
    r.close()
