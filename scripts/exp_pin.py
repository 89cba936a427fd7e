"""The synchronous loop into a pageable array, into an array from rtx_alloc_host, and into a
hipHostMalloc'ed array (a pinned torch tensor): ms per delivered 4K frame."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracer_amd import _capi
from raytracer_amd.renderer import HIPShader
from raytracer_amd.scene import synthetic_scene
scene = synthetic_scene(10_000_000, 3840, 2160, seed=12345, textured=True, nthreads=16, grid=None)
L = _capi.hip_lib()
def loop(sh, px, n=64, warm=12):
    for i in range(warm): sh.frame_counter = i; sh.Render(px)
    t0 = time.perf_counter()
    for i in range(n): sh.frame_counter = warm + i; sh.Render(px)
    return (time.perf_counter() - t0) / n * 1e3
sh = HIPShader(scene)
px = np.zeros((sh.height, sh.width, 4), np.float32)
print(f"pageable            {loop(sh, px):.3f} ms/frame", flush=True)
import ctypes
mem = L.rtx_alloc_host(px.nbytes)
own = np.ctypeslib.as_array(ctypes.cast(mem, ctypes.POINTER(ctypes.c_float)), shape=px.shape)
print(f"rtx_alloc_host      {loop(sh, own):.3f} ms/frame", flush=True)
print(f"pageable again      {loop(sh, px):.3f} ms/frame", flush=True)
pt = torch.zeros((sh.height, sh.width, 4), dtype=torch.float32).pin_memory()
print(f"hipHostMalloc'ed    {loop(sh, pt.numpy()):.3f} ms/frame", flush=True)
sh.close()
del own
L.rtx_free_host(mem)
