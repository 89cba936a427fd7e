"""K = 20 queued frames (the driver's command) with different frames-per-batch settings: ms per frame incl. delivery."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd.renderer import HIPShader
from raytracer_amd.scene import synthetic_scene
scene = synthetic_scene(10_000_000, 3840, 2160, seed=12345, textured=True, nthreads=16, grid=None)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
WORLD = int(sys.argv[2]) if len(sys.argv) > 2 else 1      # emulate rank 0 of WORLD ranks (1/WORLD of the tiles)
for fpb in [0, 5, 7, 10, 20]:
    sh = HIPShader(scene, device=0, rank=0, world_size=WORLD, frames_per_batch=fpb)
    frame = np.zeros((sh.height, sh.width, 4), dtype=np.float32)
    best = 1e9
    for rep in range(4):
        for i in range(5):
            sh.frame_counter = i; sh.Render(None, scene.camera)
        sh.download_into(frame); sh.counters()
        t0 = time.perf_counter()
        for i in range(K):
            sh.frame_counter = 5 + i; sh.Render(None, scene.camera)
        sh.download_into(frame); sh.counters()
        dt = time.perf_counter() - t0
        best = min(best, dt)
    print(f"world={WORLD} fpb={fpb} batch={sh.info().framesPerBatch} inflight={sh.info().framesInFlight} K={K}: {best/K*1e3:.3f} ms/frame", flush=True)
    sh.close()
