"""Where does a frame differ from the oracle in bits? (scene1 / orbit camera as in tests/test_gpu_parity.py::test_moving_camera)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle
from raytracer_amd import _capi
from raytracer_amd.renderer import HIPShader
from raytracer_amd.scene import HostScene, orbit

s = HostScene.load_blob(os.path.join(ROOT, "tests", "golden", "scene1.blob"))
d = s.desc()
objs = np.array(s.objects, copy=False)
for i, (yaw, pitch) in enumerate(((10.0, 0.0), (50.0, 5.0), (90.0, 10.0), (0.0, 0.0), (33.0, 0.0))):
    cam = s.camera
    orbit(cam, yaw, pitch, radius=2500.0)
    for trav in (_capi.TRAVERSE_WIDE, _capi.TRAVERSE_EXACT):
        r = HIPShader(s, traversal=trav, look_ahead=-1, frames_per_batch=1, frames_in_flight=1)
        px = np.zeros((d.height, d.width, 4), np.float32); want = np.zeros_like(px)
        r.frame_counter = 0
        r.Render(px, cam)
        r.close()
        oracle.render(d, cam, 0, want, 8)
        ids = oracle.primary_hits(d, cam, 0)
        diff = (px.view(np.uint32) != want.view(np.uint32)).any(axis=2)
        print(f"cam {i} yaw {yaw} pitch {pitch} trav {trav}: {int(diff.sum())} pixels differ; front=({cam.front.x:.6f},{cam.front.y:.6f},{cam.front.z:.6f})")
        ys, xs = np.nonzero(diff)
        kinds = {}
        for y, x in zip(ys, xs):
            k = "sky" if ids[y, x] < 0 else f"type{objs['type'][ids[y, x]]}/mat{objs['materialID'][ids[y, x]]}"
            kinds[k] = kinds.get(k, 0) + 1
        print("   primary hit of the differing pixels:", kinds)
        for y, x in list(zip(ys, xs))[:4]:
            print("   ", y, x, px[y, x], want[y, x])
