#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/ (run in the build
container, where /root/reference is mounted):

  scene2.blob    resources/scenes/2.scn loaded at -w 1000 -h 1000 (992x992 after the
                 reference's rounding) with the BVH built, then re-targeted to 200x200
                 pixels (same camera, same geometry) and with its 1024^2 textures
                 box-filtered to 64^2 so the blob stays small
  scene1.blob    resources/scenes/1.scn the same way (dragon.obj is absent upstream, so:
                 disk light, cube, floor plane and three spheres incl. the Glass one)
  *_ids_f0.npy   primary-hit object ids, frame 0            (oracle)
  *_fb_f1.npy    framebuffer after 1 frame                  (oracle)
  *_fb_f4.npy    framebuffer after 4 accumulated frames     (oracle)

The blobs are inputs (data); the expected outputs come from oracle/liboracle.so,
whose pinning against the reference is described in DESIGN.md.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle  # noqa: E402
from raytracer_amd.scene import HostScene, scene_from_arrays, shrink_textures  # noqa: E402

REF = "/root/reference/resources/scenes"
OUT = os.path.join(HERE, "golden")
RES = 200


def make(name, scn):
    full = HostScene(1000, 1000).load_scn(os.path.join(REF, scn)).build_bvh(8)
    small = shrink_textures(full, 64)
    s = scene_from_arrays(RES, RES, small.objects, small.materials, small.boxes, small.texture_info,
                          small.texture_data, small.camera)
    s.save_blob(os.path.join(OUT, name + ".blob"))
    d, cam = s.desc(), s.camera
    ids = oracle.primary_hits(d, cam, 0)
    np.save(os.path.join(OUT, name + "_ids_f0.npy"), ids)
    px = np.zeros((RES, RES, 4), np.float32)
    for f in range(4):
        oracle.render(d, cam, f, px, 8)
        if f == 0:
            np.save(os.path.join(OUT, name + "_fb_f1.npy"), px)
    np.save(os.path.join(OUT, name + "_fb_f4.npy"), px)
    print(name, "objects", d.numObjects, "boxes", d.numBoxes, "hits", int((ids >= 0).sum()), "nan px", int(np.isnan(px).any(axis=2).sum()))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    make("scene2", "2.scn")
    make("scene1", "1.scn")
