#!/usr/bin/env python3
"""Tries the obvious hashing conventions for the two hashes the survey recorded from the reference's own strict
build without recording how they were computed:
    primary-hit ID map of 2.scn, 992x992, frame 0        07544e4368e87eca
    one-frame framebuffer of 2.scn, 1000x1000 (-O2)      79a2746fda3ea8c6
Prints every (data layout, hash function) pair and marks a match. Needs /root/reference (build container only).
tests/test_oracle.py::test_recorded_hashes_* runs the same list and asserts whichever outcome this script found."""
import ctypes
import os
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SRC = r'''
#include <stdint.h>
#include <stddef.h>
uint64_t fnv1a64(const uint8_t* p, size_t n){uint64_t h=0xcbf29ce484222325ull;for(size_t i=0;i<n;i++){h^=p[i];h*=0x100000001b3ull;}return h;}
uint64_t fnv1_64(const uint8_t* p, size_t n){uint64_t h=0xcbf29ce484222325ull;for(size_t i=0;i<n;i++){h*=0x100000001b3ull;h^=p[i];}return h;}
uint64_t fnv1a64_words(const uint32_t* p, size_t n){uint64_t h=0xcbf29ce484222325ull;for(size_t i=0;i<n;i++){h^=p[i];h*=0x100000001b3ull;}return h;}
uint64_t fnv1a64_swords(const int32_t* p, size_t n){uint64_t h=0xcbf29ce484222325ull;for(size_t i=0;i<n;i++){h^=(uint64_t)(int64_t)p[i];h*=0x100000001b3ull;}return h;}
uint64_t fnv1a32(const uint8_t* p, size_t n){uint32_t h=0x811c9dc5u;for(size_t i=0;i<n;i++){h^=p[i];h*=0x01000193u;}return h;}
'''
_H = None


def hashes():
    global _H
    if _H is None:
        d = tempfile.mkdtemp()
        open(d + "/h.c", "w").write(SRC)
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", d + "/h.so", d + "/h.c"])
        _H = ctypes.CDLL(d + "/h.so")
        for f in ("fnv1a64", "fnv1_64", "fnv1a64_words", "fnv1a64_swords", "fnv1a32"):
            getattr(_H, f).restype = ctypes.c_uint64
            getattr(_H, f).argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    return _H


def all_hashes(arr):
    H = hashes()
    b = np.ascontiguousarray(arr)
    out = {"fnv1a64/bytes": H.fnv1a64(b.ctypes.data, b.nbytes), "fnv1_64/bytes": H.fnv1_64(b.ctypes.data, b.nbytes),
           "fnv1a32/bytes": H.fnv1a32(b.ctypes.data, b.nbytes)}
    if b.dtype.itemsize == 4:
        out["fnv1a64/u32 words"] = H.fnv1a64_words(b.ctypes.data, b.nbytes // 4)
        out["fnv1a64/sign-extended words"] = H.fnv1a64_swords(b.ctypes.data, b.nbytes // 4)
    return {k: "%016x" % v for k, v in out.items()}


def id_layouts(ids):
    flat = ids.reshape(-1)
    return {"int32 row-major": flat.astype("<i4"), "int32 +1": (flat + 1).astype("<i4"), "int64": flat.astype("<i8"),
            "int32 big-endian": flat.astype(">i4"), "hit mask u8": (flat >= 0).astype("u1"), "int16": flat.astype("<i2"),
            "float32": flat.astype("<f4"), "column-major": ids.T.copy().reshape(-1).astype("<i4"),
            "rows bottom-up": ids[::-1].copy().reshape(-1).astype("<i4"), "uint32 (miss = 0xffffffff)": flat.astype("<u4")}


def fb_layouts(px):
    return {"float32 rgba row-major": px.reshape(-1), "float32 rgb": px[..., :3].copy().reshape(-1),
            "rows bottom-up": px[::-1].copy().reshape(-1), "float64": px.astype("<f8").reshape(-1),
            "uint8 (x255, truncated)": (np.clip(px, 0, 1) * 255).astype("u1").reshape(-1)}


def try_all(layouts, target):
    found = []
    for lname, arr in layouts.items():
        for hname, hx in all_hashes(arr).items():
            if hx == target or hx[-8:] == target[-8:] and hname.startswith("fnv1a32"):
                found.append((lname, hname))
    return found


def main():
    import oracle
    from raytracer_amd.scene import HostScene
    scn = "/root/reference/resources/scenes/2.scn"
    s = HostScene(1000, 1000).load_scn(scn).build_bvh(4)
    ids = oracle.primary_hits(s.desc(), s.camera, 0, mode=1)
    print("ids", ids.shape, int((ids >= 0).sum()), "hits; target 07544e4368e87eca")
    for lname, arr in id_layouts(ids).items():
        for hname, hx in all_hashes(arr).items():
            print(f"  {lname:28s} {hname:28s} {hx} {'MATCH' if hx == '07544e4368e87eca' else ''}")
    s2 = HostScene(1000, 1000).set_resolution(1000, 1000).load_scn(scn).build_bvh(4)
    px = np.zeros((1000, 1000, 4), np.float32)
    oracle.render(s2.desc(), s2.camera, 0, px, 8)
    print("framebuffer rgb sum", float(px[..., :3].astype(np.float64).sum()), "target 79a2746fda3ea8c6")
    for lname, arr in fb_layouts(px).items():
        for hname, hx in all_hashes(arr).items():
            print(f"  {lname:28s} {hname:28s} {hx} {'MATCH' if hx == '79a2746fda3ea8c6' else ''}")


if __name__ == "__main__":
    main()
