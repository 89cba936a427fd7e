"""ctypes access to the CHECKER: oracle/liboracle.so (CPU restatement of the
reference's hot path) and, when present, oracle/_ref/libref_parts.so (the part of
the reference that compiles standalone).

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by raytracer_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class OrcStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("raysPerBounce", C.c_uint64 * 4), ("hitsPerBounce", C.c_uint64 * 4),
                ("innerVisits", C.c_uint64), ("leafTests", C.c_uint64)]


_lib = None
_lib_ofast = None
_ref = None


def build():
    """Compile liboracle.so (and _ref/ when /root/reference is present)."""
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def _bind(path):
    L = C.CDLL(path)
    L.orc_render_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.POINTER(OrcStats)]
    L.orc_render_bands.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(OrcStats)]
    L.orc_render.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int]
    L.orc_primary_hits.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_int]
    L.orc_primary_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.orc_trace_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.orc_rand.argtypes = [C.POINTER(C.c_uint32)]
    L.orc_rand.restype = C.c_float
    L.orc_random_direction.argtypes = [C.POINTER(C.c_uint32), C.c_void_p]
    L.orc_pixel_position.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.orc_shading_terms.argtypes = [C.c_void_p] * 7
    L.orc_get_texture_pixel.argtypes = [C.c_void_p] * 6
    L.orc_color_sample.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    L.orc_vec_ops.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    L.orc_vec_geom.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    L.orc_color_ops.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    return L


def lib():
    """The strict build (-O2 -ffp-contract=off): the parity oracle."""
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _lib = _bind(path)
    return _lib


def lib_ofast():
    """The same source with the reference's shipped flags (-Ofast, CMakeLists.txt:11). Never a parity oracle (its
    rsqrtss-based Normalize is host-dependent): used to reproduce the reference's recorded -Ofast outputs and as the
    faster end of the CPU baseline bracket. None if not built."""
    global _lib_ofast
    if _lib_ofast is None:
        path = os.path.join(_HERE, "liboracle_ofast.so")
        if not os.path.exists(path):
            return None
        _lib_ofast = _bind(path)
    return _lib_ofast


def ref_parts():
    """The standalone-compilable part of the reference, or None if not built."""
    global _ref
    if _ref is None:
        path = os.path.join(_HERE, "_ref", "libref_parts.so")
        if not os.path.exists(path):
            return None
        L = C.CDLL(path)
        L.ref_rand.argtypes = [C.POINTER(C.c_uint32)]
        L.ref_rand.restype = C.c_float
        L.ref_pixel_position.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.ref_shading_terms.argtypes = [C.c_void_p] * 7
        L.ref_get_texture_pixel.argtypes = [C.c_void_p] * 6
        L.ref_color_sample.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.ref_vec_ops.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        L.ref_vec_geom.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        L.ref_color_ops.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        if hasattr(L, "ref_camera_look_at"):
            L.ref_camera_look_at.argtypes = [C.c_void_p, C.c_void_p]
            L.ref_camera_rotate.argtypes = [C.c_void_p, C.c_double, C.c_double]
            L.ref_box_layout.argtypes = [C.c_void_p]
            L.ref_box_expand.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
            L.ref_box_expand.restype = C.c_float
        if hasattr(L, "ref_read_bmp"):
            L.ref_read_bmp.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_void_p, C.c_uint64]
            L.ref_read_bmp.restype = C.c_uint64
        _ref = L
    return _ref


def _p(x):
    return C.byref(x) if isinstance(x, C.Structure) else x


def render(desc, cam, frame_counter, pixels, nthreads=8, rows=None, stats=0, L=None):
    """One ThreadedShader::Render over `pixels` (h*w*4 float32, updated in place)."""
    assert pixels.dtype == np.float32 and pixels.flags.c_contiguous
    y0, y1 = (0, desc.height) if rows is None else rows
    st = OrcStats()
    (L or lib()).orc_render_rows(C.byref(desc), C.byref(cam), frame_counter, pixels.ctypes.data, y0, y1, nthreads, stats,
                                 C.byref(st))
    return st


def render_bands(desc, cam, frame_counter, pixels, nthreads=8, rows=None, L=None):
    """The same with ThreadedShader::Render's own schedule: static contiguous row bands, one thread each
    (src/ThreadedShader.cpp:186-207). Returns ray statistics."""
    assert pixels.dtype == np.float32 and pixels.flags.c_contiguous
    y0, y1 = (0, desc.height) if rows is None else rows
    st = OrcStats()
    (L or lib()).orc_render_bands(C.byref(desc), C.byref(cam), frame_counter, pixels.ctypes.data, y0, y1, nthreads, C.byref(st))
    return st


def primary_hits(desc, cam, frame_counter=0, mode=0, nthreads=8, L=None):
    ids = np.empty((desc.height, desc.width), dtype=np.int32)
    (L or lib()).orc_primary_hits(C.byref(desc), C.byref(cam), frame_counter, ids.ctypes.data, mode, nthreads)
    return ids


def primary_rays(desc, cam, frame_counter=0):
    o = np.empty((desc.height * desc.width, 4), dtype=np.float32)
    d = np.empty_like(o)
    lib().orc_primary_rays(C.byref(desc), C.byref(cam), frame_counter, o.ctypes.data, d.ctypes.data)
    return o, d


def trace_rays(desc, origins, directions, mode=0, nthreads=8, count=False):
    o = np.ascontiguousarray(origins, dtype=np.float32)
    d = np.ascontiguousarray(directions, dtype=np.float32)
    n = len(o)
    ids = np.empty(n, dtype=np.int32)
    pts = np.empty((n, 4), dtype=np.float32)
    nrm = np.empty((n, 4), dtype=np.float32)
    inner, leaf = C.c_uint64(0), C.c_uint64(0)
    lib().orc_trace_rays(C.byref(desc), o.ctypes.data, d.ctypes.data, n, ids.ctypes.data, pts.ctypes.data,
                         nrm.ctypes.data, mode, nthreads, C.byref(inner) if count else None,
                         C.byref(leaf) if count else None)
    if count:
        return ids, pts, nrm, inner.value, leaf.value
    return ids, pts, nrm


def fnv1a64(data):
    """64-bit FNV-1a over raw bytes (the hash BASELINE.md §2 quotes)."""
    b = np.frombuffer(np.ascontiguousarray(data).tobytes(), dtype=np.uint8)
    h = 0xcbf29ce484222325
    # chunked pure-python is slow; use a small vectorised loop over bytes via int arithmetic
    prime = 0x100000001b3
    mask = (1 << 64) - 1
    for byte in b.tolist():
        h = ((h ^ byte) * prime) & mask
    return h
