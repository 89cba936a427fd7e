"""ctypes declarations for include/rtx.h and include/rtx_host.h.

The libraries are loaded lazily and loudly: a missing ``librtx_hip.so`` raises —
there is no CPU fallback behind the render API.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class Vec4(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("w", C.c_float)]


class Camera(C.Structure):
    """RtxCamera — 96 bytes, src/Camera.h:9-65."""
    _fields_ = [("front", Vec4), ("up", Vec4), ("right", Vec4), ("position", Vec4),
                ("movementSpeed", C.c_float), ("rotationSpeed", C.c_float), ("aspectRatio", C.c_float),
                ("nearView", C.c_float), ("farView", C.c_float), ("fov", C.c_float),
                ("pitch", C.c_float), ("yaw", C.c_float)]


class SceneDesc(C.Structure):
    _fields_ = [("objects", C.c_void_p), ("numObjects", C.c_uint64),
                ("materials", C.c_void_p), ("numMaterials", C.c_uint64),
                ("boxes", C.c_void_p), ("numBoxes", C.c_uint64),
                ("textureInfo", C.c_void_p), ("numTextures", C.c_uint64),
                ("textureData", C.c_void_p), ("numTexels", C.c_uint64),
                ("width", C.c_uint32), ("height", C.c_uint32), ("useBVH", C.c_int32), ("_pad", C.c_int32)]


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("rank", C.c_int32), ("worldSize", C.c_int32),
                ("tileW", C.c_int32), ("tileH", C.c_int32), ("traversal", C.c_int32),
                ("countStats", C.c_int32), ("framesInFlight", C.c_int32),
                ("numDevices", C.c_int32), ("devices", C.c_int32 * 8), ("lookAhead", C.c_int32), ("framesPerBatch", C.c_int32)]


class Counters(C.Structure):
    _fields_ = [("frames", C.c_uint64), ("rays", C.c_uint64),
                ("raysPerBounce", C.c_uint64 * 4), ("hitsPerBounce", C.c_uint64 * 4),
                ("innerVisits", C.c_uint64 * 4), ("leafTests", C.c_uint64 * 4),
                ("lastFrameMs", C.c_double), ("lastTraceMs", C.c_double), ("lastShadeMs", C.c_double),
                ("totalMs", C.c_double), ("kernelMs", C.c_double * 8), ("wavePasses", C.c_uint64 * 4)]


class Info(C.Structure):
    _fields_ = [("traversal", C.c_int32), ("linear", C.c_int32), ("bigNodes", C.c_int32), ("cullSlack", C.c_float),
                ("framesInFlight", C.c_int32), ("framesPerBatch", C.c_int32), ("hwQueues", C.c_int32),
                ("refTreeDepth", C.c_int32), ("wideTreeDepth", C.c_int32), ("stackEntries", C.c_int32),
                ("leaves", C.c_uint64), ("innerNodes", C.c_uint64), ("wideNodes", C.c_uint64),
                ("sliverFraction", C.c_double), ("deviceBytes", C.c_uint64), ("createSeconds", C.c_double * 4),
                ("devices", C.c_int32), ("lookAhead", C.c_int32), ("setsCutShort", C.c_uint64)]


# numpy mirrors of the reference PODs (sizes/offsets as in include/rtx.h)
VEC4 = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("w", "<f4")])
OBJECT = np.dtype({"names": ["type", "radius", "position", "normals", "vertices", "uvs", "materialID"],
                   "formats": ["<i4", "<f4", ("<f4", 4), ("<f4", (3, 4)), ("<f4", (3, 4)), ("<f4", (3, 4)), "<i4"],
                   "offsets": [0, 4, 16, 32, 80, 128, 176], "itemsize": 256})
MATERIAL = np.dtype({"names": ["albedo", "tint", "specular", "transmissionFilter", "specularIntensity", "transparency",
                               "indexOfRefraction", "roughness", "metallic", "sheen", "tintRoughness",
                               "clearcoatThickness", "clearcoatRoughness", "emmissionIntensity", "anisotropy",
                               "anisotropyRotation", "textureID"],
                     "formats": [("<f4", 4)] * 4 + ["<f4"] * 12 + ["<i4"],
                     "offsets": [0, 16, 32, 48] + list(range(64, 112, 4)) + [112], "itemsize": 128})
BOX = np.dtype({"names": ["objectID", "parentID", "leftID", "rightID", "min", "max"],
                "formats": ["<i4", "<i4", "<i4", "<i4", ("<f4", 4), ("<f4", 4)],
                "offsets": [0, 4, 8, 12, 16, 32], "itemsize": 64})
TEXTURE = np.dtype({"names": ["width", "height", "offset", "checksum"],
                    "formats": ["<i4", "<i4", "<i4", "<u4"], "offsets": [0, 4, 8, 12], "itemsize": 16})

TRAVERSE_WIDE, TRAVERSE_EXACT, TRAVERSE_CULLED = 0, 1, 2
SPHERE, TRIANGLE = 0, 4

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)
_u32p = C.POINTER(C.c_uint32)


def _load(name):
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        raise RuntimeError(
            f"{name} is not built ({path}); run `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C raytracer_amd/csrc`. There is no CPU fallback for the HIP backend.")
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


_host = None
_hip = None


def host_lib():
    global _host
    if _host is None:
        L = _load("librtx_host.so")
        L.rtxh_scene_new.restype = C.c_void_p
        L.rtxh_scene_new.argtypes = [C.c_uint32, C.c_uint32]
        L.rtxh_scene_free.argtypes = [C.c_void_p]
        L.rtxh_scene_set_resolution.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        for fn in ("rtxh_scene_load_scn", "rtxh_scene_load_mtl", "rtxh_scene_load_obj", "rtxh_scene_save_blob"):
            getattr(L, fn).argtypes = [C.c_void_p, C.c_char_p]
            getattr(L, fn).restype = C.c_int
        L.rtxh_scene_material_id.argtypes = [C.c_void_p, C.c_char_p]
        L.rtxh_scene_add_texture.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
        L.rtxh_scene_add_material.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p]
        L.rtxh_scene_add_plane.argtypes = [C.c_void_p, _f32p, _f32p, _f32p, C.c_int]
        L.rtxh_scene_add_disk.argtypes = [C.c_void_p, _f32p, _f32p, C.c_float, C.c_int]
        L.rtxh_scene_add_cube.argtypes = [C.c_void_p, _f32p, _f32p, C.c_int]
        L.rtxh_scene_add_sphere.argtypes = [C.c_void_p, _f32p, C.c_float, C.c_int]
        L.rtxh_scene_add_mesh.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                          C.c_uint64, C.c_int]
        L.rtxh_scene_add_heightfield.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_float, C.c_float, C.c_float,
                                                 C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int32]
        L.rtxh_write_heightfield_obj.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint32, C.c_float, C.c_float, C.c_float,
                                                 C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int32]
        L.rtxh_scene_build_bvh.argtypes = [C.c_void_p, C.c_int]
        L.rtxh_scene_desc.argtypes = [C.c_void_p, C.POINTER(SceneDesc)]
        L.rtxh_scene_camera.argtypes = [C.c_void_p, C.POINTER(Camera)]
        L.rtxh_scene_set_camera.argtypes = [C.c_void_p, C.POINTER(Camera)]
        L.rtxh_camera_look_at.argtypes = [C.POINTER(Camera), _f32p]
        L.rtxh_camera_look_at.restype = None
        L.rtxh_camera_orbit.argtypes = [C.POINTER(Camera), C.c_float, C.c_float, C.c_float]
        L.rtxh_camera_orbit.restype = None
        L.rtxh_camera_rotate.argtypes = [C.POINTER(Camera), C.c_double, C.c_double]
        L.rtxh_camera_rotate.restype = None
        L.rtxh_scene_from_arrays.restype = C.c_void_p
        L.rtxh_scene_from_arrays.argtypes = [C.POINTER(SceneDesc), C.POINTER(Camera), C.c_uint32]
        L.rtxh_scene_load_blob.restype = C.c_void_p
        L.rtxh_scene_load_blob.argtypes = [C.c_char_p]
        L.rtxh_last_error.restype = C.c_char_p
        _host = L
    return _host


def hip_lib():
    global _hip
    if _hip is None:
        L = _load(os.environ.get("RTX_HIP_LIB", "librtx_hip.so"))
        L.rtx_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(Config), C.POINTER(C.c_void_p)]
        L.rtx_render.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_uint32, C.c_void_p]
        L.rtx_primary_hits.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_uint32, C.c_void_p]
        L.rtx_trace_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rtx_trace_rays_noalloc.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        for fn in ("rtx_upload_pixels", "rtx_download_pixels"):
            getattr(L, fn).argtypes = [C.c_void_p, C.c_void_p]
        L.rtx_clear_pixels.argtypes = [C.c_void_p]
        L.rtx_present_rgba8.argtypes = [C.c_void_p, C.c_float, C.c_void_p]
        L.rtx_alloc_host.argtypes = [C.c_uint64]
        L.rtx_alloc_host.restype = C.c_void_p
        L.rtx_free_host.argtypes = [C.c_void_p]
        L.rtx_free_host.restype = None
        L.rtx_get_counters.argtypes = [C.c_void_p, C.POINTER(Counters)]
        L.rtx_get_info.argtypes = [C.c_void_p, C.POINTER(Info)]
        L.rtx_reset_counters.argtypes = [C.c_void_p]
        L.rtx_destroy.argtypes = [C.c_void_p]
        L.rtx_destroy.restype = None
        L.rtx_last_error.restype = C.c_char_p
        L.rtx_framebuffer_device_ptr.argtypes = [C.c_void_p]
        L.rtx_framebuffer_device_ptr.restype = C.c_void_p
        L.rtx_stream.argtypes = [C.c_void_p]
        L.rtx_stream.restype = C.c_void_p
        L.rtx_local_pixel_count.argtypes = [C.c_void_p]
        L.rtx_local_pixel_count.restype = C.c_uint64
        L.rtx_local_tile_count.argtypes = [C.c_void_p]
        L.rtx_local_tile_count.restype = C.c_uint32
        L.rtx_packed_floats.argtypes = [C.c_void_p, C.c_int32]
        L.rtx_packed_floats.restype = C.c_uint64
        L.rtx_pack_tiles.argtypes = [C.c_void_p, C.c_void_p]
        L.rtx_unpack_tiles.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.rtx_version.restype = C.c_char_p
        _hip = L
    return _hip


EXPORTED_HIP = ["rtx_create", "rtx_render", "rtx_primary_hits", "rtx_trace_rays", "rtx_trace_rays_noalloc", "rtx_upload_pixels",
                "rtx_download_pixels", "rtx_clear_pixels", "rtx_present_rgba8", "rtx_alloc_host", "rtx_free_host", "rtx_get_info", "rtx_get_counters", "rtx_reset_counters", "rtx_destroy",
                "rtx_last_error", "rtx_framebuffer_device_ptr", "rtx_stream", "rtx_local_pixel_count",
                "rtx_local_tile_count", "rtx_packed_floats", "rtx_pack_tiles", "rtx_unpack_tiles", "rtx_version"]
EXPORTED_HOST = ["rtxh_scene_new", "rtxh_scene_free", "rtxh_scene_set_resolution", "rtxh_scene_load_scn", "rtxh_scene_load_mtl",
                 "rtxh_scene_load_obj", "rtxh_scene_material_id", "rtxh_scene_add_texture", "rtxh_scene_add_material",
                 "rtxh_scene_add_plane", "rtxh_scene_add_disk", "rtxh_scene_add_cube", "rtxh_scene_add_sphere",
                 "rtxh_scene_add_mesh", "rtxh_scene_add_heightfield", "rtxh_write_heightfield_obj", "rtxh_scene_build_bvh", "rtxh_scene_from_arrays", "rtxh_scene_desc",
                 "rtxh_scene_camera", "rtxh_scene_set_camera", "rtxh_camera_look_at", "rtxh_camera_orbit", "rtxh_camera_rotate",
                 "rtxh_scene_save_blob", "rtxh_scene_load_blob", "rtxh_last_error"]
