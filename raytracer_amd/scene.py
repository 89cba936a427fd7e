"""Host-side scene container (thin wrapper over librtx_host.so) and the synthetic
benchmark scenes of BASELINE.json.

The arrays a scene exposes are the five arrays of the reference's
RenderingContext (src/RenderingContext.h:31-40) in their reference layouts.
"""
import ctypes as C
import os

import numpy as np

from . import _capi
from ._capi import Camera, SceneDesc


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


class HostScene:
    def __init__(self, width=1000, height=1000, _handle=None):
        self._L = _capi.host_lib()
        self._h = _handle if _handle is not None else self._L.rtxh_scene_new(int(width), int(height))
        if not self._h:
            raise RuntimeError("rtxh_scene_new failed")
        self._keep = []

    def __del__(self):
        try:
            if self._h:
                self._L.rtxh_scene_free(self._h)
                self._h = None
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError(self._L.rtxh_last_error().decode())

    def set_resolution(self, width, height):
        """Exact width/height (no rounding); only before geometry is loaded."""
        self._chk(self._L.rtxh_scene_set_resolution(self._h, int(width), int(height)))
        return self

    # ---- ingest --------------------------------------------------------------
    def load_scn(self, path):
        self._chk(self._L.rtxh_scene_load_scn(self._h, os.fsencode(path)))
        return self

    def load_mtl(self, path):
        self._chk(self._L.rtxh_scene_load_mtl(self._h, os.fsencode(path)))
        return self

    def load_obj(self, path):
        self._chk(self._L.rtxh_scene_load_obj(self._h, os.fsencode(path)))
        return self

    def material_id(self, name):
        return self._L.rtxh_scene_material_id(self._h, name.encode())

    def add_texture(self, texels):
        t = np.ascontiguousarray(texels, dtype=np.uint32)
        return self._L.rtxh_scene_add_texture(self._h, t.ctypes.data, t.shape[1], t.shape[0])

    def add_material(self, name="", **fields):
        m = np.zeros(1, dtype=_capi.MATERIAL)
        m["albedo"] = (0.5, 0.5, 0.5, 1.0)
        m["indexOfRefraction"] = 1.45
        m["roughness"] = 0.5
        m["tintRoughness"] = 0.5
        for k, v in fields.items():
            m[k] = v
        return self._L.rtxh_scene_add_material(self._h, m.ctypes.data, name.encode())

    def add_plane(self, position, normal, scales, material=0):
        self._L.rtxh_scene_add_plane(self._h, _f3(position), _f3(normal), _f3(scales), material)

    def add_disk(self, position, normal, radius, material=0):
        self._L.rtxh_scene_add_disk(self._h, _f3(position), _f3(normal), float(radius), material)

    def add_cube(self, position, scales, material=0):
        self._L.rtxh_scene_add_cube(self._h, _f3(position), _f3(scales), material)

    def add_sphere(self, position, radius, material=0):
        self._L.rtxh_scene_add_sphere(self._h, _f3(position), float(radius), material)

    def add_mesh(self, vertices, normals, indices, material_ids=None, material=0):
        v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
        n = np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        i = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
        m = None if material_ids is None else np.ascontiguousarray(material_ids, dtype=np.int32)
        self._chk(self._L.rtxh_scene_add_mesh(self._h, v.ctypes.data, n.ctypes.data, len(v), i.ctypes.data,
                                              None if m is None else m.ctypes.data, len(i), material))

    def add_heightfield(self, ntri, seed, extent_u, extent_v, amplitude, tilt_deg, center, material_ids):
        m = np.ascontiguousarray(material_ids, dtype=np.int32)
        self._chk(self._L.rtxh_scene_add_heightfield(self._h, int(ntri), int(seed), float(extent_u), float(extent_v),
                                                     float(amplitude), float(tilt_deg), float(center[0]),
                                                     float(center[1]), float(center[2]), m.ctypes.data, len(m)))

    def write_heightfield_obj(self, path, ntri, seed, extent_u, extent_v, amplitude, tilt_deg, center, material_ids):
        """The mesh add_heightfield would add, as OBJ text (materials referenced by name through usemtl)."""
        m = np.ascontiguousarray(material_ids, dtype=np.int32)
        self._chk(self._L.rtxh_write_heightfield_obj(self._h, os.fsencode(path), int(ntri), int(seed), float(extent_u),
                                                     float(extent_v), float(amplitude), float(tilt_deg), float(center[0]),
                                                     float(center[1]), float(center[2]), m.ctypes.data, len(m)))

    def build_bvh(self, nthreads=8):
        self._chk(self._L.rtxh_scene_build_bvh(self._h, int(nthreads)))
        return self

    # ---- views ----------------------------------------------------------------
    def desc(self):
        d = SceneDesc()
        self._chk(self._L.rtxh_scene_desc(self._h, C.byref(d)))
        return d

    @property
    def camera(self):
        c = Camera()
        self._L.rtxh_scene_camera(self._h, C.byref(c))
        return c

    @camera.setter
    def camera(self, cam):
        self._L.rtxh_scene_set_camera(self._h, C.byref(cam))

    def _view(self, ptr, n, dtype):
        if not ptr or n == 0:
            return np.zeros(0, dtype=dtype)
        buf = (C.c_char * (int(n) * dtype.itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype, count=int(n))

    @property
    def width(self):
        return self.desc().width

    @property
    def height(self):
        return self.desc().height

    @property
    def objects(self):
        d = self.desc()
        return self._view(d.objects, d.numObjects, _capi.OBJECT)

    @property
    def materials(self):
        d = self.desc()
        return self._view(d.materials, d.numMaterials, _capi.MATERIAL)

    @property
    def boxes(self):
        d = self.desc()
        return self._view(d.boxes, d.numBoxes, _capi.BOX)

    @property
    def texture_info(self):
        d = self.desc()
        return self._view(d.textureInfo, d.numTextures, _capi.TEXTURE)

    @property
    def texture_data(self):
        d = self.desc()
        return self._view(d.textureData, d.numTexels, np.dtype("<u4"))

    # ---- blob -----------------------------------------------------------------
    def save_blob(self, path):
        self._chk(self._L.rtxh_scene_save_blob(self._h, os.fsencode(path)))

    @classmethod
    def load_blob(cls, path):
        L = _capi.host_lib()
        h = L.rtxh_scene_load_blob(os.fsencode(path))
        if not h:
            raise RuntimeError(L.rtxh_last_error().decode())
        return cls(_handle=h)


def scene_from_arrays(width, height, objects, materials, boxes, texture_info, texture_data, camera, use_bvh=True, depth=480):
    """New HostScene holding copies of numpy arrays in the reference layouts."""
    arrs = [np.ascontiguousarray(objects, dtype=_capi.OBJECT), np.ascontiguousarray(materials, dtype=_capi.MATERIAL),
            np.ascontiguousarray(boxes, dtype=_capi.BOX), np.ascontiguousarray(texture_info, dtype=_capi.TEXTURE),
            np.ascontiguousarray(texture_data, dtype=np.uint32)]
    d = SceneDesc()
    d.objects, d.numObjects = arrs[0].ctypes.data, len(arrs[0])
    d.materials, d.numMaterials = arrs[1].ctypes.data, len(arrs[1])
    d.boxes, d.numBoxes = (arrs[2].ctypes.data if len(arrs[2]) else None), len(arrs[2])
    d.textureInfo, d.numTextures = arrs[3].ctypes.data, len(arrs[3])
    d.textureData, d.numTexels = arrs[4].ctypes.data, len(arrs[4])
    d.width, d.height, d.useBVH = int(width), int(height), 1 if (use_bvh and len(arrs[2])) else 0
    L = _capi.host_lib()
    h = L.rtxh_scene_from_arrays(C.byref(d), C.byref(camera), depth)
    return HostScene(_handle=h)


def shrink_textures(scene, size=64):
    """Copy of `scene` whose textures are box-filtered down to at most size x size
    (integer arithmetic); used to keep committed fixtures small."""
    info = np.array(scene.texture_info, copy=True)
    data = scene.texture_data
    new_data, new_info = [], np.array(info, copy=True)
    off = 0
    for i, t in enumerate(info):
        w, h, o = int(t["width"]), int(t["height"]), int(t["offset"])
        tex = data[o:o + w * h].reshape(h, w)
        if w > size or h > size:
            fx, fy = w // size, h // size
            ch = [((tex >> (8 * k)) & 0xFF).astype(np.uint32)[:fy * size, :fx * size].reshape(size, fy, size, fx).sum(axis=(1, 3)) // (fx * fy)
                  for k in range(3)]
            tex = (ch[0] | (ch[1] << 8) | (ch[2] << 16) | np.uint32(255 << 24)).astype(np.uint32)
            w = h = size
        new_info[i]["width"], new_info[i]["height"], new_info[i]["offset"] = w, h, off
        new_data.append(np.ascontiguousarray(tex).reshape(-1))
        off += w * h
    return scene_from_arrays(scene.width, scene.height, scene.objects, scene.materials, scene.boxes, new_info,
                             np.concatenate(new_data), scene.camera, use_bvh=len(scene.boxes) > 0)


def look_at(cam, target):
    _capi.host_lib().rtxh_camera_look_at(C.byref(cam), _f3(target))
    return cam


def orbit(cam, yaw_deg, pitch_deg, radius=5000.0):
    """One camera placement of the -O loop (src/Main.cpp:68-77)."""
    _capi.host_lib().rtxh_camera_orbit(C.byref(cam), float(yaw_deg), float(pitch_deg), float(radius))
    return cam


# ---- synthetic scenes -----------------------------------------------------------

def procedural_texture(size, seed):
    """Deterministic RGBA8 texture (R|G<<8|B<<16|255<<24), integer arithmetic only."""
    y, x = np.mgrid[0:size, 0:size].astype(np.uint32)
    h = (x * np.uint32(73856093)) ^ (y * np.uint32(19349663)) ^ np.uint32(seed * 83492791 & 0xFFFFFFFF)
    h ^= h >> np.uint32(13)
    h = h * np.uint32(1274126177)
    h ^= h >> np.uint32(16)
    stripes = ((x // np.uint32(max(1, size // 16)) + y // np.uint32(max(1, size // 16))) & np.uint32(1)) * np.uint32(60)
    r = (np.uint32(120) + (h & np.uint32(63)) + stripes) & np.uint32(255)
    g = (np.uint32(100) + ((h >> np.uint32(6)) & np.uint32(63)) + stripes) & np.uint32(255)
    b = (np.uint32(90) + ((h >> np.uint32(12)) & np.uint32(63)) + (np.uint32(seed * 40) & np.uint32(127))) & np.uint32(255)
    return (r | (g << np.uint32(8)) | (b << np.uint32(16)) | np.uint32(255 << 24)).astype(np.uint32)


def synthetic_scene(ntri, width, height, seed=12345, textured=True, tex_size=1024, nthreads=8, bvh=True, via_obj=None, grid=None):
    """Floor plane + disk light (as resources/scenes/2.scn) + a seeded height-field
    mesh of exactly `ntri` triangles that fills the default camera's view
    (SURVEY.md §8d). Texture 1 (hard-wired as the sky, src/ThreadedShader.cpp:162)
    is a procedural sky texture.
    via_obj: path of an OBJ text file to write the mesh to and load it back from (the reference's own mesh route,
    src/MeshSerializer.cpp:141-243; BASELINE.json configs[2] "1 M-tri OBJ") instead of filling the objects directly.
    grid = (nx, ny): the mesh is the base mesh of ntri // (nx*ny) triangles replicated on an nx x ny grid of
    explicit copies (BASELINE.json configs[4]: a 1 M-triangle base mesh x 100; the reference has no instancing,
    SURVEY.md H6)."""
    s = HostScene(width, height)
    w, h = s.width, s.height
    sky = s.add_texture(procedural_texture(tex_size, 1))
    assert sky == 1
    tex = [s.add_texture(procedural_texture(tex_size, 2 + i)) for i in range(3)] if textured else [0, 0, 0]
    light = s.add_material("Light", albedo=(1, 1, 1, 1), emmissionIntensity=0.5)
    floor = s.add_material("Floor", albedo=(0.5, 0.5, 0.5, 1), tint=(0.5, 0.5, 0.5, 1), roughness=1.0, textureID=tex[2])
    mats = []
    for i in range(8):
        mats.append(s.add_material(
            f"Mesh{i}",
            albedo=(0.45 + 0.05 * (i % 3), 0.5, 0.35 + 0.05 * (i % 4), 1.0),
            tint=(0.8, 0.8, 0.6, 1.0), specular=(1, 1, 1, 1),
            roughness=0.3 + 0.1 * (i % 5), metallic=1.0 if i % 4 == 3 else 0.0,
            sheen=0.25 * (i % 2), clearcoatThickness=0.5 * (i % 3 == 1), clearcoatRoughness=0.5,
            anisotropy=0.5 * (i % 2), indexOfRefraction=1.5, textureID=tex[i % 3]))
    s.add_disk((500, 4000, 500), (0, -1, 0), 1000, light)
    s.add_plane((500, 100, 500), (0, 1, 0), (5000, 5000, 5000), floor)
    # object-space rectangle that, after v*100*aspect + (w/2,h/2,240), covers the view of the
    # default camera (500,500,-900), fov 45 deg, looking +z (src/Configurator.cpp:10, src/Camera.h:20-22)
    aspect = np.float32(w) / np.float32(h)
    scale = 100.0 * float(aspect)
    half_h = 0.41421356 * (900 + 240)
    half_w = half_h * float(aspect)
    cx, cy, cz = (500 - w / 2.0) / scale, (560 - h / 2.0) / scale, 0.0
    hf = (seed, 2.1 * half_w / scale, 2.1 * half_h / scale, 60.0 / scale, 25.0)
    if grid is not None:
        nx, ny = grid
        base = ntri // (nx * ny)
        assert base * nx * ny == ntri
        # copies side by side, each 1/nx x 1/ny of the extent: same footprint as the single mesh, one seed per copy
        for j in range(ny):
            for i in range(nx):
                c = (cx + ((i + 0.5) / nx - 0.5) * hf[1], cy + ((j + 0.5) / ny - 0.5) * hf[2] * float(np.cos(np.radians(25.0))),
                     cz + ((j + 0.5) / ny - 0.5) * hf[2] * float(np.sin(np.radians(25.0))))
                s.add_heightfield(base, seed, hf[1] / nx, hf[2] / ny, hf[3] / max(nx, ny), hf[4], c, mats)
    elif via_obj is not None:
        s.write_heightfield_obj(via_obj, ntri, *hf, (cx, cy, cz), mats)
        s.load_obj(via_obj)
    else:
        s.add_heightfield(ntri, *hf, (cx, cy, cz), mats)
    if bvh:
        s.build_bvh(nthreads)
    return s
