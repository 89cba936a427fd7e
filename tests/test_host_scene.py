"""Host-side scene container: .scn/.mtl/.obj/.bmp ingest, tessellators, BVH build,
blob round trip (the inputs of the hot path)."""
import os
import struct

import numpy as np
import pytest

from conftest import REFERENCE, have_reference
from raytracer_amd import _capi
from raytracer_amd.scene import HostScene, synthetic_scene


def write_bmp_v5(path, w, h, rgb):
    """Minimal BITMAPV5HEADER 24-bpp file (the only flavour the reference reads, src/BitmapReader.cpp:28-31)."""
    row = (w * 3 + 3) & ~3
    data = bytearray()
    for y in range(h):
        line = bytearray()
        for x in range(w):
            r, g, b = rgb(x, y)
            line += bytes([b, g, r])
        line += bytes(row - w * 3)
        data += line
    header = b"BM" + struct.pack("<III", 14 + 124 + len(data), 0, 14 + 124)
    info = struct.pack("<IiiHHII", 124, w, h, 1, 24, 0, len(data)) + bytes(124 - 24)
    open(path, "wb").write(header + info + bytes(data))


SCN = """mtllib sky.mtl
mtllib m.mtl

!comment line is echoed

scene
{

disk
{
    position 500 3000 500
    normal 0 -1 0
    radius 800
    material Light
}

cube
{
    position -100 500 800
    scale 200 200 200
    material Shiny
}

plane
{
    position 500 100 500
    normal 0 1 0
    scale 5000
}

sphere
{
    position 300 500 500
    radius 100
    material Shiny
}

mesh tri.obj

}
"""
MTL = """newmtl Light
Ka 1.0 1.0 1.0
Ke 0.5
Pr 0.5

newmtl Shiny
Ka 0.5 0.25 0.125
Kd 0.8 0.8 0.6
Ks 1 1 1
Pm 2.0
Pc 0.5
Pcr 0.25
d  0.25
map_Ka tex.bmp

newmtl Plain
Ka 0.1 0.2 0.3

newmtl Shiny
Ka 0 0 0
"""
OBJ = """# tiny mesh
v 0 0 0
v 1 0 0
v 0 1 0
v 0 0 1
vn 0 0 1
vn 0 1 0
usemtl Plain
f 1//1 2//1 3//1
f 1/1/2 3/1/2 4/1/2
usemtl Nope
f 2 3 4
"""


@pytest.fixture()
def tiny_scene(tmp_path, libs):
    (tmp_path / "s.scn").write_text(SCN)
    (tmp_path / "sky.mtl").write_text("newmtl Skybox\nmap_Ka missing.bmp\n")
    (tmp_path / "m.mtl").write_text(MTL)
    (tmp_path / "tri.obj").write_text(OBJ)
    write_bmp_v5(str(tmp_path / "tex.bmp"), 5, 3, lambda x, y: (10 * x, 20 * y, 7))
    return HostScene(1000, 500).load_scn(str(tmp_path / "s.scn"))


def test_resolution_rounding_and_defaults(libs):
    # src/Configurator.cpp:144-149: (v+16)/32*32, min 32
    for req, got in [(1000, 992), (1080, 1088), (2160, 2176), (1920, 1920), (3840, 3840), (10, 32), (100, 96)]:
        assert HostScene(req, req).width == got
    s = HostScene(1920, 1080)
    cam = s.camera
    assert (cam.position.x, cam.position.y, cam.position.z) == (500.0, 500.0, -900.0)
    assert cam.aspectRatio == np.float32(1920) / np.float32(1088)
    assert len(s.materials) == 1 and len(s.texture_info) == 1 and s.texture_data[0] == 0xFFFFFFFF


def test_scn_ingest(tiny_scene):
    s = tiny_scene
    o = s.objects
    # disk 32 + cube 12 + plane 2 + sphere 1 + mesh 3
    assert len(o) == 50
    assert (o["type"][:46] == _capi.TRIANGLE).all() and o["type"][46] == _capi.SPHERE
    m = s.materials
    names = {n: s.material_id(n) for n in ("Skybox", "Light", "Shiny", "Plain", "Nope")}
    assert names == {"Skybox": 1, "Light": 2, "Shiny": 3, "Plain": 4, "Nope": 0}
    assert len(m) == 5          # the duplicate "Shiny" is dropped
    assert (o["materialID"][:32] == 2).all() and (o["materialID"][32:44] == 3).all() and (o["materialID"][44:46] == 0).all()
    assert o["materialID"][46] == 3 and list(o["materialID"][47:]) == [4, 4, 0]
    # Ke zeroes roughness, a later Pr sets it again; Pm is clamped; d inverts; clearcoat is sticky
    assert m["emmissionIntensity"][2] == 0.5 and m["roughness"][2] == 0.5
    assert m["metallic"][3] == 1.0 and m["transparency"][3] == 0.75
    assert m["clearcoatThickness"][3] == 0.5 and m["clearcoatThickness"][4] == 0.5 and m["clearcoatRoughness"][4] == 0.25
    # texture: rows in file order, texel = R | G<<8 | B<<16 | 255<<24
    ti = s.texture_info
    assert len(ti) == 2 and (ti["width"][1], ti["height"][1], ti["offset"][1]) == (5, 3, 1)
    assert m["textureID"][3] == 1 and m["textureID"][4] == 0
    t = s.texture_data[1:].reshape(3, 5)
    assert t[2, 4] == (40 | (40 << 8) | (7 << 16) | (255 << 24))
    # sphere
    assert tuple(o["position"][46][:3]) == (300.0, 500.0, 500.0) and o["radius"][46] == 100.0
    # plane "scale 5000" -> (5000, 5000, 0); corners at 45 degrees
    v = o["vertices"][44]
    assert np.allclose(v[:, 1], 100.0) and np.allclose(np.abs(v[:, [0, 2]] - 500.0), 5000 * np.sqrt(0.5), rtol=1e-6)
    # tessellated shapes keep position 0 (they sort first in the BVH median split)
    assert (o["position"][:46] == 0).all()
    # mesh placement: v*100*aspect + (w/2, h/2, 240), position = centroid (src/MeshSerializer.cpp:96-115)
    aspect = np.float32(992) / np.float32(512)
    a = np.float32(0) * (np.float32(100) * aspect) + np.float32(992 / 2.0)
    assert o["vertices"][47][0][0] == a and o["vertices"][47][1][0] == np.float32(1) * (np.float32(100) * aspect) + np.float32(496.0)
    assert tuple(o["normals"][47][0][:3]) == (0.0, 0.0, 1.0) and tuple(o["normals"][48][1][:3]) == (0.0, 1.0, 0.0)
    # "f 2 3 4": the vertex index doubles as the normal index (src/MeshSerializer.cpp:55-61)
    assert tuple(o["normals"][49][0][:3]) == (0.0, 1.0, 0.0)


def check_bvh(s):
    o, b = s.objects, s.boxes
    n = len(o)
    assert len(b) == 2 * n - 1
    leaf = b["objectID"] != -1
    assert leaf.sum() == n and sorted(b["objectID"][leaf]) == list(range(n))
    inner = np.where(~leaf)[0]
    l, r = b["leftID"][inner], b["rightID"][inner]
    assert (l == inner + 1).all()                                  # pre-order, left subtree first
    assert (np.minimum(b["min"][l], b["min"][r])[:, :3] == b["min"][inner][:, :3]).all()   # exact union
    assert (np.maximum(b["max"][l], b["max"][r])[:, :3] == b["max"][inner][:, :3]).all()
    tri = o["type"] == _capi.TRIANGLE
    lo = np.where(tri[:, None], o["vertices"][:, :, :3].min(axis=1), o["position"][:, :3] - o["radius"][:, None])
    hi = np.where(tri[:, None], o["vertices"][:, :, :3].max(axis=1), o["position"][:, :3] + o["radius"][:, None])
    ids = b["objectID"][leaf]
    assert (b["min"][leaf][:, :3] == lo[ids]).all() and (b["max"][leaf][:, :3] == hi[ids]).all()
    # median split on position[depth % 3]: left keys <= right keys, sizes n>>1 / n-(n>>1)
    def walk(node, depth):
        if b["objectID"][node] != -1:
            return [b["objectID"][node]]
        L, R = walk(b["leftID"][node], depth + 1), walk(b["rightID"][node], depth + 1)
        assert len(L) == (len(L) + len(R)) >> 1
        k = o["position"][:, depth % 3]
        assert k[L].max() <= k[R].min()
        return L + R
    assert sorted(walk(0, 0)) == list(range(n))


def test_bvh_structure(tiny_scene, libs):
    check_bvh(tiny_scene.build_bvh(1))
    s = synthetic_scene(3001, 256, 256, tex_size=16, nthreads=4)
    check_bvh(s)


def test_bvh_parallel_build_is_deterministic(libs):
    a = synthetic_scene(20000, 256, 256, tex_size=16, nthreads=1)
    b = synthetic_scene(20000, 256, 256, tex_size=16, nthreads=8)
    assert a.boxes.tobytes() == b.boxes.tobytes()


def test_blob_round_trip(tiny_scene, tmp_path):
    s = tiny_scene.build_bvh(2)
    p = str(tmp_path / "x.blob")
    s.save_blob(p)
    t = HostScene.load_blob(p)
    for attr in ("objects", "materials", "boxes", "texture_info", "texture_data"):
        assert getattr(s, attr).tobytes() == getattr(t, attr).tobytes()
    assert bytes(s.camera) == bytes(t.camera) and (s.width, s.height) == (t.width, t.height)


def test_synthetic_scene_is_seeded(libs):
    a = synthetic_scene(5000, 256, 256, seed=3, tex_size=16, nthreads=2)
    b = synthetic_scene(5000, 256, 256, seed=3, tex_size=16, nthreads=2)
    c = synthetic_scene(5000, 256, 256, seed=4, tex_size=16, nthreads=2)
    assert a.objects.tobytes() == b.objects.tobytes() and a.objects.tobytes() != c.objects.tobytes()
    assert len(a.objects) == 5000 + 32 + 2


@pytest.mark.skipif(not have_reference(), reason="needs /root/reference (build container only)")
def test_bundled_scene_counts():
    # SURVEY.md §6.1 [probed]: 2.scn -> 32 + 2 + 192 triangles, 451 boxes, 14 materials
    s = HostScene(1000, 1000).load_scn(os.path.join(REFERENCE, "resources/scenes/2.scn")).build_bvh(2)
    assert (len(s.objects), len(s.boxes), len(s.materials)) == (226, 451, 14)
    assert (s.objects["type"] == _capi.TRIANGLE).all()
    assert s.texture_info["width"][1] == 1024        # first texture that loads becomes the sky (copper.bmp)


def test_obj_route_equals_direct_fill(libs, tmp_path):
    """SURVEY.md H6 / BASELINE.json configs[2] ("1 M-tri OBJ"): the mesh written as OBJ text and read back through the
    MeshSerializer-equivalent loader (src/MeshSerializer.cpp:96-137,141-243: atof, usemtl by name, f a//a b//b c//c,
    v*100*aspect + (w/2,h/2,depth/2), position = centroid) gives byte-identical objects — and therefore the identical
    BVH — to the direct fill the large configurations use. (tests/test_gpu_configs.py repeats this at 1 M triangles.)"""
    a = synthetic_scene(60_000, 640, 360, seed=99, tex_size=16, nthreads=4)
    obj = str(tmp_path / "mesh.obj")
    b = synthetic_scene(60_000, 640, 360, seed=99, tex_size=16, nthreads=4, via_obj=obj)
    text = open(obj).read(4096)
    assert text.startswith("# synthetic") and "\nv " in text
    assert sum(1 for line in open(obj) if line.startswith("f ")) == 60_000
    assert sum(1 for line in open(obj) if line.startswith("usemtl ")) > 1
    assert a.objects.tobytes() == b.objects.tobytes()
    assert a.boxes.tobytes() == b.boxes.tobytes()
    assert a.materials.tobytes() == b.materials.tobytes()


def test_grid_replication_makes_explicit_copies(libs):
    """BASELINE.json configs[4]: a base mesh replicated on a grid into explicit triangles (no instancing in the reference)."""
    s = synthetic_scene(40_000, 640, 360, seed=5, tex_size=16, nthreads=4, grid=(5, 4))
    assert len(s.objects) == 40_000 + 32 + 2 and len(s.boxes) == 2 * len(s.objects) - 1
    tri = s.objects[34:]
    per = 2_000
    # every copy is the base mesh translated: same edge vectors up to rounding of the translation, different positions
    e0 = tri["vertices"][:per, 1, :3] - tri["vertices"][:per, 0, :3]
    e1 = tri["vertices"][per:2 * per, 1, :3] - tri["vertices"][per:2 * per, 0, :3]
    assert np.allclose(e0, e1, atol=1e-2) and not np.allclose(tri["position"][:per], tri["position"][per:2 * per])
