"""The oracle (oracle/rt_oracle.cpp) against (a) the part of the reference that
compiles standalone (oracle/_ref/libref_parts.so: Shading.h, Camera.h, Vector3.h,
Color.h, Random.cpp), bit for bit; (b) the reference outputs recorded in
BASELINE.md §2; (c) the committed golden fixtures."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle
from conftest import REFERENCE, golden, have_reference
from raytracer_amd import _capi

ref = oracle.ref_parts()
needs_ref = pytest.mark.skipif(ref is None, reason="oracle/_ref/libref_parts.so not built (needs /root/reference)")
rng = np.random.default_rng(1234)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def random_material(i):
    m = np.zeros(1, dtype=_capi.MATERIAL)
    r = np.random.default_rng(100 + i)
    for k in ("albedo", "tint", "specular", "transmissionFilter"):
        m[k] = np.append(r.random(3), 1.0)
    if i % 7 == 0:
        m["albedo"] = (0, 0, 0, 1)                  # black albedo: Tint() divides by zero luminance
    for k in ("specularIntensity", "transparency", "roughness", "metallic", "sheen", "tintRoughness",
              "clearcoatThickness", "clearcoatRoughness", "anisotropy"):
        m[k] = r.random()
    m["roughness"] = max(1e-6, float(m["roughness"][0])) if i % 5 else 0.0
    m["indexOfRefraction"] = 1.0 + 2.0 * r.random()
    m["emmissionIntensity"] = r.random() * (i % 3 == 0)
    return m


def unit(v):
    return (v / np.linalg.norm(v)).astype(np.float32)


@needs_ref
def test_rand_stream_bit_exact():
    L = oracle.lib()
    for seed0 in [0, 1, 12345, 0xFFFFFFFF, 0x80000000, 747796405]:
        a, b = C.c_uint32(seed0), C.c_uint32(seed0)
        for _ in range(2000):
            x, y = L.orc_rand(C.byref(a)), ref.ref_rand(C.byref(b))
            assert a.value == b.value and np.float32(x).view(np.uint32) == np.float32(y).view(np.uint32)


@needs_ref
def test_pixel_position_bit_exact():
    L = oracle.lib()
    for trial in range(50):
        cam = _capi.Camera()
        f = unit(rng.normal(size=3)); u = unit(np.cross(f, rng.normal(size=3))); r = unit(np.cross(u, f))
        for name, v in (("front", f), ("up", u), ("right", r)):
            setattr(cam, name, _capi.Vec4(*[float(x) for x in v], 0.0))
        cam.position = _capi.Vec4(*[float(x) for x in rng.normal(size=3) * 1000], 0.0)
        cam.aspectRatio = float(rng.uniform(0.5, 2.5)); cam.nearView = 0.1; cam.fov = float(rng.uniform(20, 100))
        w, h = int(rng.integers(32, 4000)), int(rng.integers(32, 3000))
        for _ in range(40):
            x, y = int(rng.integers(-1, w + 1)), int(rng.integers(-1, h + 1))
            a, b = np.zeros(4, np.float32), np.zeros(4, np.float32)
            L.orc_pixel_position(C.byref(cam), x, y, w, h, a.ctypes.data)
            ref.ref_pixel_position(C.byref(cam), x, y, w, h, b.ctypes.data)
            assert (bits(a)[:3] == bits(b)[:3]).all()


@needs_ref
def test_shading_terms_bit_exact():
    L = oracle.lib()
    n_checked = 0
    for i in range(60):
        m = random_material(i)
        for _ in range(50):
            n = unit(rng.normal(size=3)); l = unit(rng.normal(size=3)); v = unit(rng.normal(size=3)); hv = unit(l + v)
            c = np.array([max(1e-6, float(n @ v)), max(1e-6, float(n @ l)), max(1e-6, float(l @ hv))], np.float32)
            a, b = np.zeros(9, np.float32), np.zeros(9, np.float32)
            args = [m.ctypes.data, n.ctypes.data, l.ctypes.data, v.ctypes.data, hv.ctypes.data, c.ctypes.data]
            L.orc_shading_terms(*args, a.ctypes.data)
            ref.ref_shading_terms(*args, b.ctypes.data)
            assert (bits(a) == bits(b)).all(), (i, a, b)
            n_checked += 1
    assert n_checked == 3000


@needs_ref
def test_texture_fetch_bit_exact(scene1):
    L = oracle.lib()
    d = scene1.desc()
    info = scene1.texture_info
    objs = scene1.objects
    # the reference reads texels past the end without bounds (UB); give it a padded copy with the
    # zeros the oracle defines there
    tex = np.concatenate([scene1.texture_data, np.zeros(8194, np.uint32)])
    for t in range(len(info)):
        ti = info[t:t + 1]
        for _ in range(300):
            u, v = np.float32(rng.uniform(0, 2.0)), np.float32(rng.uniform(0, 1.0))
            a, b = np.zeros(4, np.float32), np.zeros(4, np.float32)
            L.orc_color_sample(C.byref(d), u, v, ti.ctypes.data, a.ctypes.data)
            ref.ref_color_sample(tex.ctypes.data, u, v, ti.ctypes.data, b.ctypes.data)
            assert (bits(a) == bits(b)).all()
        for u, v in [(1.0, 1.0), (0.0, 0.0), (2.0, 1.0), (1.0, 0.5)]:
            a, b = np.zeros(4, np.float32), np.zeros(4, np.float32)
            L.orc_color_sample(C.byref(d), np.float32(u), np.float32(v), ti.ctypes.data, a.ctypes.data)
            ref.ref_color_sample(tex.ctypes.data, np.float32(u), np.float32(v), ti.ctypes.data, b.ctypes.data)
            assert (bits(a) == bits(b)).all()
    for k in range(0, len(objs), 3):
        o = objs[k:k + 1]
        ti = info[1:2]
        for _ in range(20):
            if o["type"][0] == _capi.TRIANGLE:
                w = rng.dirichlet((1, 1, 1)); p = (w[:, None] * o["vertices"][0][:, :3]).sum(0).astype(np.float32)
            else:
                p = (o["position"][0][:3] + unit(rng.normal(size=3)) * o["radius"][0]).astype(np.float32)
            n = unit(rng.normal(size=3))
            a, b = np.zeros(4, np.float32), np.zeros(4, np.float32)
            L.orc_get_texture_pixel(C.byref(d), o.ctypes.data, ti.ctypes.data, p.ctypes.data, n.ctypes.data, a.ctypes.data)
            ref.ref_get_texture_pixel(tex.ctypes.data, o.ctypes.data, ti.ctypes.data, p.ctypes.data, n.ctypes.data, b.ctypes.data)
            assert (bits(a) == bits(b)).all()


def same_bits_or_both_nan(a, b):
    """Bit equality; two NaNs count as equal whatever their payload (the payload of x86's NaN results depends on
    the operand order the compiler picked for a commutative SSE instruction, not on the arithmetic)."""
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    return bool(((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))).all())


SPECIALS = np.array([0.0, -0.0, 1.0, -1.0, 0.5, 1.0000001, 0.99999994, 2.0, 1e-45, -1e-45, 1.1754942e-38, 1e-38, 3.4028235e38,
                     -3.4028235e38, np.inf, -np.inf, np.nan, 1e-6, -1e-6, 0.01, 1e20, 1e-20, 255.0, 3.1415926535], np.float32)


def primitive_cases():
    """Operand pairs for the Vector3 / Color primitives: random vectors at several magnitudes, every pair of special
    values broadcast into single lanes (NaN, +-inf, +-0, denormals, FLT_MAX), zero-length and huge vectors."""
    r = np.random.default_rng(77)
    cases = []
    for scale in (1.0, 1e-3, 1e3, 1e-20, 1e18):
        for _ in range(200):
            cases.append((r.normal(size=4).astype(np.float32) * np.float32(scale), r.normal(size=4).astype(np.float32) * np.float32(scale),
                          np.float32(r.uniform(-0.5, 1.5))))
    for i, x in enumerate(SPECIALS):
        for j, y in enumerate(SPECIALS):
            a = r.normal(size=4).astype(np.float32); b = r.normal(size=4).astype(np.float32)
            a[(i + j) % 4] = x; b[(i + j) % 4] = y
            cases.append((a, b, SPECIALS[(i * 7 + j) % len(SPECIALS)]))
            cases.append((np.full(4, x, np.float32), np.full(4, y, np.float32), y))
    z = np.zeros(4, np.float32)
    cases += [(z, z, np.float32(0.5)), (z, np.ones(4, np.float32), np.float32(np.nan)), (np.full(4, 1e30, np.float32), z, np.float32(2.0)),
              (np.array([1e-30, 0, 0, 5], np.float32), z, np.float32(-1.0)), (np.array([0, 0, 0, 7], np.float32), z, np.float32(1.0))]
    return cases


@needs_ref
def test_vector_and_color_primitives_bit_exact():
    """src/Vector3.h:34-52 (Dot, Cross, Lerp with its in-place clamp of t), :54-91 (SSE Minimal/Maximal/Clamp: the SECOND
    operand wins when either is NaN), :93-174 (+ - * / on all four lanes, / multiplies by 1.0f/s), :184-199 (Magnitude over
    xyz, Normalize scales all four lanes); src/Color.h:15-78 (+ - *), :95-98 (Lerp, std::fmax/fmin), :109-122 (Clamp).
    The oracle's add/sub/mul/divs/sse_min/sse_max/clamp01_sse/dot/cross/normalize/clamp01_libm+lerp_v/lerp_c against the
    reference's own code compiled where it lies (oracle/_ref/libref_parts.so). With this, every function of
    ThreadedShader.cpp that the oracle restates is a composition of reference-pinned primitives, libm calls and control flow."""
    L = oracle.lib()
    n = 0
    for a, b, t in primitive_cases():
        a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
        for fn, size in (("vec_ops", 32), ("vec_geom", 15), ("color_ops", 12)):
            got, want = np.zeros(size, np.float32), np.zeros(size, np.float32)
            getattr(L, "orc_" + fn)(a.ctypes.data, b.ctypes.data, C.c_float(float(t)), got.ctypes.data)
            getattr(ref, "ref_" + fn)(a.ctypes.data, b.ctypes.data, C.c_float(float(t)), want.ctypes.data)
            assert same_bits_or_both_nan(got, want), (fn, a, b, t, got, want)
        n += 1
    assert n > 2000


def test_sse_nan_rule_is_the_second_operand():
    """The rule itself, independent of the reference build: _mm_min_ps(a, b) = a < b ? a : b."""
    L = oracle.lib()
    a = np.array([np.nan, 1.0, np.nan, 2.0], np.float32); b = np.array([3.0, np.nan, np.nan, 1.0], np.float32)
    out = np.zeros(32, np.float32)
    L.orc_vec_ops(a.ctypes.data, b.ctypes.data, C.c_float(1.0), out.ctypes.data)
    mn, mx, cl = out[20:24], out[24:28], out[28:32]
    assert mn[0] == 3.0 and np.isnan(mn[1]) and np.isnan(mn[2]) and mn[3] == 1.0
    assert mx[0] == 3.0 and np.isnan(mx[1]) and np.isnan(mx[2]) and mx[3] == 2.0
    # Clamp(NaN) = max(min(NaN, 1), 0) = max(1, 0) = 1: how a degenerate triangle's texture coordinates become (1, 1)
    assert cl[0] == 1.0 and cl[1] == 1.0 and cl[2] == 1.0 and cl[3] == 1.0


@pytest.mark.skipif(not have_reference(), reason="needs /root/reference (build container only)")
@needs_ref
@pytest.mark.skipif(not have_reference(), reason="needs the bundled textures under /root/reference")
@pytest.mark.parametrize("name", ["dunes.bmp", "metal.bmp", "sand.bmp"])
def test_bitmap_ingest_matches_the_reference_reader(name, tmp_path):
    """Our .bmp ingest (host/ingest.cpp read_bmp, reached through a .mtl's map_Ka like
    src/MaterialSerializer.cpp:215-219) against the reference's own BitmapReader::ReadFile
    (src/BitmapReader.cpp:70-137, compiled into libref_parts.so): same size, same texel words."""
    from raytracer_amd.scene import HostScene
    bmp = os.path.join(REFERENCE, "resources/textures", name)
    w, h = C.c_uint32(0), C.c_uint32(0)
    cap = 2048 * 2048
    want = np.zeros(cap, dtype=np.uint32)
    n = ref.ref_read_bmp(bmp.encode(), C.byref(w), C.byref(h), want.ctypes.data, cap)
    assert 0 < n <= cap
    (tmp_path / "one.mtl").write_text(f"newmtl A\nmap_Ka {os.path.relpath(bmp, tmp_path)}\n")
    s = HostScene(128, 128).load_mtl(str(tmp_path / "one.mtl"))
    info = s.texture_info[-1]
    assert (int(info["width"]), int(info["height"])) == (w.value, h.value)
    got = s.texture_data[int(info["offset"]):int(info["offset"]) + n]
    assert np.array_equal(got, want[:n])


def test_recorded_reference_outputs_2scn():
    """BASELINE.md §2 / SURVEY.md §8c: the reference's own strict (-O2) build on
    resources/scenes/2.scn gave 550 050 primary hits of 984 064 pixels at 992x992, frame 0
    (the -Ofast build: 550 057), BVH and linear traversal agreeing on every pixel; the 1000x1000
    one-frame framebuffer summed to 1040022.3 (-Ofast: 1039975.7)."""
    from raytracer_amd.scene import HostScene
    s = HostScene(1000, 1000).load_scn(os.path.join(REFERENCE, "resources/scenes/2.scn")).build_bvh(4)
    d = s.desc()
    assert (d.width, d.height, d.numObjects, d.numBoxes, d.numMaterials) == (992, 992, 226, 451, 14)
    ids = oracle.primary_hits(d, s.camera, 0, mode=1)
    assert int((ids >= 0).sum()) == 550050
    assert (oracle.primary_hits(d, s.camera, 0, mode=2) == ids).all()
    s2 = HostScene(1000, 1000).set_resolution(1000, 1000).load_scn(os.path.join(REFERENCE, "resources/scenes/2.scn")).build_bvh(4)
    px = np.zeros((1000, 1000, 4), np.float32)
    oracle.render(s2.desc(), s2.camera, 0, px, 8)
    rgb_sum = float(px[..., :3].astype(np.float64).sum())
    # material 0 (the floor) has uninitialised fields in the reference (src/Configurator.cpp:24-36),
    # so only agreement to ~1e-6 relative can be expected; the -Ofast build is 46 away
    assert abs(rgb_sum - 1040022.3) < 1.0, rgb_sum


@pytest.mark.skipif(not have_reference(), reason="needs /root/reference (build container only)")
def test_recorded_reference_outputs_2scn_ofast_build():
    """The reference's second recorded pair: its as-shipped -Ofast build gave 550 057
    primary hits and a 1000x1000 one-frame RGB sum of 1039975.7. oracle/rt_oracle.cpp compiled with -Ofast
    (oracle/liboracle_ofast.so) reproduces both — so the one restatement, compiled two ways, lands on the reference's
    numbers for both of its builds. -Ofast turns Normalize's 1.0f/sqrtf into rsqrtss + a Newton step, whose bits depend on
    the CPU; on a host whose estimate differs from the survey's the test reports instead of failing."""
    from raytracer_amd.scene import HostScene
    L = oracle.lib_ofast()
    assert L is not None, "oracle/liboracle_ofast.so not built (make -C oracle)"
    scn = os.path.join(REFERENCE, "resources/scenes/2.scn")
    s = HostScene(1000, 1000).load_scn(scn).build_bvh(4)
    ids = oracle.primary_hits(s.desc(), s.camera, 0, mode=1, L=L)
    hits = int((ids >= 0).sum())
    s2 = HostScene(1000, 1000).set_resolution(1000, 1000).load_scn(scn).build_bvh(4)
    px = np.zeros((1000, 1000, 4), np.float32)
    oracle.render(s2.desc(), s2.camera, 0, px, 8, L=L)
    rgb_sum = float(px[..., :3].astype(np.float64).sum())
    if hits != 550057 or abs(rgb_sum - 1039975.7) >= 1.0:
        pytest.xfail(f"-Ofast build on this host: {hits} hits, sum {rgb_sum} (recorded on the survey host: 550057, 1039975.7)")
    # and the two builds really differ where the reference's two builds differed: a handful of silhouette pixels
    strict = oracle.primary_hits(s.desc(), s.camera, 0, mode=1)
    assert 0 < int((ids != strict).sum()) < 5000


@pytest.mark.skipif(not have_reference(), reason="needs /root/reference (build container only)")
def test_recorded_hashes_are_not_reproduced_by_any_obvious_convention():
    """BASELINE.md §2 also quotes hashes of the reference's ID map (07544e4368e87eca) and framebuffer (79a2746fda3ea8c6,
    -O2) without saying how they were computed. scripts/hash_probe.py tries FNV-1a-64 / FNV-1-64 / FNV-1a-32 over bytes and
    over words, for int32 / +1 / int64 / big-endian / hit-mask / int16 / float / column-major / bottom-up layouts: none
    matches (the framebuffer cannot: the floor's material 0 has uninitialised fields in the reference,
    src/Configurator.cpp:24-36). This test keeps that outcome honest: if a convention ever matches, it fails and the
    match should become a real pin. Until then those two quantities are PARITY UNPINNED beyond the hit count and the sum."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("hash_probe", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "hash_probe.py"))
    hp = importlib.util.module_from_spec(spec); spec.loader.exec_module(hp)
    from raytracer_amd.scene import HostScene
    s = HostScene(1000, 1000).load_scn(os.path.join(REFERENCE, "resources/scenes/2.scn")).build_bvh(4)
    ids = oracle.primary_hits(s.desc(), s.camera, 0, mode=1)
    assert hp.try_all(hp.id_layouts(ids), "07544e4368e87eca") == []


def test_static_bands_render_the_same_pixels(scene2):
    """orc_render_bands (ThreadedShader::Render's schedule, src/ThreadedShader.cpp:186-207) == the dynamic hand-out."""
    d, cam = scene2.desc(), scene2.camera
    a = np.zeros((d.height, d.width, 4), np.float32); b = np.zeros_like(a)
    sa = oracle.render(d, cam, 2, a, 4, stats=1)
    sb = oracle.render_bands(d, cam, 2, b, 3)
    assert (bits(a) == bits(b)).all() and sa.rays == sb.rays


@pytest.mark.parametrize("name", ["scene1", "scene2"])
def test_oracle_reproduces_golden(name, request):
    s = request.getfixturevalue(name)
    d, cam = s.desc(), s.camera
    assert (oracle.primary_hits(d, cam, 0) == golden(name + "_ids_f0.npy")).all()
    px = np.zeros((d.height, d.width, 4), np.float32)
    for f in range(4):
        oracle.render(d, cam, f, px, 4)
        if f == 0:
            assert (bits(px) == bits(golden(name + "_fb_f1.npy"))).all()
    assert (bits(px) == bits(golden(name + "_fb_f4.npy"))).all()


def test_bvh_and_linear_agree(scene1, scene2, synth_small):
    for s in (scene1, scene2, synth_small):
        d = s.desc()
        a = oracle.primary_hits(d, s.camera, 3, mode=1)
        b = oracle.primary_hits(d, s.camera, 3, mode=2)
        assert (a == b).all()


def test_render_is_independent_of_row_split(scene2):
    d, cam = scene2.desc(), scene2.camera
    a = np.zeros((d.height, d.width, 4), np.float32); b = np.zeros_like(a)
    oracle.render(d, cam, 5, a, 1)
    for y0 in range(0, d.height, 37):
        oracle.render(d, cam, 5, b, 3, rows=(y0, min(d.height, y0 + 37)))
    assert (bits(a) == bits(b)).all()


@needs_ref
def test_camera_look_at_rotate_and_orbit_basis_bit_exact():
    """rtxh_camera_look_at / rtxh_camera_rotate / the basis rtxh_camera_orbit builds (what cfg5's orbit loop rests on)
    against the reference's own Camera::LookAt / Camera::Rotate (src/Camera.h:26-55), compiled where it lies: all 96 bytes
    of the camera, bit for bit. (The orbit POSITION formula lives in src/Main.cpp:68-77, which cannot be compiled here;
    the basis for that position is Camera::LookAt's.)"""
    H = _capi.host_lib()
    r = np.random.default_rng(77)

    def same(a, b):
        # NaN lanes (looking straight up: normalize(0)) must be NaN in both; the sign / payload of a NaN is the compiler's choice
        fa, fb = np.frombuffer(bytes(a), np.float32), np.frombuffer(bytes(b), np.float32)
        na, nb = np.isnan(fa), np.isnan(fb)
        return (na == nb).all() and (fa.view(np.uint32)[~na] == fb.view(np.uint32)[~nb]).all()

    for trial in range(400):
        cam = _capi.Camera()
        cam.front = _capi.Vec4(0, 0, 1, 0); cam.up = _capi.Vec4(0, 1, 0, 0); cam.right = _capi.Vec4(1, 0, 0, 0)
        cam.position = _capi.Vec4(*[float(np.float32(x)) for x in r.normal(size=3) * 2000], 0.0)
        cam.movementSpeed, cam.rotationSpeed, cam.aspectRatio, cam.nearView, cam.farView, cam.fov = 500.0, float(np.float32(r.uniform(1, 200))), 1.0, 0.1, 100.0, 45.0
        cam.pitch, cam.yaw = float(np.float32(r.uniform(-120, 120))), float(np.float32(r.uniform(-400, 400)))
        tgt = (r.normal(size=3) * 500).astype(np.float32)
        if trial % 50 == 0:
            tgt = np.array([cam.position.x, cam.position.y + 10.0, cam.position.z], np.float32)     # looking straight up: worldUp x front = 0
        a, b = _capi.Camera.from_buffer_copy(cam), _capi.Camera.from_buffer_copy(cam)
        H.rtxh_camera_look_at(C.byref(a), tgt.ctypes.data_as(C.POINTER(C.c_float)))
        ref.ref_camera_look_at(C.byref(b), tgt.ctypes.data)
        assert same(a, b)
        ox, oy = float(r.normal() * 0.3), float(r.normal() * 0.3)
        H.rtxh_camera_rotate(C.byref(a), ox, oy)
        ref.ref_camera_rotate(C.byref(b), ox, oy)
        assert same(a, b)
        # the orbit step: position from src/Main.cpp:68-77, then LookAt(0, 0, 0)
        yaw, pitch = float(np.float32(0.1 * trial)), float(np.float32(0.1 * trial))
        H.rtxh_camera_orbit(C.byref(a), yaw, pitch, 5000.0)
        b.position = a.position
        zero = np.zeros(3, np.float32)
        ref.ref_camera_look_at(C.byref(b), zero.ctypes.data)
        assert same(a, b)


@needs_ref
def test_bounding_box_layout_and_expand_match_the_reference():
    """RtxBoundingBox (include/rtx.h) and the numpy mirror against the reference's own struct (src/BoundingBox.h:7-48):
    size, alignment, offsets (also static_asserted in oracle/ref_parts.cpp), the default box (-1 ids, +-inf corners) and
    Expand() — libm fmin / fmax, NaN rule included — against the box union our builders form."""
    lay = np.zeros(8, np.int32)
    ref.ref_box_layout(lay.ctypes.data)
    B = _capi.BOX
    assert list(lay) == [B.itemsize, B.fields["objectID"][1], B.fields["parentID"][1], B.fields["leftID"][1], B.fields["rightID"][1],
                         B.fields["min"][1], B.fields["max"][1], 64]
    r = np.random.default_rng(5)
    for trial in range(200):
        n = int(r.integers(0, 6))
        boxes = np.zeros(max(n, 1), dtype=B)
        lo = r.normal(size=(max(n, 1), 3)).astype(np.float32) * 100
        hi = lo + r.random((max(n, 1), 3)).astype(np.float32) * 50
        boxes["min"][:, :3] = lo; boxes["max"][:, :3] = hi
        out = np.zeros(1, dtype=B)
        area = ref.ref_box_expand(boxes.ctypes.data, n, out.ctypes.data)
        assert out["objectID"][0] == -1 and out["leftID"][0] == -1 and out["rightID"][0] == -1
        if n == 0:
            assert np.isposinf(out["min"][0][:3]).all() and np.isneginf(out["max"][0][:3]).all()
            continue
        assert (bits(out["min"][0][:3]) == bits(lo[:n].min(axis=0))).all() and (bits(out["max"][0][:3]) == bits(hi[:n].max(axis=0))).all()
        d = out["max"][0][:3] - out["min"][0][:3]
        want = np.float32(2.0) * ((d[0] * d[1] + d[0] * d[2]) + d[1] * d[2])
        assert np.float32(area).view(np.uint32) == np.float32(want).view(np.uint32)
