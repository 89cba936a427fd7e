"""Parity of the HIP backend (through the C-ABI in include/rtx.h) with the oracle
and the committed golden fixtures. Bars (BASELINE.json north_star):
  * primary-hit object ids: bit-exact;
  * framebuffer: the contract is max-abs error <= 2e-3 and PSNR >= 50 dB at equal spp;
    what is asserted is more: the oracle's image BIT FOR BIT (conftest.check_fb —
    the tolerance is only a fallback behind RTX_TEST_TOLERANCE_ONLY=1, and even then
    the number of pixels that differ is printed)."""
import os

import numpy as np
import pytest

import oracle
from conftest import check_fb, golden
from raytracer_amd import _capi
from raytracer_amd.renderer import HIPShader

pytestmark = pytest.mark.gpu

@pytest.mark.parametrize("name", ["scene1", "scene2"])
@pytest.mark.parametrize("traversal", [_capi.TRAVERSE_CULLED, _capi.TRAVERSE_EXACT, _capi.TRAVERSE_WIDE])
def test_golden_scene(name, traversal, request):
    s = request.getfixturevalue(name)
    r = HIPShader(s, traversal=traversal)
    ids = r.primary_hits(0)
    assert (ids == golden(name + "_ids_f0.npy")).all()
    px = np.zeros((r.height, r.width, 4), np.float32)
    for f in range(4):
        r.frame_counter = f
        r.Render(px)
        if f == 0:
            check_fb(px, golden(name + "_fb_f1.npy"))
    check_fb(px, golden(name + "_fb_f4.npy"))
    assert (r.download() == px).all()
    r.close()


def test_linear_traversal_matches_oracle(scene1):
    from raytracer_amd.scene import scene_from_arrays
    s = scene_from_arrays(scene1.width, scene1.height, scene1.objects, scene1.materials, scene1.boxes[:0],
                          scene1.texture_info, scene1.texture_data, scene1.camera, use_bvh=False)
    d, cam = s.desc(), s.camera
    assert d.useBVH == 0
    r = HIPShader(s)
    assert (r.primary_hits(2) == oracle.primary_hits(d, cam, 2)).all()
    px = np.zeros((d.height, d.width, 4), np.float32); want = np.zeros_like(px)
    for f in range(2):
        r.frame_counter = f
        r.Render(px)
        oracle.render(d, cam, f, want, 8)
    check_fb(px, want)
    r.close()


def test_synthetic_scene_ids_fb_and_ray_counts(synth_small):
    s = synth_small
    d, cam = s.desc(), s.camera
    r = HIPShader(s, count_stats=True)
    for fc in (0, 1, 77):
        assert (r.primary_hits(fc) == oracle.primary_hits(d, cam, fc)).all()
    r.reset_counters()
    px = np.zeros((d.height, d.width, 4), np.float32); want = np.zeros_like(px)
    rays = [0, 0, 0, 0]; hits = [0, 0, 0, 0]
    for f in range(3):
        r.frame_counter = f
        r.Render(px)
        st = oracle.render(d, cam, f, want, 8, stats=1)
        rays = [a + b for a, b in zip(rays, st.raysPerBounce)]
        hits = [a + b for a, b in zip(hits, st.hitsPerBounce)]
    check_fb(px, want)
    c = r.counters()
    assert list(c.raysPerBounce) == rays and list(c.hitsPerBounce) == hits and c.rays == sum(rays)
    assert c.frames == 3 and sum(c.innerVisits) > 0 and sum(c.leafTests) > 0
    r.close()


def test_culled_and_exact_traversal_agree_on_incoherent_rays(synth_small):
    s = synth_small
    d = s.desc()
    rng = np.random.default_rng(5)
    n = 200_000
    o = np.zeros((n, 4), np.float32); dirs = np.zeros((n, 4), np.float32)
    o[:, :3] = rng.uniform([-400, 0, -600], [1400, 1200, 700], size=(n, 3))
    v = rng.normal(size=(n, 3)); dirs[:, :3] = v / np.linalg.norm(v, axis=1, keepdims=True)
    # axis-parallel rays: zero direction components exercise the 0*inf slab path
    dirs[:1000, :3] = 0; dirs[:1000, 2] = 1
    dirs[1000:2000, 0] = 0
    v2 = dirs[1000:2000, :3]; dirs[1000:2000, :3] = v2 / np.linalg.norm(v2, axis=1, keepdims=True)
    want_ids, want_pts, want_nrm = oracle.trace_rays(d, o, dirs)
    for trav in (_capi.TRAVERSE_CULLED, _capi.TRAVERSE_EXACT, _capi.TRAVERSE_WIDE):
        r = HIPShader(s, traversal=trav)
        ids, pts, nrm = r.trace_rays(o, dirs)
        assert (ids == want_ids).all()
        hit = ids >= 0
        assert hit.sum() > 10_000
        assert (pts[hit, :3].view(np.uint32) == want_pts[hit, :3].view(np.uint32)).all()
        assert (nrm[hit, :3].view(np.uint32) == want_nrm[hit, :3].view(np.uint32)).all()
        r.close()


def test_running_mean_persists_and_resets(scene2):
    """pixels is a read-modify-write accumulator (src/ThreadedShader.cpp:179-180); frameCounter = 0
    makes the Lerp scale 1 (camera moves reset it, src/Main.cpp:75,103)."""
    d, cam = scene2.desc(), scene2.camera
    r = HIPShader(scene2)
    px = np.zeros((d.height, d.width, 4), np.float32); want = np.zeros_like(px)
    for fc in (0, 1, 2, 0, 1):
        r.frame_counter = fc
        r.Render(px)
        oracle.render(d, cam, fc, want, 8)
        check_fb(px, want)
    start = np.random.default_rng(1).random(px.shape, dtype=np.float32)
    r.upload(start); want = start.copy()
    r.frame_counter = 3
    r.Render(px); oracle.render(d, cam, 3, want, 8)
    check_fb(px, want)
    r.clear()
    assert (r.download() == 0).all()
    r.close()


def test_moving_camera(scene1):
    from raytracer_amd.scene import orbit
    d = scene1.desc()
    r = HIPShader(scene1)
    cam = scene1.camera
    px = np.zeros((d.height, d.width, 4), np.float32)
    # `pixels` persists across the frames (src/ThreadedShader.cpp:179-180): with frameCounter = 0 the Lerp scale is 1, and
    # old + (new - old) * 1 is the new colour only up to a rounding — so the oracle carries its image along as well
    want = np.zeros_like(px)
    for i in range(3):
        orbit(cam, 40.0 * i + 10, 5.0 * i, radius=2500.0)
        r.frame_counter = 0
        r.Render(px, cam)
        oracle.render(d, cam, 0, want, 8)
        check_fb(px, want)
        assert (r.primary_hits(0, cam) == oracle.primary_hits(d, cam, 0)).all()
    r.close()


def test_tile_partition_assembles_the_single_gpu_frame(scene2):
    """world_size = 3 on one device: each rank renders only its tiles; packing and
    unpacking them into rank 0's framebuffer reproduces the 1-rank frame."""
    import ctypes as C
    import torch
    d = scene2.desc()
    single = HIPShader(scene2)
    full = np.zeros((d.height, d.width, 4), np.float32)
    for f in range(2):
        single.frame_counter = f
        single.Render(full)
    single.close()
    ranks = [HIPShader(scene2, rank=k, world_size=3, tile=(32, 16)) for k in range(3)]
    assert sum(x.local_pixel_count() for x in ranks) == d.width * d.height
    partial = np.full_like(full, -7.0)
    for x in ranks:
        for f in range(2):
            x.frame_counter = f
            x.Render(partial)               # writes only that rank's tiles
    assert (partial == full).all()
    for k in (1, 2):
        buf = torch.zeros(int(ranks[k].packed_floats(k)), dtype=torch.float32, device="cuda")
        ranks[k].pack_tiles(buf.data_ptr())
        torch.cuda.synchronize()
        ranks[0].unpack_tiles(buf.data_ptr(), k)
    assert (ranks[0].download() == full).all()
    for x in ranks:
        x.close()


def test_rejects_bad_input(scene1):
    import ctypes as C
    L = _capi.hip_lib()
    d = scene1.desc()
    objs = np.array(scene1.objects, copy=True)
    objs["materialID"][3] = 999
    d.objects = objs.ctypes.data
    h = C.c_void_p()
    assert L.rtx_create(C.byref(d), None, C.byref(h)) != 0
    assert b"materialID" in L.rtx_last_error()
    d = scene1.desc()
    boxes = np.array(scene1.boxes, copy=True)
    boxes["leftID"][0] = 0                      # cycle
    d.boxes = boxes.ctypes.data
    assert L.rtx_create(C.byref(d), None, C.byref(h)) != 0
    cfg = _capi.Config(tileW=12, tileH=8)
    d = scene1.desc()
    assert L.rtx_create(C.byref(d), C.byref(cfg), C.byref(h)) != 0
    cfg = _capi.Config(numDevices=-1)                # the internal marker of a device-group member is not for callers
    assert L.rtx_create(C.byref(d), C.byref(cfg), C.byref(h)) != 0 and b"numDevices" in L.rtx_last_error()
    cfg = _capi.Config(device=63)                    # no such GPU
    assert L.rtx_create(C.byref(d), C.byref(cfg), C.byref(h)) != 0
    # a refused create leaves nothing behind (no sticky runtime error): the next renderer works
    r = HIPShader(scene1)
    px = np.zeros((r.height, r.width, 4), np.float32)
    r.Render(px)
    check_fb(px, golden("scene1_fb_f1.npy"))
    r.close()


def test_empty_scene_renders_sky(libs):
    from raytracer_amd.scene import HostScene, procedural_texture
    s = HostScene(64, 64)
    s.add_texture(procedural_texture(32, 9))
    d, cam = s.desc(), s.camera
    r = HIPShader(s)
    px = np.zeros((d.height, d.width, 4), np.float32); want = np.zeros_like(px)
    r.Render(px); oracle.render(d, cam, 0, want, 2)
    check_fb(px, want)
    assert (r.primary_hits(0) == -1).all()
    r.close()


def _tiny_scene(kind):
    from raytracer_amd.scene import HostScene, procedural_texture
    s = HostScene(160, 96)
    s.add_texture(procedural_texture(32, 3))
    s.add_texture(procedural_texture(32, 4))
    m = s.add_material("m", albedo=(0.8, 0.4, 0.2, 1.0), roughness=0.6, emmissionIntensity=0.5)
    if kind == "one_sphere":
        s.add_sphere((80, 48, 200), 60.0, m)
    elif kind == "one_plane":            # two triangles: a wide node with two empty slots
        s.add_plane((80, 20, 300), (0, 1, 0), (400, 400, 1), m)
    else:                                # 2 + 1 + 12 leaves: partially filled wide nodes at several depths
        s.add_plane((80, 10, 300), (0, 1, 0), (400, 400, 1), m)
        s.add_sphere((60, 50, 220), 30.0, m)
        s.add_cube((110, 40, 260), (30, 30, 30), m)
    s.build_bvh(2)
    cam = s.camera
    from raytracer_amd.scene import look_at
    cam.position.x, cam.position.y, cam.position.z = 80.0, 200.0, 0.0
    look_at(cam, (80.0, 20.0, 300.0))
    s.camera = cam
    return s


@pytest.mark.parametrize("kind", ["one_sphere", "one_plane", "mixed15"])
def test_tiny_trees(libs, kind):
    """One leaf (the wide tree is a single node holding it), two leaves, a handful: the 4-wide nodes'
    empty slots must never report a hit."""
    s = _tiny_scene(kind)
    d, cam = s.desc(), s.camera
    for trav in (_capi.TRAVERSE_CULLED, _capi.TRAVERSE_EXACT, _capi.TRAVERSE_WIDE):
        r = HIPShader(s, traversal=trav)
        ids = r.primary_hits(0)
        assert (ids == oracle.primary_hits(d, cam, 0)).all()
        assert (ids >= 0).any()
        px = np.zeros((d.height, d.width, 4), np.float32); want = np.zeros_like(px)
        for f in range(2):
            r.frame_counter = f
            r.Render(px)
            oracle.render(d, cam, f, want, 2)
        check_fb(px, want)
        r.close()


@pytest.mark.parametrize("copies", [6, 80])
def test_coincident_primitives_tie_break(libs, monkeypatch, copies):
    """Several copies of the same plane and the same sphere: every ray that hits them gets the SAME t from up to six leaves,
    and the reference's winner is decided by its visiting order alone (descending box id in the tree, ascending object id
    in the list; src/ThreadedShader.cpp:219-237, :370-392). In the default mode those equal candidates of one ray meet in
    one 64-bit minimum per set of leaves tested together (rtx_traverse.h: q_wave_serve) — ids and frames must still be the
    oracle's, with the leaves tested per wave and per lane (RTX_LEAFQ=0), with and without a tree. With 80 copies a single ray
    queues well over a hundred leaves: more than one set of 64 for one owner lane, several expansions per ray."""
    from raytracer_amd.scene import HostScene, procedural_texture, look_at
    def build(bvh):
        s = HostScene(192, 128)
        s.add_texture(procedural_texture(32, 5))
        s.add_texture(procedural_texture(32, 6))
        mats = [s.add_material(f"m{i}", albedo=(0.2 + 0.1 * i, 0.9 - 0.1 * i, 0.5, 1.0), roughness=0.3 + 0.1 * i,
                               emmissionIntensity=0.2 * i) for i in range(6)]
        for i in range(copies):
            s.add_plane((96, 10, 300), (0, 1, 0), (300, 300, 1), mats[i % 6])     # copies x 2 coincident triangles
        for i in range(3):
            s.add_sphere((70, 60, 240), 35.0, mats[5 - i])                          # 3 coincident spheres
        s.add_cube((130, 40, 280), (30, 30, 30), mats[2])
        s.add_cube((130, 40, 280), (30, 30, 30), mats[4])                           # and two coincident cubes
        if bvh:
            s.build_bvh(2)
        cam = s.camera
        cam.position.x, cam.position.y, cam.position.z = 96.0, 180.0, 0.0
        look_at(cam, (96.0, 10.0, 300.0))
        s.camera = cam
        return s
    for bvh in (True, False):
        s = build(bvh)
        d, cam = s.desc(), s.camera
        want_ids = oracle.primary_hits(d, cam, 0)
        assert len(np.unique(want_ids[want_ids >= 0])) >= 3
        want = np.zeros((d.height, d.width, 4), np.float32)
        for f in range(3):
            oracle.render(d, cam, f, want, 2)
        for trav, leafq in ((_capi.TRAVERSE_WIDE, "1"), (_capi.TRAVERSE_WIDE, "0"), (_capi.TRAVERSE_CULLED, "1"), (_capi.TRAVERSE_EXACT, "1")):
            monkeypatch.setenv("RTX_LEAFQ", leafq)
            r = HIPShader(s, traversal=trav)
            assert (r.primary_hits(0) == want_ids).all(), (bvh, trav, leafq)
            px = np.zeros_like(want)
            for f in range(3):
                r.frame_counter = f
                r.Render(px)
            check_fb(px, want)
            r.close()
    monkeypatch.delenv("RTX_LEAFQ")


def test_one_object_tree_intersects_without_testing_the_root_box(libs):
    """A BVH of one object: box 0 is the leaf, and the reference pops it and intersects the object without ever testing a
    box (src/ThreadedShader.cpp:349-368). A root leaf box that does not contain its sphere must therefore not hide it —
    in any traversal mode (rtx_create keeps such a tree on the reference topology)."""
    from raytracer_amd.scene import scene_from_arrays
    s = _tiny_scene("one_sphere")
    boxes = np.array(s.boxes, copy=True)
    assert len(boxes) == 1 and boxes["objectID"][0] == 0
    boxes["min"][0][:3] = (5000.0, 5000.0, 5000.0)         # nowhere near the sphere
    boxes["max"][0][:3] = (5001.0, 5001.0, 5001.0)
    s2 = scene_from_arrays(s.width, s.height, s.objects, s.materials, boxes, s.texture_info, s.texture_data, s.camera)
    d, cam = s2.desc(), s2.camera
    want_ids = oracle.primary_hits(d, cam, 0)
    assert (want_ids >= 0).any()
    want = np.zeros((d.height, d.width, 4), np.float32)
    oracle.render(d, cam, 0, want, 2)
    for trav in (_capi.TRAVERSE_WIDE, _capi.TRAVERSE_CULLED, _capi.TRAVERSE_EXACT):
        r = HIPShader(s2, traversal=trav)
        assert (r.primary_hits(0) == want_ids).all()
        px = np.zeros_like(want)
        r.Render(px)
        check_fb(px, want)
        r.close()


def test_float_node_fallback_of_the_default_mode(synth_small, monkeypatch):
    """Scenes beyond 2^27 - 4 leaves cannot use the quantised nodes (their work-list entries hold base << 4 in an int,
    rtx_build.h kQuantMaxLeaves) and walk the same tree with float nodes, unordered semantics kept (no distance culling).
    RTX_WIDE_ORDERED=1 forces that path on a small scene: same ids, same image, same leaf set as the quantised walk."""
    s = synth_small
    d, cam = s.desc(), s.camera
    out = {}
    for forced in (False, True):
        if forced:
            monkeypatch.setenv("RTX_WIDE_ORDERED", "1")
        r = HIPShader(s, traversal=_capi.TRAVERSE_WIDE, count_stats=True)
        info = r.info()
        assert info.traversal == _capi.TRAVERSE_WIDE and np.isinf(info.cullSlack)
        ids = r.primary_hits(0)
        px = np.zeros((d.height, d.width, 4), np.float32)
        r.reset_counters()
        for f in range(2):
            r.frame_counter = f
            r.Render(px)
        c = r.counters()
        out[forced] = (ids, px, list(c.leafTests), list(c.raysPerBounce))
        r.close()
    monkeypatch.delenv("RTX_WIDE_ORDERED")
    assert (out[True][0] == out[False][0]).all() and (out[True][0] == oracle.primary_hits(d, cam, 0)).all()
    assert (out[True][1].view(np.uint32) == out[False][1].view(np.uint32)).all()
    assert out[True][2:] == out[False][2:]


def test_boxes_that_break_the_fast_path_preconditions_still_match(scene1):
    """The culled path needs nested, well-formed boxes (DESIGN.md "Topology freedom"). A tree with an inner
    box that does not contain its children, and one with an inverted (min > max) leaf box, must be rendered
    exactly as the reference would render THOSE boxes: rtx_create falls back to the reference topology."""
    from raytracer_amd.scene import scene_from_arrays
    boxes0 = np.array(scene1.boxes, copy=True)
    inner = np.flatnonzero((boxes0["objectID"] == -1) & (np.arange(len(boxes0)) > 0))
    leaves = np.flatnonzero(boxes0["objectID"] != -1)
    variants = []
    b = boxes0.copy()                              # shrink a mid-tree inner box to a sliver of itself
    k = inner[len(inner) // 3]
    b["max"][k][:3] = b["min"][k][:3] + 0.25 * (b["max"][k][:3] - b["min"][k][:3])
    variants.append(b)
    b = boxes0.copy()                              # swap min and max of a leaf box on one axis
    k = leaves[len(leaves) // 2]
    b["min"][k][0], b["max"][k][0] = boxes0["max"][k][0], boxes0["min"][k][0]
    variants.append(b)
    for b in variants:
        s = scene_from_arrays(scene1.width, scene1.height, scene1.objects, scene1.materials, b,
                              scene1.texture_info, scene1.texture_data, scene1.camera)
        d, cam = s.desc(), s.camera
        r = HIPShader(s)                           # asks for the culled path
        assert (r.primary_hits(0) == oracle.primary_hits(d, cam, 0)).all()
        px = np.zeros((d.height, d.width, 4), np.float32); want = np.zeros_like(px)
        r.Render(px); oracle.render(d, cam, 0, want, 8)
        check_fb(px, want)
        r.close()


def test_wave_pass_counter_and_queue_holes(synth_small):
    """countStats: lane utilisation of the refill kernel is a sane fraction; the queue holes that k_shade's
    run reservation leaves are not counted as rays (ray counts equal the oracle's in the test above)."""
    r = HIPShader(synth_small, count_stats=True)
    for f in range(2):
        r.frame_counter = f
        r.Render(None)
    c = r.counters()
    for b in (1, 2, 3):
        assert c.raysPerBounce[b] == c.hitsPerBounce[b - 1]
        if c.raysPerBounce[b]:
            assert c.wavePasses[b] > 0
            work = c.innerVisits[b] + c.leafTests[b]
            assert 0.0 < work / (64.0 * c.wavePasses[b]) <= 2.0      # a lane can do a node step and a leaf test in one pass
    r.close()


def test_queues_grow_with_the_first_queued_frame(synth_small, monkeypatch):
    """rtx_create allocates ray queues and path state for the first sets of the synchronous loop only (4 frames per context;
    a context grows to 8 when a set of 8 lands on it); the first frame that is queued (Render without pixels) grows them to
    whole batches. A caller that only ever asks for delivered frames — the reference's loop — never holds the memory of a
    32-frame batch; results are those of a renderer that allocates everything at create."""
    d, cam = synth_small.desc(), synth_small.camera
    per_frame = d.width * d.height * 100          # at least this much per frame of a batch and context (about 125 B per pixel)
    r = HIPShader(synth_small)
    at_create = int(r.info().deviceBytes)
    assert r.info().framesPerBatch == 32
    px = np.zeros((d.height, d.width, 4), np.float32)
    for f in range(20):                           # delivered frames: the look-ahead runs its sets of 1, 2, 4, 8, 8 ... frames
        r.frame_counter = f
        r.Render(px)
    assert int(r.info().deviceBytes) - at_create <= 3 * (8 - 4) * d.width * d.height * 160, (at_create, int(r.info().deviceBytes))
    for f in range(100, 140):                     # queued frames — not the ones started ahead: one whole batch and a part of the next
        r.frame_counter = f
        r.Render(None)
    got = r.download()
    grown = int(r.info().deviceBytes)
    assert grown - at_create > 3 * (32 - 4) * per_frame // 2, (at_create, grown)
    c = r.counters()
    assert c.frames == 60
    r.close()
    monkeypatch.setenv("RTX_QUEUES_AT_CREATE", "1")
    r2 = HIPShader(synth_small)
    assert abs(int(r2.info().deviceBytes) - grown) < grown // 4        # (free-memory deltas: the runtime's allocator keeps some of what was freed)
    for f in list(range(20)) + list(range(100, 140)):
        r2.frame_counter = f
        r2.Render(None)
    check_fb(got, r2.download())
    assert list(r2.counters().raysPerBounce) == list(c.raysPerBounce)
    r2.close()


def test_frames_in_flight_keep_frame_order(scene2):
    """rtx_render(pixels = NULL) only queues the frame; several frames overlap on the device, each on its own
    stream with its own path state, while the running mean still takes them in order (k_resolve launches are
    chained by events). The result must be the serial one, bit for bit, and the oracle's."""
    d, cam = scene2.desc(), scene2.camera
    frames = (0, 1, 2, 3, 4, 5, 6, 0, 1, 2, 3)          # includes a restart of the mean (frameCounter = 0)
    out = {}
    for fif in (1, 3, 14):                              # 14: more streams than hardware queues, more contexts than frames
        r = HIPShader(scene2, frames_in_flight=fif)
        for fc in frames:
            r.frame_counter = fc
            r.Render(None)
        out[fif] = r.download()
        c = r.counters()
        assert c.frames == len(frames) and c.raysPerBounce[0] == len(frames) * d.width * d.height
        # mixing queued and synchronous calls
        px = np.zeros((d.height, d.width, 4), np.float32)
        r.frame_counter = 4
        r.Render(None); r.frame_counter = 5; r.Render(px)
        out[(fif, "sync")] = px
        r.close()
    assert (out[1] == out[3]).all() and (out[(1, "sync")] == out[(3, "sync")]).all()
    assert (out[1] == out[14]).all() and (out[(1, "sync")] == out[(14, "sync")]).all()
    want = np.zeros_like(out[1])
    for fc in frames:
        oracle.render(d, cam, fc, want, 8)
    check_fb(out[3], want)


def test_frame_batches_render_the_serial_frames(scene2, synth_small, monkeypatch):
    """A rank that owns 1/N of the tiles collects frames queued with pixels = NULL and renders up to N (at most
    8) of them with one set of launches (DevBatch). The frames still enter the running mean in call order, each with
    its own camera and frameCounter: the image must be the one the frame-by-frame path produces, bit for bit —
    with full batches, a partial batch at the end, a restart of the mean and a camera change inside a batch."""
    from raytracer_amd.scene import orbit
    for scene, world, rank, forced in ((scene2, 4, 1, None), (synth_small, 8, 3, None), (scene2, 1, 0, "3"), (synth_small, 1, 0, None)):     # None: the default (8 on a single rank)
        d = scene.desc()
        frames = (0, 1, 2, 3, 4, 0, 1, 2, 3, 4, 5)           # 11 frames: partial last batch whatever the batch size
        out = {}
        for batch in ("1", forced):
            if batch is None:
                monkeypatch.delenv("RTX_BATCH", raising=False)
            else:
                monkeypatch.setenv("RTX_BATCH", batch)
            r = HIPShader(scene, rank=rank, world_size=world, frames_in_flight=3)
            cam = type(scene.camera).from_buffer_copy(scene.camera)
            for i, fc in enumerate(frames):
                if i == 6:
                    orbit(cam, 20.0, 5.0)                       # a different camera from the 7th frame on
                r.frame_counter = fc
                r.Render(None, cam)
            c = r.counters()
            assert c.frames == len(frames) and c.raysPerBounce[0] == len(frames) * r.local_pixel_count()
            out[batch] = r.download()
            # a synchronous call right after queued ones
            r.frame_counter = 6; r.Render(None, cam)
            px = np.zeros((d.height, d.width, 4), np.float32)
            r.frame_counter = 7; r.Render(px, cam)
            out[(batch, "sync")] = px
            r.close()
        assert (out["1"] == out[forced]).all() and (out[("1", "sync")] == out[(forced, "sync")]).all()
    monkeypatch.delenv("RTX_BATCH", raising=False)


def test_wide_walk_tests_exactly_the_reference_leaf_set(scene1, synth_small):
    """The default traversal (own 4-wide tree, byte boxes, no order, no culling) must intersect a primitive iff the
    reference does: a leaf is tested iff its own box passes. Counted on the device: the number of
    primitive intersections per bounce equals that of the reference-topology walk, and the total equals the oracle's
    count (src/ThreadedShader.cpp:362-368 executions) for the same frames."""
    for s in (scene1, synth_small):
        d, cam = s.desc(), s.camera
        counts = {}
        for name, trav in (("wide", _capi.TRAVERSE_WIDE), ("exact", _capi.TRAVERSE_EXACT)):
            r = HIPShader(s, traversal=trav, count_stats=True)
            for f in range(3):
                r.frame_counter = f
                r.Render(None)
            c = r.counters()
            counts[name] = (list(c.leafTests), list(c.raysPerBounce), list(c.hitsPerBounce))
            r.close()
        assert counts["wide"] == counts["exact"]
        px = np.zeros((d.height, d.width, 4), np.float32)
        leaf = 0
        for f in range(3):
            leaf += oracle.render(d, cam, f, px, 8, stats=2).leafTests
        assert sum(counts["wide"][0]) == leaf


def test_look_ahead_of_the_synchronous_loop_is_invisible(scene2, monkeypatch):
    """rtx_render(pixels) starts the frames it expects next (same camera, frameCounter + 1, + 2, ...) before it copies the
    current one out (rtx_api.hip RtxRenderer::Ahead). Whatever the caller does next — the expected frame, another frame
    counter, another camera, queued frames, upload / clear / probes in between — images and work counters must be those of
    a renderer that never looks ahead, bit for bit, and the oracle's."""
    from raytracer_amd.scene import orbit
    d, cam = scene2.desc(), scene2.camera
    cam2 = type(cam).from_buffer_copy(cam); orbit(cam2, 30.0, 10.0, radius=2500.0)
    start = np.random.default_rng(5).random((d.height, d.width, 4), dtype=np.float32)
    # (op, argument): r = synchronous Render(pixels), q = queued Render(NULL), the rest are the calls that may come in between
    script = [("r", (cam, 0)), ("r", (cam, 1)), ("r", (cam, 2)), ("r", (cam, 3)),          # the loop: look-ahead kicks in and is used
              ("r", (cam, 7)),                                                               # a jump: what was started is dropped
              ("r", (cam, 8)), ("r", (cam, 9)), ("r", (cam2, 10)),                           # another camera
              ("r", (cam2, 11)), ("r", (cam2, 12)), ("upload", start), ("r", (cam2, 13)),    # the accumulator is replaced while frames are ahead
              ("r", (cam2, 14)), ("q", (cam2, 15)), ("q", (cam2, 16)), ("r", (cam2, 17)),    # queued frames in between
              ("r", (cam2, 18)), ("ids", (cam, 3)), ("r", (cam2, 19)), ("r", (cam2, 20)),
              ("p", (cam2, 21)), ("p", (cam2, 22)), ("p", (cam2, 23)), ("p", (cam2, 24)), ("d", (cam2, 25)), ("d", (cam2, 26)),   # the loop with present / download as the delivery
              ("d", (cam2, 27)), ("p", (cam, 28)), ("r", (cam, 29)), ("d", (cam, 30)), ("q", (cam, 31)), ("p", (cam, 32)),
              ("clear", None), ("r", (cam, 0)), ("r", (cam, 1)), ("r", (cam, 2)), ("counters", None), ("r", (cam, 3))]
    runs = {}
    for ahead in ("0", "2"):
        monkeypatch.setenv("RTX_LOOKAHEAD", ahead)
        r = HIPShader(scene2)
        px = np.zeros((d.height, d.width, 4), np.float32)
        log = []
        for op, arg in script:
            if op in ("r", "q", "p", "d"):
                r.frame_counter = arg[1]
                r.Render(px if op == "r" else None, arg[0])
                if op == "r": log.append(px.copy())
                if op == "p": log.append(r.present(2.2).view(np.uint32).copy())
                if op == "d": log.append(r.download())
            elif op == "upload": r.upload(arg)
            elif op == "clear": r.clear()
            elif op == "ids": log.append(r.primary_hits(arg[1], arg[0]).copy())
            elif op == "counters":
                c = r.counters(); log.append(np.array([c.frames] + list(c.raysPerBounce) + list(c.hitsPerBounce), np.int64))
        c = r.counters()
        log.append(np.array([c.frames] + list(c.raysPerBounce) + list(c.hitsPerBounce), np.int64))
        runs[ahead] = log
        r.close()
    assert len(runs["0"]) == len(runs["2"])
    for a, b in zip(runs["0"], runs["2"]):
        assert a.dtype == b.dtype and (a.view(np.uint32) == b.view(np.uint32)).all()
    frames = sum(1 for op, _ in script if op in ("r", "q", "p", "d"))
    assert runs["2"][-1][0] == frames and runs["2"][-1][1] == frames * d.width * d.height      # frames dropped unused are not counted
    # and the oracle's image for the last stretch (after the clear)
    want = np.zeros((d.height, d.width, 4), np.float32)
    for fc in range(4):
        oracle.render(d, cam, fc, want, 8)
    check_fb(runs["2"][-2], want)


def test_sets_started_for_nothing_are_cut_short_and_nobody_notices(monkeypatch):
    """A caller that leaves the loop while sets of frames are ahead: the library overwrites the work counters of those sets
    from a control stream so that their launches run out at once (rtx_api.hip abort_ahead; RtxInfo.setsCutShort counts them).
    At 1920x1088 the sets are in the middle of their launches when the jump comes. Images and counters must be those of a
    renderer that lets the sets run out (RTX_NO_ABORT=1), bit for bit — also for the frames that follow on the very contexts
    whose sets were cut short (their next launches wait for the abort to land before they clear the counters) — and the
    oracle's."""
    from raytracer_amd.scene import synthetic_scene
    s = synthetic_scene(200_000, 1920, 1080, seed=11, tex_size=64, nthreads=8)
    d, cam = s.desc(), s.camera
    # loop, jump, loop again at once (the contexts of the dropped sets are reused straight away), a queued batch, jump back
    script = [("r", fc) for fc in range(10)] + [("r", 500)] + [("r", fc) for fc in range(501, 512)] + [("r", 40)] + \
             [("q", fc) for fc in range(41, 47)] + [("r", 47), ("r", 48), ("r", 49), ("r", 3)]
    runs = {}
    for no_abort in (False, True):
        if no_abort: monkeypatch.setenv("RTX_NO_ABORT", "1")
        else: monkeypatch.delenv("RTX_NO_ABORT", raising=False)
        r = HIPShader(s)
        px = np.zeros((d.height, d.width, 4), np.float32)
        log = []
        for op, fc in script:
            r.frame_counter = fc
            r.Render(px if op == "r" else None)
            if op == "r" and fc in (2, 9, 500, 501, 511, 40, 47, 49, 3): log.append(px.copy())
        cut = r.info().setsCutShort
        c = r.counters()
        log.append(np.array([c.frames] + list(c.raysPerBounce) + list(c.hitsPerBounce), np.int64))
        runs[no_abort] = (log, cut)
        r.close()
    monkeypatch.delenv("RTX_NO_ABORT", raising=False)
    assert runs[True][1] == 0 and runs[False][1] >= 3, (runs[True][1], runs[False][1])      # three jumps, at least one untouched set each
    for a, b in zip(runs[False][0], runs[True][0]):
        assert (a.view(np.uint32) == b.view(np.uint32)).all()
    assert runs[False][0][-1][0] == len(script)
    want = np.zeros((d.height, d.width, 4), np.float32)
    for fc in range(3):
        oracle.render(d, cam, fc, want, 8)
    check_fb(runs[False][0][0], want)


def test_page_locked_destination_takes_the_frame_directly():
    """A caller's array that is page-locked (here: a pinned torch tensor seen through numpy) receives the frame by one
    direct device-to-host copy instead of the staged copy for pageable memory (rtx_api.hip deliver_bytes) — same bytes,
    through the synchronous loop with its look-ahead, through rtx_download_pixels, as a device group, and into memory from
    rtx_alloc_host."""
    import torch
    from raytracer_amd.scene import synthetic_scene
    s = synthetic_scene(20_000, 1280, 704, seed=3, tex_size=64, nthreads=8)       # 14.4 MB per frame: above the staging threshold
    d = s.desc()
    pinned = torch.empty((d.height, d.width, 4), dtype=torch.float32).pin_memory()
    assert pinned.is_pinned()
    out = {}
    # ... and an array from the library's own rtx_alloc_host (what a maintainer would put in place of `new Color[w * h]`)
    L = _capi.hip_lib()
    nbytes = d.height * d.width * 16
    mem = L.rtx_alloc_host(nbytes)
    assert mem, "rtx_alloc_host failed"
    import ctypes
    own = np.ctypeslib.as_array(ctypes.cast(mem, ctypes.POINTER(ctypes.c_float)), shape=(d.height, d.width, 4))
    for kind in ("pageable", "pinned", "pinned group", "rtx_alloc_host"):
        r = HIPShader(s, devices=[0, 0]) if kind.endswith("group") else HIPShader(s)
        px = np.zeros((d.height, d.width, 4), np.float32) if kind == "pageable" else (own if kind == "rtx_alloc_host" else pinned.numpy())
        px[...] = 0
        log = []
        for fc in range(6):
            r.frame_counter = fc; r.Render(px)
            log.append(px.copy())
        r.frame_counter = 6; r.Render(None)
        r.download_into(px); log.append(px.copy())
        out[kind] = log
        r.close()
    del own
    L.rtx_free_host(mem)
    for kind in ("pinned", "pinned group", "rtx_alloc_host"):
        for a, b in zip(out["pageable"], out[kind]):
            assert (a.view(np.uint32) == b.view(np.uint32)).all(), kind
    want = np.zeros((d.height, d.width, 4), np.float32)
    for fc in range(2):
        oracle.render(d, s.camera, fc, want, 8)
    check_fb(out["pinned"][1], want)


def test_config_knobs_are_reported_and_do_not_change_results(scene1):
    """RtxConfig.framesPerBatch / lookAhead / framesInFlight as resolved by rtx_create (rtx_get_info), and the same image
    whatever they are."""
    d = scene1.desc()
    out = []
    for kw, want in ((dict(), dict(framesPerBatch=32, lookAhead=2, framesInFlight=3)),
                     (dict(frames_per_batch=1, look_ahead=-1, frames_in_flight=1), dict(framesPerBatch=1, lookAhead=0, framesInFlight=1)),
                     (dict(frames_per_batch=3, look_ahead=1, frames_in_flight=2), dict(framesPerBatch=3, lookAhead=1, framesInFlight=2)),
                     (dict(frames_per_batch=100, look_ahead=9, frames_in_flight=4), dict(framesPerBatch=32, lookAhead=3, framesInFlight=4))):
        r = HIPShader(scene1, **kw)
        i = r.info()
        if os.environ.get("GPU_MAX_HW_QUEUES") in (None, "4") or kw:       # the defaults depend on the hardware queues of the process
            for k, v in want.items():
                assert getattr(i, k) == v, (kw, k, getattr(i, k), v)
        px = np.zeros((d.height, d.width, 4), np.float32)
        for f in range(5):
            r.frame_counter = f; r.Render(None)
        for f in range(5, 9):
            r.frame_counter = f; r.Render(px)
        out.append(px)
        r.close()
    for px in out[1:]:
        assert (px.view(np.uint32) == out[0].view(np.uint32)).all()
