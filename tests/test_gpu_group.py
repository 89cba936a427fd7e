"""Device groups (RtxConfig.numDevices > 1): one handle, one process, several GPUs — the multi-GPU path a
HIPShader : ComputeShader inside the reference takes, with Main.cpp untouched. Rehearsed here with several members on
the one GPU of the test box: the group must deliver the single-renderer frame bit for bit."""
import os
import subprocess

import numpy as np
import pytest

import oracle
from conftest import check_fb
from raytracer_amd import _capi
from raytracer_amd.renderer import HIPShader
from raytracer_amd.scene import orbit

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
RUN = os.path.join(os.path.dirname(HERE), "raytracer_amd", "rtx_run")


@pytest.mark.parametrize("members", [2, 3, 8])
def test_group_delivers_the_single_renderer_frame(scene2, synth_small, members):
    for scene in (scene2, synth_small):
        d = scene.desc()
        one = HIPShader(scene, count_stats=True)
        grp = HIPShader(scene, devices=[0] * members, count_stats=True)
        assert grp.info().devices == members and one.info().devices == 1
        assert grp.local_pixel_count() == d.width * d.height
        cam = type(scene.camera).from_buffer_copy(scene.camera)
        a = np.zeros((d.height, d.width, 4), np.float32); b = np.full_like(a, -3.0)
        for i, fc in enumerate((0, 1, 2, 3, 0, 1, 2)):            # queued frames, a restart of the mean, a camera change
            if i == 4:
                orbit(cam, 30.0, 4.0, radius=2500.0)
            for r in (one, grp):
                r.frame_counter = fc
                r.Render(None, cam)
        one.frame_counter = grp.frame_counter = 3
        one.Render(a, cam); grp.Render(b, cam)                    # delivery: gather + one copy
        assert (a.view(np.uint32) == b.view(np.uint32)).all()
        assert (grp.download().view(np.uint32) == a.view(np.uint32)).all()
        assert (grp.primary_hits(5, cam) == one.primary_hits(5, cam)).all()
        ca, cb = one.counters(), grp.counters()
        assert list(ca.raysPerBounce) == list(cb.raysPerBounce) and list(ca.hitsPerBounce) == list(cb.hitsPerBounce)
        assert ca.frames == cb.frames and list(ca.leafTests) == list(cb.leafTests)
        # the running mean can be replaced and cleared through the group handle
        start = np.random.default_rng(3).random(a.shape, dtype=np.float32)
        for r in (one, grp):
            r.upload(start); r.frame_counter = 2; r.Render(None, cam)
        assert (grp.download().view(np.uint32) == one.download().view(np.uint32)).all()
        grp.clear()
        assert (grp.download() == 0).all()
        one.close(); grp.close()


def test_group_matches_the_oracle(scene1):
    d, cam = scene1.desc(), scene1.camera
    grp = HIPShader(scene1, devices=[0, 0, 0, 0])
    px = np.zeros((d.height, d.width, 4), np.float32); want = np.zeros_like(px)
    for f in range(3):
        grp.frame_counter = f
        grp.Render(px)
        oracle.render(d, cam, f, want, 8)
    check_fb(px, want)
    assert (grp.primary_hits(0) == oracle.primary_hits(d, cam, 0)).all()
    grp.close()


def test_group_looks_ahead_in_step_on_all_members(scene2):
    """The synchronous loop on a device group: frames are started ahead on every member (rtx_api.hip rtx_render, group branch)
    and taken, dropped or topped up in step. Every delivered frame and the counters must be those of a single renderer
    that never looks ahead, bit for bit — through the loop, a jump of the frame counter, a camera change, queued frames,
    a download and an upload in between."""
    from raytracer_amd.scene import orbit
    d, cam = scene2.desc(), scene2.camera
    cam2 = type(cam).from_buffer_copy(cam); orbit(cam2, 25.0, 8.0, radius=2500.0)
    start = np.random.default_rng(9).random((d.height, d.width, 4), dtype=np.float32)
    script = [("r", (cam, 0)), ("r", (cam, 1)), ("r", (cam, 2)), ("r", (cam, 3)), ("r", (cam, 4)), ("r", (cam, 9)), ("r", (cam, 10)),
              ("r", (cam2, 11)), ("r", (cam2, 12)), ("r", (cam2, 13)), ("q", (cam2, 14)), ("q", (cam2, 15)), ("r", (cam2, 16)), ("r", (cam2, 17)),
              ("d", None), ("r", (cam2, 18)), ("r", (cam2, 19)), ("upload", start), ("r", (cam2, 20)), ("r", (cam2, 21)), ("r", (cam2, 22))]
    logs = {}
    for name, kw in (("one", dict(look_ahead=-1)), ("group", dict(devices=[0, 0, 0]))):
        r = HIPShader(scene2, **kw)
        if name == "group": assert r.info().lookAhead >= 1 and r.info().devices == 3
        px = np.zeros((d.height, d.width, 4), np.float32)
        log = []
        for op, arg in script:
            if op in ("r", "q"):
                r.frame_counter = arg[1]
                r.Render(px if op == "r" else None, arg[0])
                if op == "r": log.append(px.copy())
            elif op == "d": log.append(r.download())
            elif op == "upload": r.upload(arg)
        c = r.counters()
        log.append(np.array(list(c.raysPerBounce) + list(c.hitsPerBounce), np.int64))
        logs[name] = log
        r.close()
    for a, b in zip(logs["one"], logs["group"]):
        assert (a.view(np.uint32) == b.view(np.uint32)).all()


def test_group_rejects_bad_configurations(scene1):
    import ctypes as C
    L = _capi.hip_lib()
    d = scene1.desc()
    h = C.c_void_p()
    cfg = _capi.Config(numDevices=2, worldSize=2, rank=0)
    assert L.rtx_create(C.byref(d), C.byref(cfg), C.byref(h)) != 0 and b"device group" in L.rtx_last_error()
    cfg = _capi.Config(numDevices=2)
    cfg.devices[1] = 99
    assert L.rtx_create(C.byref(d), C.byref(cfg), C.byref(h)) != 0 and b"no such device" in L.rtx_last_error()
    cfg = _capi.Config(numDevices=9)
    assert L.rtx_create(C.byref(d), C.byref(cfg), C.byref(h)) != 0


def test_rtx_run_on_a_device_group(tmp_path):
    """rtx_run -G 2: HIPShader::Render(Color*) (the C++ class of INTEGRATION.md) over a two-member group writes the same
    image as the single-GPU run."""
    if not os.path.exists(RUN):
        pytest.fail("raytracer_amd/rtx_run is not built")
    imgs = []
    for extra in ([], ["-G", "2"], ["-G", "5"]):
        out = subprocess.run([RUN, "-N", "20000", "-w", "320", "-h", "200", "-B", "-F", "4", "-o", "f.pfm"] + extra,
                             cwd=tmp_path, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr + out.stdout
        assert ("Device group" in out.stdout) == bool(extra)
        imgs.append((tmp_path / "f.pfm").read_bytes())
    assert imgs[0] == imgs[1] == imgs[2]
