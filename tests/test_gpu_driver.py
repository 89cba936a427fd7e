"""The headless driver raytracer_amd/rtx_run: the reference's flags (src/Configurator.cpp:78-142) with HIPShader in
the place of ThreadedShader (C++ mirror of the reference's render-service interface over the C ABI)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
RUN = os.path.join(os.path.dirname(HERE), "raytracer_amd", "rtx_run")


def read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = (int(v) for v in f.readline().split())
        assert float(f.readline()) < 0          # little endian
        return np.frombuffer(f.read(), dtype="<f4").reshape(h, w, 3)


def test_headless_driver_renders_and_logs(tmp_path):
    if not os.path.exists(RUN):
        pytest.fail("raytracer_amd/rtx_run is not built (python -c 'import __graft_entry__ as g; g.build()')")
    out = subprocess.run([RUN, "-N", "3000", "-w", "200", "-h", "130", "-B", "-F", "3", "-o", "frame.pfm"],
                         cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "Resolution : 192 x 128" in out.stdout          # rounded to multiples of 32 (src/Configurator.cpp:144-145)
    img = read_pfm(tmp_path / "frame.pfm")
    assert img.shape == (128, 192, 3) and np.isfinite(img).all() and img.max() <= 1.0 and img.min() >= 0.0 and img.std() > 0.01
    log = (tmp_path / "Performance_log.csv").read_text().splitlines()
    assert log[0] == "fps;frametime(ms)" and len(log) == 4   # src/PerformanceMonitor.cpp:10,27
    # the same run through the BMP writer with display gamma
    out = subprocess.run([RUN, "-N", "3000", "-w", "200", "-h", "130", "-B", "-F", "3", "-o", "frame.bmp", "-g", "2.2"],
                         cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    raw = (tmp_path / "frame.bmp").read_bytes()
    assert raw[:2] == b"BM" and len(raw) == 54 + 192 * 3 * 128
    bgr = np.frombuffer(raw[54:], dtype=np.uint8).reshape(128, 192, 3)
    want = np.clip(img, 0, 1) ** (1 / 2.2) * 255.0 + 0.5
    assert np.abs(bgr[:, :, ::-1].astype(np.float32) - np.floor(want)).max() <= 1.0


@pytest.mark.parametrize("name", ["scene1", "scene2"])
def test_cxx_boundary_renders_the_oracles_image(tmp_path, name, request):
    """The C++ face of the boundary against the oracle: rtx_run constructs rtxhost::HIPShader over a RenderingContext
    (src/IFrameRender.h:6-9, src/ComputeShader.h:7-18) filled from the committed scene blob and runs the reference's
    bounded loop — Render(pixels); frameCounter++ (src/Main.cpp:46-54, src/WindowManager.cpp:154-158) — for 4 frames;
    the PFM it writes from `pixels` must hold the oracle's running mean bit for bit (RGB; a PFM has no alpha). Once
    more as a device group of two members (RtxConfig.numDevices, -G 2), the route by which the reference's untouched
    Main.cpp would drive several GPUs."""
    import oracle
    from conftest import GOLDEN, check_fb
    s = request.getfixturevalue(name)
    d, cam = s.desc(), s.camera
    want = np.zeros((d.height, d.width, 4), np.float32)
    for f in range(4):
        oracle.render(d, cam, f, want, 8)
    for extra in ([], ["-G", "2"]):
        out = subprocess.run([RUN, "-L", os.path.join(GOLDEN, name + ".blob"), "-F", "4", "-o", "frame.pfm"] + extra,
                             cwd=tmp_path, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        assert f"Resolution : {d.width} x {d.height}" in out.stdout
        img = read_pfm(tmp_path / "frame.pfm")
        check_fb(img, want[..., :3])
        os.remove(tmp_path / "frame.pfm")


def test_present_path_delivers_the_8bit_image(scene2):
    """rtx_present_rgba8: what WindowManager::UpdateWindow hands to the display (src/WindowManager.cpp:160-166, a GL_RGBA
    texture fed from the float pixels: clamp to [0,1], round to 8 bits), converted on the device."""
    from raytracer_amd.renderer import HIPShader
    r = HIPShader(scene2)
    for f in range(3):
        r.frame_counter = f
        r.Render(None)
    img = r.present()
    px = r.download()
    want = np.floor(np.clip(px, 0.0, 1.0) * np.float32(255.0) + np.float32(0.5)).astype(np.uint8)
    assert img.shape == px.shape and (img == want).all()
    g = r.present(gamma=2.2)
    wantg = np.clip(px, 0.0, 1.0).astype(np.float64)
    wantg[..., :3] = wantg[..., :3] ** (1 / 2.2)
    assert np.abs(g.astype(np.int32) - np.floor(wantg * 255.0 + 0.5).astype(np.int32)).max() <= 1     # device powf vs numpy
    assert (g[..., 3] == want[..., 3]).all()
    # values outside [0,1] and NaN in the running mean
    odd = np.zeros_like(px); odd[0, 0] = (-1.0, 2.0, np.nan, 0.5)
    r.upload(odd)
    assert tuple(r.present()[0, 0]) == (0, 255, 0, 128)
    r.close()


def test_headless_present_loop(tmp_path):
    out = subprocess.run([RUN, "-N", "3000", "-w", "200", "-h", "130", "-B", "-F", "3", "-P", "-g", "2.2", "-o", "shot.bmp"],
                         cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    raw = (tmp_path / "shot.bmp").read_bytes()
    assert raw[:2] == b"BM" and len(raw) == 54 + 192 * 128 * 4 and raw[28] == 32
    assert len((tmp_path / "Performance_log.csv").read_text().splitlines()) == 4
    bgra = np.frombuffer(raw[54:], dtype=np.uint8).reshape(128, 192, 4)
    assert bgra[..., :3].std() > 2
