"""The C-ABI libraries load and export every symbol include/*.h declares
(no compute calls: this runs without a GPU)."""
import ctypes
import os
import re

import numpy as np

from conftest import ROOT


def declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", text)))


def test_hip_library_exports_every_declared_symbol():
    path = os.path.join(ROOT, "raytracer_amd", "librtx_hip.so")
    assert os.path.exists(path), "librtx_hip.so not built: run __graft_entry__.build()"
    lib = ctypes.CDLL(path)
    names = declared("rtx.h", "rtx_")
    assert len(names) >= 19
    for n in names:
        assert hasattr(lib, n), n


def test_host_library_exports_every_declared_symbol(libs):
    lib = ctypes.CDLL(os.path.join(ROOT, "raytracer_amd", "librtx_host.so"))
    names = declared("rtx_host.h", "rtxh_")
    assert len(names) >= 24
    for n in names:
        assert hasattr(lib, n), n


def test_python_binding_lists_match_headers():
    from raytracer_amd import _capi
    assert sorted(_capi.EXPORTED_HIP) == declared("rtx.h", "rtx_")
    assert sorted(_capi.EXPORTED_HOST) == declared("rtx_host.h", "rtxh_")


def test_pod_layouts_match_reference_sizes():
    # sizes / offsets probed from the reference's structs
    from raytracer_amd import _capi
    assert _capi.OBJECT.itemsize == 256 and _capi.MATERIAL.itemsize == 128
    assert _capi.BOX.itemsize == 64 and _capi.TEXTURE.itemsize == 16
    assert ctypes.sizeof(_capi.Camera) == 96
    assert _capi.OBJECT.fields["vertices"][1] == 80 and _capi.OBJECT.fields["materialID"][1] == 176
    assert _capi.MATERIAL.fields["textureID"][1] == 112 and _capi.MATERIAL.fields["emmissionIntensity"][1] == 100
    assert _capi.Camera.position.offset == 48 and _capi.Camera.aspectRatio.offset == 72 and _capi.Camera.fov.offset == 84


def test_c_header_layout_static_asserts(tmp_path):
    import subprocess
    src = tmp_path / "layout.cpp"
    src.write_text('#include "rtx.h"\n#include "rtx_host.h"\n#include <cstddef>\n'
                   'static_assert(sizeof(RtxObject)==256 && offsetof(RtxObject,position)==16 && offsetof(RtxObject,normals)==32 && '
                   'offsetof(RtxObject,vertices)==80 && offsetof(RtxObject,uvs)==128 && offsetof(RtxObject,materialID)==176,"Object");\n'
                   'static_assert(sizeof(RtxMaterial)==128 && offsetof(RtxMaterial,specularIntensity)==64 && offsetof(RtxMaterial,textureID)==112,"Material");\n'
                   'static_assert(sizeof(RtxBoundingBox)==64 && offsetof(RtxBoundingBox,minimalPosition)==16 && offsetof(RtxBoundingBox,maximalPosition)==32,"Box");\n'
                   'static_assert(sizeof(RtxTexture)==16 && sizeof(RtxCamera)==96 && offsetof(RtxCamera,position)==48 && offsetof(RtxCamera,fov)==84,"Tex/Cam");\n'
                   'int main(){return 0;}\n')
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)])


def test_ctypes_mirrors_of_config_info_counters_match_the_header(tmp_path):
    """RtxConfig / RtxInfo / RtxCounters as the Python binding declares them must be what a C caller of include/rtx.h sees."""
    import subprocess
    from raytracer_amd import _capi
    src = tmp_path / "sizes.c"
    src.write_text('#include "rtx.h"\n#include <stddef.h>\n#include <stdio.h>\n'
                   'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(RtxConfig), offsetof(RtxConfig, numDevices), offsetof(RtxConfig, lookAhead), '
                   'offsetof(RtxConfig, framesPerBatch), sizeof(RtxInfo), offsetof(RtxInfo, setsCutShort), sizeof(RtxCounters), offsetof(RtxCounters, wavePasses));return 0;}\n')
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])      # the header is C
    got = [int(x) for x in subprocess.check_output([str(exe)], text=True).split()]
    C, I, K = _capi.Config, _capi.Info, _capi.Counters
    assert got == [ctypes.sizeof(C), C.numDevices.offset, C.lookAhead.offset, C.framesPerBatch.offset,
                   ctypes.sizeof(I), I.setsCutShort.offset, ctypes.sizeof(K), K.wavePasses.offset]


def test_hip_backend_has_no_cpu_fallback(monkeypatch):
    """The product path fails loudly when the HIP library is missing."""
    from raytracer_amd import _capi
    monkeypatch.setattr(_capi, "_hip", None)
    monkeypatch.setattr(_capi, "_HERE", "/nonexistent")
    try:
        _capi.hip_lib()
    except RuntimeError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("hip_lib() did not fail")


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "raytracer_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in text and "import oracle" not in text and "oracle/" not in text, f
