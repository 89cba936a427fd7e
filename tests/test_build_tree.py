"""Host logic of rtx_create's own tree (raytracer_amd/csrc/rtx_build.h): compiled with g++ and run on the CPU."""
import os
import subprocess

from conftest import ROOT


def test_sah_build_and_wide_fold(tmp_path):
    exe = tmp_path / "check_build_tree"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", str(exe),
                           os.path.join(ROOT, "tests", "cpp", "check_build_tree.cpp")])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith("OK")
