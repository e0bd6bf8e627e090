"""SURVEY.md section 5, "race detection / sanitizers": the host side of the library -- both scene loaders (untrusted
text), validation, packing, the k-d cluster build, band arithmetic, the PPM writer -- compiled from the library's own
sources under AddressSanitizer + UndefinedBehaviorSanitizer (make -C ray-tracing_amd/csrc asan) and run on every scene
file of the suite plus seeded mutations of each (tools/host_asan_check.cpp).  CPU only; no device code is involved."""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "ray-tracing_amd", "host_asan_check")


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not installed")
def test_host_code_is_clean_under_asan_and_ubsan():
    p = subprocess.run(["make", "-C", os.path.join(ROOT, "ray-tracing_amd", "csrc"), "asan"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode == 0, p.stdout.decode()[-3000:]
    scenes = sorted(glob.glob(os.path.join(ROOT, "tests", "scenes", "*.txt")) + glob.glob(os.path.join(ROOT, "tests", "golden", "bundled", "*.txt")))
    assert len(scenes) >= 12
    p = subprocess.run([EXE, "150"] + scenes, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode()
    assert p.returncode == 0 and "no sanitizer report" in out, out[-4000:]
    assert "ERROR: AddressSanitizer" not in out and "runtime error" not in out, out[-4000:]
