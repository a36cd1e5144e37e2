"""The CLI keeps the reference's command lines and writes the reference's PPM format (GPU needed: it renders)."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAIN = os.path.join(ROOT, "raytracing-course-hw_amd", "rtamd_main")
SCENES = os.path.join(ROOT, "tests", "golden", "scenes")


def test_cli_gltf_surface(tmp_path):
    out = tmp_path / "o.ppm"
    r = subprocess.run([MAIN, os.path.join(SCENES, "hw8_sphere", "sphere_emissive.gltf"), "48", "32", "4", str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "FINISH" in r.stderr, r.stderr
    data = out.read_bytes()
    assert data.startswith(b"P6\n48 32\n255\n") and len(data) == len(b"P6\n48 32\n255\n") + 48 * 32 * 3


def test_cli_txt_surface_hw1_md5(tmp_path):
    out = tmp_path / "o.ppm"
    env = dict(os.environ, RTAMD_SNAPSHOT="hw1")
    r = subprocess.run([MAIN, os.path.join(SCENES, "txt", "hw1_sample.txt"), str(out)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "FINISH" in r.stderr, r.stderr
    assert "UNKNOWN COMMAND: YA" in r.stderr  # the reference prints this for the file's last line too
    assert hashlib.md5(out.read_bytes()).hexdigest() == "353a1038e8aaa368d2957931be2cf87d"  # reference program's file


def test_cli_reports_errors(tmp_path):
    r = subprocess.run([MAIN, str(tmp_path / "missing.gltf"), "8", "8", "1", str(tmp_path / "o.ppm")], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "error" in r.stderr
