import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rt():
    import importlib
    return importlib.import_module("raytracing-course-hw_amd")


@pytest.fixture(scope="session")
def sphere_scene(rt):
    return rt.load_gltf(os.path.join(ROOT, "tests", "golden", "scenes", "hw8_sphere", "sphere_emissive.gltf"))
