"""GPU parity for the hw7 snapshot (hw8's integrator before textures: ungated per-material BRDF, alpha = roughness^2,
geometric normal in the light pdf).  The expected radiance comes from the reference's own hw7 sources
(tests/golden/pins_hw7_render.npz, produced by oracle/ref/ref_hw7_scene.cpp)."""
import os

import numpy as np
import pytest

import pin_cases

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RMSE_TOL = 1e-3
CASES = {"practice7_1": (lambda: pin_cases.load_hw7("practice7_1"), 48, 48, 8), "practice7_4": (lambda: pin_cases.load_hw7("practice7_4"), 48, 48, 8),
         "sphere_as_hw7": (lambda: pin_cases.as_hw7(pin_cases.load_sphere()), 40, 40, 6),
         "soup_as_hw7": (lambda: pin_cases.as_hw7(pin_cases.random_triangle_scene()), 40, 32, 6)}


@pytest.mark.parametrize("kernel", ["wavefront", "mega"])
@pytest.mark.parametrize("name", sorted(CASES))
def test_hw7_matches_the_reference_radiance(rt, monkeypatch, name, kernel):
    monkeypatch.setenv("RTAMD_KERNEL", kernel)
    mk, w, h, spp = CASES[name]
    sd = mk()
    scene = rt.Scene(sd)
    rgb, rgb8, _ = scene.render(w, h, spp, integrator=rt.RT_INTEGRATOR_HW7)
    gold = np.load(os.path.join(GOLD, "pins_hw7_render.npz"))
    ref, ref8 = gold[name + "_rgb"], gold[name + "_rgb8"]
    rmse = float(np.sqrt(np.mean((rgb.astype(np.float64) - ref) ** 2)))
    nbad = int((rgb.view(np.uint32) != ref.view(np.uint32)).any(axis=2).sum())
    print(f"hw7 {name} [{kernel}]: rmse {rmse:.3e}, {nbad} of {w * h} pixels differ in any bit, byte mismatches {(rgb8 != ref8).sum()}")
    assert ref.mean() > 0.01 and rmse < RMSE_TOL and nbad <= 2
    scene.close()


def test_hw7_and_hw8_integrators_differ_on_the_same_scene(rt):
    """Sanity: the switch does something (gated vs ungated BRDF, shading vs geometric light normal)."""
    sd = pin_cases.random_triangle_scene()
    scene = rt.Scene(sd)
    a, _, _ = scene.render(40, 32, 6, integrator=rt.RT_INTEGRATOR_HW7, want_rgb8=False)
    b, _, _ = scene.render(40, 32, 6, integrator=rt.RT_INTEGRATOR_HW8, want_rgb8=False)
    assert not np.array_equal(a, b)
    scene.close()
