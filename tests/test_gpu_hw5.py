"""GPU parity for the hw5 snapshot: .txt scenes with TRIANGLE figures, BVH, box / ellipsoid / triangle lights and one
engine per pixel — the reference program's own seeding, so the HIP path is compared with the reference's pixels."""
import hashlib
import os

import numpy as np
import pytest

import oracle_lib
import pin_cases

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TXT = os.path.join(GOLD, "scenes", "txt")
RMSE_TOL = 1e-3


def _ppm(w, h, rgb8):
    return b"P6\n%d %d\n255\n" % (w, h) + rgb8.tobytes()


@pytest.mark.parametrize("name", pin_cases.HW5_CASES)
def test_hw5_matches_the_reference_radiance(rt, name):
    """Against the float radiance the reference's own sources produced (tests/golden/pins_hw5_render.npz).  Tolerance
    1e-3 RMSE; in practice bit-exact (the only licensed difference is the double stand-in for the long-double eps sum)."""
    sd, w, h, spp, depth = rt.load_txt(os.path.join(TXT, name + ".txt"), rt.RT_INTEGRATOR_HW5)
    scene = rt.Scene(sd)
    rgb, rgb8, _ = scene.render(w, h, spp, integrator=rt.RT_INTEGRATOR_HW5, ray_depth=depth)
    gold = np.load(os.path.join(GOLD, "pins_hw5_render.npz"))
    ref = gold[name + "_rgb"]
    rmse = float(np.sqrt(np.mean((rgb.astype(np.float64) - ref) ** 2)))
    nbad = int((rgb.view(np.uint32) != ref.view(np.uint32)).any(axis=2).sum())
    md5_ok = hashlib.md5(_ppm(w, h, rgb8)).hexdigest() == bytes(gold[name + "_md5"]).decode()
    print(f"hw5 {name}: rmse {rmse:.3e}, {nbad} of {w * h} pixels differ in any bit, PPM md5 equal to the program's: {md5_ok}")
    assert ref.mean() > 0.01 and rmse < RMSE_TOL
    assert nbad == 0 and md5_ok   # the reference program's own floats and its PPM, bit for bit (the 80-bit `t + eps` stand-in has never shown)
    scene.close()


def test_hw5_figure_and_light_order_follow_the_reference(rt):
    """Scene::initBVH's reordering and FiguresMix's light list decide which light a random number selects and the order of
    the pdf additions: the product's host preparation must reproduce them (the oracle's are pinned against the reference)."""
    sd, w, h, spp, depth = rt.load_txt(os.path.join(TXT, "hw5_mixed_figures.txt"), rt.RT_INTEGRATOR_HW5)
    scene = rt.Scene(sd)
    _, lo = oracle_lib.Hw5Oracle(sd).orders()
    assert scene.info().n_lights == len(lo) == 8
    assert np.array_equal(scene.light_order(), lo)
    scene.close()


def test_hw5_larger_frame_against_oracle_and_sharded(rt):
    sd, _, _, _, depth = rt.load_txt(os.path.join(TXT, "hw5_mixed_figures.txt"), rt.RT_INTEGRATOR_HW5)
    w, h, spp = 200, 150, 16
    scene = rt.Scene(sd)
    rgb, _, st = scene.render(w, h, spp, integrator=rt.RT_INTEGRATOR_HW5, ray_depth=depth, want_rgb8=False)
    ref, _ = oracle_lib.Hw5Oracle(sd).render(w, h, spp, depth)
    rmse = float(np.sqrt(np.mean((rgb.astype(np.float64) - ref) ** 2)))
    print(f"hw5 200x150x16: rmse {rmse:.3e} bit_exact {np.array_equal(rgb, ref)}; {w * h * spp / st.kernel_ms / 1e3:.1f} Msamples/s")
    assert np.array_equal(rgb, ref)
    full = np.zeros_like(rgb)
    for k in range(2):
        buf, _, _ = scene.render(w, h, spp, integrator=rt.RT_INTEGRATOR_HW5, ray_depth=depth, shard_index=k, shard_count=2, want_rgb8=False)
        full += rt.unshard(rt.make_params(w, h, spp, integrator=rt.RT_INTEGRATOR_HW5, ray_depth=depth, shard_index=k, shard_count=2), buf)
    assert np.array_equal(full, rgb)
    scene.close()


def test_txt_scene_with_triangles_is_refused_by_older_integrators(rt):
    sd, w, h, spp, depth = rt.load_txt(os.path.join(TXT, "hw5_mixed_figures.txt"), rt.RT_INTEGRATOR_HW5)
    scene = rt.Scene(sd)
    with pytest.raises(rt.RtError):
        scene.render(w, h, spp, integrator=rt.RT_INTEGRATOR_HW3, ray_depth=depth)
    scene.close()


def test_hw5_emissive_point_light_box_rounding(rt, tmp_path):
    """An emissive zero-area TRIANGLE makes FiguresMix::sample aim rays EXACTLY at a vertex, i.e. through the corners of the boxes of the
    reference's light tree and scene tree — where its own slab test (six divisions on the re-centred box, hw5/src/primitives.cpp:92-116,
    221-223) decides by one ulp.  The hw5 kernel walks the reference's trees with that very test and pruning rule (bvh.h:111-141), so
    the frame must be the oracle's — pinned bit-exact to the compiled reference, tests/golden/pins_hw5_* — bit for bit."""
    src = open(os.path.join(TXT, "hw5_mixed_figures.txt")).read()
    extra = ("NEW_PRIMITIVE\nTRIANGLE 0.2 0.1 -0.3 0.2 0.1 -0.3 0.2 0.1 -0.3\nPOSITION 0.4 2.1 0.8\nROTATION 0.1 0.2 0.05 0.97\nEMISSION 3 3 2\n"
             "NEW_PRIMITIVE\nTRIANGLE -0.4 0.0 0.1 0.5 0.05 0.1 0.0 0.6 -0.2\nPOSITION -0.5 1.5 1.0\nROTATION 0 0.1 0 0.995\nEMISSION 1 2 3\n")
    path = str(tmp_path / "hw5_point_light.txt")
    open(path, "w").write(src + extra)
    sd, w, h, spp, depth = rt.load_txt(path, rt.RT_INTEGRATOR_HW5)
    scene = rt.Scene(sd)
    rgb, _, _ = scene.render(w, h, 12, integrator=rt.RT_INTEGRATOR_HW5, ray_depth=depth, want_rgb8=False)
    scene.close()
    ref, _ = oracle_lib.Hw5Oracle(sd).render(w, h, 12, depth)
    differing = int((rgb.view(np.uint32) != ref.view(np.uint32)).any(axis=2).sum())
    print(f"hw5 emissive point light {w}x{h}x12: {differing} pixels differ from the oracle in any bit, non-finite pixels {int((~np.isfinite(ref)).any(axis=2).sum())}")
    assert np.array_equal(rgb, ref, equal_nan=True)
