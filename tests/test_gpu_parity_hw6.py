"""GPU parity for the hw6 integrator (BASELINE.json configs[2]): HIP path through the C-ABI vs the CPU oracle
(oracle/oracle_hw6.cpp, itself pinned bit-exact against the compiled hw6 reference)."""
import numpy as np
import pytest

import oracle_lib
import pin_cases

pytestmark = pytest.mark.gpu
RMSE_TOL = 1e-3  # BASELINE.json north_star tolerance on linear radiance


def _cmp(tag, rgb, ref, rgb8, ref8):
    rmse = float(np.sqrt(np.mean((rgb.astype(np.float64) - ref) ** 2)))
    bad = int((np.abs(rgb.astype(np.float64) - ref).max(axis=2) > 1e-3).sum())
    print(f"{tag}: rmse {rmse:.3e} desync_pixels {bad}/{rgb.shape[0] * rgb.shape[1]} bit_exact {np.array_equal(rgb, ref)} byte_mismatch {(rgb8 != ref8).sum()}")
    return rmse, bad


@pytest.mark.parametrize("kernel", ["persistent", "mega"])
@pytest.mark.parametrize("name,w,h,spp", [("practice6_1", 64, 48, 6), ("hw6_soup", 64, 48, 8), ("practice6_2", 40, 40, 4)])
def test_hw6_scene_matches_oracle(rt, monkeypatch, name, w, h, spp, kernel):
    """Both organisations of the hw6 integrator: the persistent dataflow pipeline (default: path frames in HBM records, walkers that
    refill, a shader role that runs the frame machine between walks) and the per-lane path machine (RTAMD_KERNEL=mega)."""
    monkeypatch.setenv("RTAMD_KERNEL", kernel)
    sd = pin_cases.HW6_CASES[name][0]()
    scene = rt.Scene(sd)
    assert np.array_equal(scene.light_order(), oracle_lib.Hw6Oracle(sd).light_order())
    rgb, rgb8, st = scene.render(w, h, spp, integrator=rt.RT_INTEGRATOR_HW6, counters=True)
    ref, ref8, _ = oracle_lib.Hw6Oracle(sd).render(w, h, spp)
    rmse, bad = _cmp(f"hw6 {name}[{kernel}] {w}x{h}x{spp} (pipeline {st.pipeline}, {st.closest_hit_queries}+{st.light_pdf_queries} queries)", rgb, ref, rgb8, ref8)
    assert ref.mean() > 0.01
    if kernel == "persistent":  # the default (exactness gate on): the reference's pixels bit for bit
        assert st.reference_exact == 1 and np.array_equal(rgb, ref, equal_nan=True) and np.array_equal(rgb8, ref8)
    else:                       # the per-lane path machine keeps the walkers' padded-box answer: north_star tolerance
        assert rmse < RMSE_TOL and bad <= 1
    assert st.pipeline == (rt.RT_PIPELINE_PERSISTENT if kernel == "persistent" else rt.RT_PIPELINE_SINGLE) and st.closest_hit_queries > w * h * spp
    scene.close()


@pytest.mark.parametrize("depth", [1, 2, 3, 8])
def test_hw6_ray_depths_and_shards_persistent_equals_path_machine(rt, monkeypatch, depth):
    """Depth limits cut the recursion tree at every kind of frame (a DIFFUSE bounce whose child lies beyond the limit still needs its
    light-pdf sum; dielectric reflect / refract children return 0 there), and shards use the compact layout: the two organisations
    must agree bit for bit, and with the oracle."""
    sd = pin_cases.load_hw6("practice6_1")
    scene = rt.Scene(sd)
    monkeypatch.setenv("RTAMD_KERNEL", "mega")
    a, a8, _ = scene.render(72, 56, 5, integrator=rt.RT_INTEGRATOR_HW6, ray_depth=depth)
    sa, _, _ = scene.render(72, 56, 5, integrator=rt.RT_INTEGRATOR_HW6, ray_depth=depth, shard_index=1, shard_count=3, tile=16)
    monkeypatch.setenv("RTAMD_KERNEL", "persistent")
    b, b8, st = scene.render(72, 56, 5, integrator=rt.RT_INTEGRATOR_HW6, ray_depth=depth)
    sb, _, _ = scene.render(72, 56, 5, integrator=rt.RT_INTEGRATOR_HW6, ray_depth=depth, shard_index=1, shard_count=3, tile=16)
    scene.close()
    assert st.pipeline == rt.RT_PIPELINE_PERSISTENT
    assert np.array_equal(a, b, equal_nan=True) and np.array_equal(a8, b8) and np.array_equal(sa, sb, equal_nan=True)
    ref, _, _ = oracle_lib.Hw6Oracle(sd).render(72, 56, 5, ray_depth=depth)
    assert np.array_equal(b, ref, equal_nan=True)


def test_hw6_scene_rejects_wrong_integrator(rt):
    sd = pin_cases.hw6_soup()
    scene = rt.Scene(sd)
    with pytest.raises(rt.RtError):
        scene.render(16, 16, 1, integrator=rt.RT_INTEGRATOR_HW8)
    scene.close()


def test_config3_practice6_2_1024x1024x256_crops(rt):
    """BASELINE.json configs[2] at full size on the GPU; the oracle replays eight 16x16 crops at all 256 samples (the reference's own
    BVH is degenerate on this scene — 5,350 box tests per traversal — so the CPU needs hours for the frame): four on the glass
    bunny and its edge, two on the side walls, one at the ceiling light, one on the floor.  Bit for bit, floats and bytes; one of
    them also against the reference's own compiled hw6 integrator."""
    sd = pin_cases.load_hw6("practice6_2")
    scene = rt.Scene(sd)
    rgb, rgb8, st = scene.render(1024, 1024, 256, integrator=rt.RT_INTEGRATOR_HW6)
    print(f"config 3: {st.kernel_ms:.0f} ms kernel = {1024 * 1024 * 256 / st.kernel_ms / 1e3:.1f} Msamples/s")
    assert np.isfinite(rgb).all() and st.reference_exact == 1
    orc = oracle_lib.Hw6Oracle(sd)
    for (cx, cy) in [(448, 384), (560, 470), (672, 470), (64, 512), (944, 300), (504, 24), (512, 960)]:
        cref, cref8, _ = orc.render(1024, 1024, 256, rect=(cx, cy, 16, 16))
        _cmp(f"config3 crop ({cx},{cy})", rgb[cy:cy + 16, cx:cx + 16], cref, rgb8[cy:cy + 16, cx:cx + 16], cref8)
        assert np.array_equal(rgb[cy:cy + 16, cx:cx + 16], cref, equal_nan=True) and np.array_equal(rgb8[cy:cy + 16, cx:cx + 16], cref8)
    x0, y0 = 500, 560
    ref, ref8, _ = orc.render(1024, 1024, 256, rect=(x0, y0, 16, 16))
    rmse, bad = _cmp("config3 crop", rgb[y0:y0 + 16, x0:x0 + 16], ref, rgb8[y0:y0 + 16, x0:x0 + 16], ref8)
    assert np.array_equal(rgb[y0:y0 + 16, x0:x0 + 16], ref, equal_nan=True) and np.array_equal(rgb8[y0:y0 + 16, x0:x0 + 16], ref8)
    if oracle_lib.ref_path("libref_hw6.so"):  # the reference's own hw6 code on the same crop (65,536 camera samples)
        rref, rref8, _ = oracle_lib.Ref6(sd).render(1024, 1024, 256, rect=(x0, y0, 16, 16))
        rmse, bad = _cmp("config3 crop vs the reference itself", rgb[y0:y0 + 16, x0:x0 + 16], rref, rgb8[y0:y0 + 16, x0:x0 + 16], rref8)
        assert np.array_equal(rgb[y0:y0 + 16, x0:x0 + 16], rref, equal_nan=True) and np.array_equal(rgb8[y0:y0 + 16, x0:x0 + 16], rref8)
    scene.close()


def test_full_size_pixels_that_need_the_references_own_box_decisions(rt):
    """Found by tests/diagnostics/hw6_fullframe_check.py: seven pixels of a 384x384 crop of the 1024x1024 frame (16 spp) were off by up to 0.33
    before the exactness gate of DESIGN.md 3 was ported to hw6 (hits at a box boundary of the reference's degenerate trees, its pruning
    against hits in a box face).  Here: an 8x8 block around each of them against the oracle, bit for bit."""
    sd = pin_cases.load_hw6("practice6_2")
    scene = rt.Scene(sd)
    w = h = 1024
    rgb, _, st = scene.render(w, h, 16, integrator=rt.RT_INTEGRATOR_HW6, want_rgb8=False)
    scene.close()
    assert st.exact_closest_hits > 0 and st.exact_light_sums > 0       # the exact roles ran (about 1e-4 of the queries)
    orc = oracle_lib.Hw6Oracle(sd)
    for (x, y) in [(510, 375), (521, 381), (551, 469), (673, 469), (565, 513), (476, 603), (451, 609)]:
        x0, y0 = x - 4, y - 4
        ref, _, _ = orc.render(w, h, 16, rect=(x0, y0, 8, 8))
        assert np.array_equal(rgb[y0:y0 + 8, x0:x0 + 8], ref), f"block at ({x0},{y0}) differs from the oracle"
