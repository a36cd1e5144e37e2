"""Throughput mode (rt_render_params.sample_streams = K > 1): K decorrelated random streams per pixel.

It is NOT the reference's pixel stream, so its parity with the reference is statistical; what can be checked exactly is that
stream k of a pixel is an ordinary replay of the reference's per-pixel loop (hw8/src/sceneio.cpp:387-396, scene.cpp:167-177) with
the engine seeded y*W+x + k*W*H — which the oracle reproduces through its seed-offset hook."""
import numpy as np
import pytest

import oracle_lib
import pin_cases

pytestmark = pytest.mark.gpu


def _oracle_streams(sd, w, h, spp, k_streams, depth=0):
    orc = oracle_lib.Hw8Oracle(sd)
    per = spp // k_streams
    total = np.zeros((h, w, 3), np.float32)
    for k in range(k_streams):
        part, _, _ = orc.render(w, h, per, ray_depth=depth, seed_offset=k * w * h)
        total = total + part * np.float32(per)          # float(1/per) * sum -> sum (to an ulp)
    return total * np.float32(1.0 / spp)


@pytest.mark.parametrize("case", ["sphere", "soup"])
def test_each_stream_is_a_replay_with_an_offset_seed(rt, sphere_scene, case):
    sd = sphere_scene if case == "sphere" else pin_cases.random_triangle_scene(n=300, seed=5)
    w, h, spp, k = 48, 36, 12, 4
    scene = rt.Scene(sd)
    rgb, rgb8, st = scene.render(w, h, spp, sample_streams=k)
    ref = _oracle_streams(sd, w, h, spp, k)
    err = np.abs(rgb.astype(np.float64) - ref)
    tol = 2e-6 * np.maximum(1.0, np.abs(ref))           # the oracle's per-stream mean is multiplied back to a sum: ulp-level slack
    print(f"{case}: K={k} max |gpu - composed oracle| {err.max():.2e}")
    assert np.all(err <= tol)
    assert st.samples == w * h * spp
    # one stream is the replay mode itself
    a, a8, _ = scene.render(w, h, spp)
    b, b8, _ = scene.render(w, h, spp, sample_streams=1)
    assert np.array_equal(a, b, equal_nan=True) and np.array_equal(a8, b8)
    # deterministic, and different from replay mode
    again, _, _ = scene.render(w, h, spp, sample_streams=k)
    assert np.array_equal(again, rgb, equal_nan=True) and not np.array_equal(a, rgb, equal_nan=True)
    scene.close()


def test_shards_of_a_throughput_render_assemble_to_the_frame(rt, sphere_scene):
    scene = rt.Scene(sphere_scene)
    w, h, spp, k = 72, 40, 8, 4                          # border tiles are padded
    full, full8, _ = scene.render(w, h, spp, sample_streams=k)
    acc, acc8 = np.zeros_like(full), np.zeros_like(full8)
    for r in range(3):
        p = rt.make_params(w, h, spp, shard_index=r, shard_count=3, tile=16, sample_streams=k)
        buf, buf8, _ = scene.render(w, h, spp, shard_index=r, shard_count=3, tile=16, sample_streams=k)
        acc += rt.unshard(p, buf)
        acc8 += rt.unshard(p, buf8)
    assert np.array_equal(acc, full) and np.array_equal(acc8, full8)
    scene.close()


def test_statistical_agreement_with_replay_mode(rt, sphere_scene):
    """Same estimator, different random numbers: against a 2048-spp replay render both 64-spp images have the same error level and
    no bias (mean signed difference within 4 standard errors)."""
    scene = rt.Scene(sphere_scene)
    w, h = 96, 64
    conv, _, _ = scene.render(w, h, 2048, want_rgb8=False)
    replay, _, _ = scene.render(w, h, 64, want_rgb8=False)
    thr, _, _ = scene.render(w, h, 64, want_rgb8=False, sample_streams=8)
    scene.close()
    ok = np.isfinite(conv).all(axis=2) & np.isfinite(replay).all(axis=2) & np.isfinite(thr).all(axis=2)
    e_r, e_t = (replay - conv)[ok].astype(np.float64), (thr - conv)[ok].astype(np.float64)
    rmse_r, rmse_t = np.sqrt((e_r ** 2).mean()), np.sqrt((e_t ** 2).mean())
    bias_t, se_t = e_t.mean(), e_t.std() / np.sqrt(e_t.size)
    print(f"rmse vs 2048 spp: replay {rmse_r:.4f}, throughput(K=8) {rmse_t:.4f}; bias {bias_t:.2e} (standard error {se_t:.2e})")
    assert 0.75 < rmse_t / rmse_r < 1.33
    assert abs(bias_t) < 4 * se_t + 1e-6


def test_throughput_mode_rejects_what_it_cannot_do(rt, sphere_scene):
    scene = rt.Scene(sphere_scene)
    with pytest.raises(rt.RtError):
        scene.render(16, 16, 10, sample_streams=4)       # 10 is not a multiple of 4
    with pytest.raises(rt.RtError):
        scene.render(16, 16, 1024, sample_streams=512)   # more than 256 streams
    scene.close()
    s6 = rt.Scene(pin_cases.hw6_soup())
    with pytest.raises(rt.RtError):   # hw6 has the streams, not the other estimator options
        s6.render(16, 16, 4, integrator=rt.RT_INTEGRATOR_HW6, sample_streams=2, flags=rt.RT_FLAG_SAMPLE_SEEDS)
    s6.close()


@pytest.mark.parametrize("name", ["practice6_1", "hw6_soup"])
def test_hw6_streams_are_replays_with_an_offset_seed(rt, name):
    """Throughput mode for the hw6 integrator (hw6/src/scene.cpp:47-105, seeding hw6/src/sceneio.cpp:280-284): stream k of a pixel is the
    reference's per-pixel loop with the engine seeded y*W+x + k*W*H; the oracle composes the K replays through its seed-offset hook.
    Also: one stream is replay mode itself, shards assemble to the frame, and the render is deterministic."""
    sd = pin_cases.HW6_CASES[name][0]()
    w, h, spp, k = 56, 40, 12, 4
    scene = rt.Scene(sd)
    rgb, rgb8, st = scene.render(w, h, spp, integrator=rt.RT_INTEGRATOR_HW6, sample_streams=k)
    orc = oracle_lib.Hw6Oracle(sd)
    per = spp // k
    total = np.zeros((h, w, 3), np.float32)
    for i in range(k):
        part, _, _ = orc.render(w, h, per, seed_offset=i * w * h)
        total = total + part * np.float32(per)
    ref = total * np.float32(1.0 / spp)
    err = np.abs(rgb.astype(np.float64) - ref)
    print(f"hw6 {name}: K={k} max |gpu - composed oracle| {err.max():.2e}, pipeline {st.pipeline}")
    assert np.all(err <= 2e-6 * np.maximum(1.0, np.abs(ref))) and st.samples == w * h * spp and st.pipeline == rt.RT_PIPELINE_PERSISTENT
    a, a8, _ = scene.render(w, h, spp, integrator=rt.RT_INTEGRATOR_HW6)
    b, b8, _ = scene.render(w, h, spp, integrator=rt.RT_INTEGRATOR_HW6, sample_streams=1)
    assert np.array_equal(a, b, equal_nan=True) and np.array_equal(a8, b8) and not np.array_equal(a, rgb, equal_nan=True)
    again, _, _ = scene.render(w, h, spp, integrator=rt.RT_INTEGRATOR_HW6, sample_streams=k)
    assert np.array_equal(again, rgb, equal_nan=True)
    acc = np.zeros_like(rgb)
    for r in range(3):
        p = rt.make_params(w, h, spp, integrator=rt.RT_INTEGRATOR_HW6, shard_index=r, shard_count=3, tile=16, sample_streams=k)
        buf, _, _ = scene.render(w, h, spp, integrator=rt.RT_INTEGRATOR_HW6, shard_index=r, shard_count=3, tile=16, sample_streams=k, want_rgb8=False)
        acc += rt.unshard(p, buf)
    assert np.array_equal(acc, rgb)
    scene.close()


def test_russian_roulette_and_per_sample_seeds(rt, sphere_scene):
    """SURVEY 8(f)3, the non-replay estimator options of throughput mode.  RT_FLAG_SAMPLE_SEEDS: every camera sample has its own
    engine seeded from (pixel, sample index), so the image does not depend on how the samples are dealt to streams (up to the
    order of the float additions).  RT_FLAG_RUSSIAN_ROULETTE: paths die early with probability 1 - q and survivors are weighted
    1 / q — fewer queries, the same expectation: against a 2048-spp replay render the bias stays within 4 standard errors, and variance x
    queries does not get worse."""
    scene = rt.Scene(sphere_scene)
    w, h = 96, 64
    conv, _, _ = scene.render(w, h, 2048, want_rgb8=False)
    thr, _, st_thr = scene.render(w, h, 64, want_rgb8=False, sample_streams=8)
    s4, _, _ = scene.render(w, h, 64, want_rgb8=False, sample_streams=4, flags=rt.RT_FLAG_SAMPLE_SEEDS)
    s8, _, _ = scene.render(w, h, 64, want_rgb8=False, sample_streams=8, flags=rt.RT_FLAG_SAMPLE_SEEDS)
    rr, _, st_rr = scene.render(w, h, 64, want_rgb8=False, sample_streams=8, flags=rt.RT_FLAG_SAMPLE_SEEDS | rt.RT_FLAG_RUSSIAN_ROULETTE)
    with pytest.raises(rt.RtError):
        scene.render(w, h, 64, flags=rt.RT_FLAG_RUSSIAN_ROULETTE)      # replay mode keeps the reference's estimator
    scene.close()
    ok = np.isfinite(conv).all(axis=2) & np.isfinite(thr).all(axis=2) & np.isfinite(s8).all(axis=2) & np.isfinite(rr).all(axis=2) & np.isfinite(s4).all(axis=2)
    assert ok.mean() > 0.99
    assert np.allclose(s4[ok], s8[ok], rtol=2e-5, atol=2e-6) and not np.array_equal(s8, thr)
    def stats(img):
        e = (img - conv)[ok].astype(np.float64)
        return np.sqrt((e ** 2).mean()), e.mean(), e.std() / np.sqrt(e.size)
    rmse_t, _, _ = stats(thr)
    rmse_s, bias_s, se_s = stats(s8)
    rmse_r, bias_r, se_r = stats(rr)
    print(f"rmse vs 2048 spp: streams {rmse_t:.4f}, per-sample seeds {rmse_s:.4f} (bias {bias_s:.2e}, se {se_s:.2e}), + roulette {rmse_r:.4f} (bias {bias_r:.2e}, se {se_r:.2e}); "
          f"closest-hit queries {st_thr.closest_hit_queries} -> {st_rr.closest_hit_queries}")
    assert 0.75 < rmse_s / rmse_t < 1.33 and abs(bias_s) < 4 * se_s + 1e-6
    assert 0.75 < rmse_r / rmse_t < 2.0 and abs(bias_r) < 4 * se_r + 1e-6
    assert st_rr.closest_hit_queries < st_thr.closest_hit_queries
    # and it has to pay: variance x work (closest-hit queries) must not exceed the figure without roulette (q follows the path's accumulated
    # throughput, floor 0.25; round 2's per-bounce q lost a factor of two here)
    eff = (rmse_r ** 2 * st_rr.closest_hit_queries) / (rmse_s ** 2 * st_thr.closest_hit_queries)
    print(f"roulette: variance x queries relative to no roulette = {eff:.3f}")
    assert eff <= 1.05
