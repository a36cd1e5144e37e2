"""rt_multi_*: several GPUs in one process under the C-ABI.  On the one-GPU test machine the device list repeats device 0, which
runs the whole path — one scene replica and one host thread per list entry, shard renders side by side, peer copies to the
first device, the tile-scatter kernel — with the exchange degenerating to device-to-device copies.  The frame must be
bit-identical to the single-device render (the per-pixel seed is global: hw8/src/sceneio.cpp:389-391)."""
import numpy as np
import pytest

import oracle_lib
import pin_cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 2, 3, 5])
def test_multi_device_frame_is_bit_identical(rt, n):
    sd = pin_cases.random_triangle_scene(n=400, seed=33)
    w, h, spp = 150, 100, 5                      # 5 x 4 tiles of 32: border tiles on both edges
    scene = rt.Scene(sd)
    ref, ref8, st1 = scene.render(w, h, spp)
    scene.close()
    multi = rt.MultiScene(sd, [0] * n)
    rgb, rgb8, st = multi.render(w, h, spp)
    assert np.array_equal(rgb, ref, equal_nan=True) and np.array_equal(rgb8, ref8)
    assert st.samples == st1.samples == w * h * spp
    assert (st.closest_hit_queries, st.light_pdf_queries) == (st1.closest_hit_queries, st1.light_pdf_queries)
    only8 = multi.render(w, h, spp, want_float=False)[1]     # u8 only, as the CLI asks for
    assert np.array_equal(only8, ref8)
    multi.close()


def test_multi_device_more_devices_than_tiles_and_oracle(rt, sphere_scene):
    multi = rt.MultiScene(sphere_scene, [0, 0, 0, 0, 0, 0])
    rgb, rgb8, st = multi.render(40, 33, 4)                   # 2 x 2 tiles for six shards: two shards are empty
    ref, ref8, _ = oracle_lib.Hw8Oracle(sphere_scene).render(40, 33, 4)
    assert np.array_equal(rgb, ref) and np.array_equal(rgb8, ref8) and st.samples == 40 * 33 * 4
    multi.close()


def test_multi_device_errors(rt, sphere_scene):
    with pytest.raises(rt.RtError):
        rt.MultiScene(sphere_scene, [99])
    multi = rt.MultiScene(sphere_scene, [0, 0])
    import ctypes as C
    p = rt.make_params(64, 64, 2, shard_index=1, shard_count=2)   # a sharded request is refused: the multi object shards itself
    buf = np.zeros(64 * 64 * 3, np.uint8)
    assert rt.lib.rt_multi_render(multi._h, C.byref(p), None, buf.ctypes.data, None) == -1
    multi.close()
