"""GPU parity for the hw2 snapshot (Whitted-style tracer with point / directional lights): deterministic, so the HIP path
must reproduce the reference's float radiance bit for bit and the reference program's PPM byte for byte."""
import hashlib
import os

import numpy as np
import pytest

import oracle_lib
import pin_cases

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TXT = os.path.join(GOLD, "scenes", "txt")


def _ppm(w, h, rgb8):
    return b"P6\n%d %d\n255\n" % (w, h) + rgb8.tobytes()


@pytest.mark.parametrize("name", pin_cases.HW2_CASES)
def test_hw2_matches_reference_radiance_bit_for_bit(rt, name):
    sd, w, h, _, depth = rt.load_txt(os.path.join(TXT, name + ".txt"), rt.RT_INTEGRATOR_HW2)
    scene = rt.Scene(sd)
    rgb, rgb8, _ = scene.render(w, h, 1, integrator=rt.RT_INTEGRATOR_HW2, ray_depth=depth)
    gold = np.load(os.path.join(GOLD, "pins_hw2_render.npz"))
    ref = gold[name + "_rgb"]
    diff = int((rgb.view(np.uint32) != ref.view(np.uint32)).sum())
    print(f"hw2 {name}: {diff} differing floats of {ref.size}, max abs {np.abs(rgb - ref).max():.3e}")
    assert diff == 0
    assert hashlib.md5(_ppm(w, h, rgb8)).hexdigest() == bytes(gold[name + "_md5"]).decode()
    scene.close()


def test_hw2_sample_full_size_matches_program_md5_and_oracle(rt):
    """The reference's own hw2/sample.txt at its full 664x510: PPM md5 of the unmodified program, float radiance of the oracle."""
    sd, w, h, _, depth = rt.load_txt(os.path.join(TXT, "hw2_sample.txt"), rt.RT_INTEGRATOR_HW2)
    scene = rt.Scene(sd)
    rgb, rgb8, st = scene.render(w, h, 1, integrator=rt.RT_INTEGRATOR_HW2, ray_depth=depth)
    ref, ref8 = oracle_lib.Hw2Oracle(sd).render(w, h, depth)
    assert np.array_equal(rgb.view(np.uint32), ref.view(np.uint32))
    gold = np.load(os.path.join(GOLD, "pins_hw2_render.npz"))
    assert hashlib.md5(_ppm(w, h, rgb8)).hexdigest() == bytes(gold["hw2_sample_md5"]).decode()
    print(f"hw2 sample {w}x{h}: {st.kernel_ms:.3f} ms on the GPU")
    scene.close()


def test_hw2_sharded_render_equals_whole_frame(rt):
    sd, w, h, _, depth = rt.load_txt(os.path.join(TXT, "hw2_glass_stack.txt"), rt.RT_INTEGRATOR_HW2)
    scene = rt.Scene(sd)
    whole, _, _ = scene.render(w, h, 1, integrator=rt.RT_INTEGRATOR_HW2, ray_depth=depth)
    full = np.zeros_like(whole)
    for k in range(3):
        buf, _, _ = scene.render(w, h, 1, integrator=rt.RT_INTEGRATOR_HW2, ray_depth=depth, shard_index=k, shard_count=3, want_rgb8=False)
        p = rt.make_params(w, h, 1, integrator=rt.RT_INTEGRATOR_HW2, ray_depth=depth, shard_index=k, shard_count=3)
        part = rt.unshard(p, buf)
        full += part
    assert np.array_equal(full, whole)
    scene.close()
