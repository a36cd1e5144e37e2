"""Device building blocks against the host libraries the reference links: rt_logf vs glibc logf (the reference's
normal_distribution calls std::log(float)), and the device minstd/uniform/normal streams vs libstdc++."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hooks():
    L = C.CDLL(os.path.join(ROOT, "raytracing-course-hw_amd", "librtamd_testhooks.so"))
    L.rtt_logf.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.rtt_rng_streams.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p]
    return L


def test_device_logf_is_bit_identical_to_host_libm(hooks):
    """Every float in (0, 1] with a stride (the polar method only ever takes logs of r2 in (0,1]) plus random
    positive floats.  A host without FMA would select glibc's non-FMA logf and may differ in the last bit; the
    test reports that case explicitly."""
    bits = np.arange(0x00800000, 0x3F800001, 257, dtype=np.uint32)       # ~4.1M normal floats up to 1.0
    x = np.concatenate([bits.view(np.float32), np.random.default_rng(1).uniform(1e-30, 1e30, 1 << 20).astype(np.float32)])
    dev = np.zeros_like(x)
    assert hooks.rtt_logf(x.ctypes.data, dev.ctypes.data, x.size) == 0
    host = np.zeros_like(x)
    L = oracle_lib.lib()
    L.rto_logf_array.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.rto_logf_array(x.ctypes.data, host.ctypes.data, x.size)
    diff = int((dev.view(np.uint32) != host.view(np.uint32)).sum())
    fma = "fma" in open("/proc/cpuinfo").read()
    print(f"logf: {x.size} inputs, {diff} differ from the host libm (host cpu has fma: {fma})")
    assert diff == 0 or not fma


def test_device_rng_streams_match_libstdcxx(hooks):
    n_seeds, n_u, n_n = 2000, 5, 9
    dev = np.zeros((n_seeds, n_u + n_n), np.float32)
    assert hooks.rtt_rng_streams(0, n_seeds, n_u, n_n, dev.ctypes.data) == 0
    L = oracle_lib.lib()
    host = np.zeros_like(dev)
    for s in range(n_seeds):
        L.rto_rng_kat(s, n_u, n_n, host[s].ctypes.data)
    assert np.array_equal(dev.view(np.uint32), host.view(np.uint32))
    assert np.array_equal(dev[0], dev[1])  # seed 0 == seed 1


def test_runner_up_gap_code_is_a_floor_within_a_factor_of_two(hooks):
    """rt_exact.h: the runner-up's distance behind the hit travels in six bits of the hit word as a power of two of t.  The gate works
    with pt_gap_floor, which must never exceed the true gap (else a hit that needs the exact walk could pass) and stay within 2x."""
    hooks.rtt_gap_code.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    rng = np.random.default_rng(3)
    t = np.exp(rng.uniform(np.log(1e-5), np.log(1e3), 200000)).astype(np.float32)
    rel = np.exp(rng.uniform(np.log(1e-13), np.log(1e6), t.size)).astype(np.float32)
    t2 = (t + t * rel).astype(np.float32)
    t2[:1000] = t[:1000]                                   # exact ties
    t2[1000:2000] = 2e4                                    # no runner-up (2 x RT_T_MAX)
    fl = np.zeros_like(t); code = np.zeros(t.size, np.uint32)
    assert hooks.rtt_gap_code(t.ctypes.data, t2.ctypes.data, fl.ctypes.data, code.ctypes.data, t.size) == 0
    gap = (t2.astype(np.float64) - t.astype(np.float64))
    gap32 = (t2 - t).astype(np.float64)                    # what the device subtracts
    assert np.all((code & ~np.uint32(0x3F000000)) == 0)    # only the six code bits
    assert np.all(fl.astype(np.float64) <= np.maximum(gap32, 0) * (1 + 1e-6))
    c = (code >> 24).astype(np.int64)
    mid = (c > 0) & (c < 63)
    assert np.all(fl[mid].astype(np.float64) * 2.000002 >= gap32[mid])    # within a factor of two (and the two ulp the device rounds down by) where the code is not saturated
    assert np.all(fl[:1000] == 0) and np.all(c[1000:2000] > 40)
    assert gap.min() >= 0


def test_gate_decisions_on_constructed_cases(hooks):
    """pt_hit_stands on boxes / rays / hits built by hand.  Stands: a hit well inside its box with a distant runner-up; a hit 0.01 in front
    of its box (the reference's triangle test reports such) with nothing near it; a hit in a flat box with nothing near it.  Does not: an
    exact tie or a runner-up within 4 ulp; the hit in front of its box with a runner-up inside the window (0.0075 here: the reference
    prunes the box if that one came first); the flat-box hit and a hit in the entry face of its box with a runner-up 1e-6 / 1e-7 behind."""
    hooks.rtt_hit_stands.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    c2, k = np.float32(10 * 2.0 ** -20), np.float32(2.0 ** -7)
    d = np.array([0.3, -0.5, 0.81], np.float64); d /= np.linalg.norm(d)
    o = np.array([1.0, 6.0, -4.0])
    def case(lo, hi, t, gap):
        return [*lo, *hi, *o, *d, t, gap, c2, k]
    P = lambda t: o + t * d
    t = 5.0
    p = P(t)
    box = (p - 0.05, p + 0.05)
    flat = (np.array([p[0] - 1, p[1], p[2] - 1]), np.array([p[0] + 1, p[1], p[2] + 1]))     # flat in y, the hit lies in it
    ahead = (P(t + 0.01) - 0.002, P(t + 0.01) + 0.002)                                       # the ray meets this box only behind the hit
    edge = (np.array([p[0], p[1] - 0.05, p[2] - 0.05]), np.array([p[0] + 0.05, p[1] + 0.05, p[2] + 0.05]))  # the hit lies in the entry face: window > 0
    cases = np.array([case(*box, t, 1.0), case(*box, t, 0.0), case(*box, t, 1e-6), case(*ahead, t, 1.0), case(*ahead, t, 0.005),
                      case(*flat, t, 1.0), case(*flat, t, 1e-6), case(*edge, t, 1e-7)], np.float32)
    out = np.zeros(len(cases), np.uint32)
    assert hooks.rtt_hit_stands(cases.ctypes.data, out.ctypes.data, len(cases)) == 0
    print("gate decisions:", out.tolist())
    assert out.tolist() == [1, 0, 0, 1, 0, 1, 0, 0]


def test_grid_boxes_are_entered_by_every_ray_that_enters_the_float_box(hooks):
    """The persistent pipeline's walkers read their node boxes from a 16-bit grid over the scene (rt_types.h GpuNodeQ,
    device/rt_node_grid.h) and test them with the ray moved into grid coordinates (rt_device.h slab_test_q).  Pruning must stay
    conservative: whenever the float test (slab_test, what every other pipeline walks with) enters a box, the grid test enters it
    too — for origins anywhere in the grid's box (camera corner included), rays along the axes, flat boxes, origins on box
    faces, thin scenes."""
    rng = np.random.default_rng(11)
    hooks.rtt_slab_q.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    total_hits = 0
    for scene_lo, scene_hi in (((-10.0, -3.0, -10.0), (10.0, 5.0, 10.0)),        # a room
                               ((100.0, 100.0, -0.001), (100.5, 130.0, 0.001)),     # far from the origin, thin along z
                               ((-1e-3, -1e-3, -1e-3), (1e-3, 1e-3, 1e-3)),         # tiny
                               ((0.0, 0.0, 0.0), (4000.0, 1.0, 0.0))):              # flat along z, long along x
        lo_s, hi_s = np.array(scene_lo), np.array(scene_hi)
        n = 400_000
        ext = hi_s - lo_s
        c = rng.uniform(lo_s, hi_s, (n, 3))
        half = rng.uniform(0, 1, (n, 3)) ** 4 * 0.25 * ext * rng.choice([0.0, 1.0, 1.0, 1.0], (n, 3))  # some flat boxes
        blo = np.maximum(c - half, lo_s).astype(np.float32)
        bhi = np.minimum(c + half, hi_s).astype(np.float32)
        bhi = np.maximum(bhi, blo)
        o = rng.uniform(lo_s, hi_s, (n, 3))
        o[: n // 8] = lo_s                                                        # the grid's own corner: the largest grid coordinates are at the other end
        o[n // 8: n // 4] = hi_s
        face = rng.integers(0, 3, n)
        on_face = rng.random(n) < 0.2                                             # origins on a face of their box
        o[on_face, face[on_face]] = blo[on_face, face[on_face]]
        target = rng.uniform(blo, np.maximum(bhi, blo))                           # aim at the box, so that most rays matter
        d = target - o + rng.normal(0, 1e-3, (n, 3)) * ext                     # noise per axis: thin scenes keep their hits
        axis_par = rng.random(n) < 0.15                                           # rays along an axis: a zero component
        d[axis_par, face[axis_par]] = 0.0
        nz = np.linalg.norm(d, axis=1) > 0
        d[~nz] = (1.0, 0.0, 0.0)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        tbest = np.where(rng.random(n) < 0.5, 3.0e38, rng.uniform(0, 2, n) * np.linalg.norm(ext)).astype(np.float32)
        cases = np.concatenate([blo, bhi, o.astype(np.float32), d.astype(np.float32), tbest[:, None]], axis=1).astype(np.float32)
        cases = np.ascontiguousarray(cases)
        grid_box = np.concatenate([lo_s, hi_s]).astype(np.float32)
        out = np.zeros(n, np.uint32)
        assert hooks.rtt_slab_q(cases.ctypes.data, grid_box.ctypes.data, out.ctypes.data, n) == 0
        fits = (out & 4) != 0
        hit_f, hit_q = (out & 1) != 0, (out & 2) != 0
        lost = hit_f & ~hit_q
        total_hits += int(hit_f.sum())
        print(f"grid over {scene_lo}..{scene_hi}: {int(hit_f.sum())} float hits, {int(hit_q.sum())} grid hits, {int(lost.sum())} lost, {int((~fits).sum())} misfits")
        assert fits.all()
        assert not lost.any(), cases[lost][:5]
    assert total_hits > 300_000     # the cases do exercise the test
