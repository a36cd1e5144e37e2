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
