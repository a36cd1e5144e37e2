#!/usr/bin/env python3
"""Scene-tree build on the GPU vs on the host for hw6 practice6_2 (BASELINE.json configs[2]): build times, tree shape, work per sample
and Msamples/s with either tree.  usage: bvh_build_probe.py [--spp N]"""
import argparse, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
rt = importlib.import_module("raytracing-course-hw_amd")
import pin_cases
ap = argparse.ArgumentParser()
ap.add_argument("--spp", type=int, default=64)
a = ap.parse_args()
sd = pin_cases.load_hw6("practice6_2")
frames = {}
for host in (False, True):
    if host: os.environ["RTAMD_HOST_BVH"] = "1"
    else: os.environ.pop("RTAMD_HOST_BVH", None)
    rt.Scene(sd).close()  # first creation in the process: code-object load, allocator warm-up
    t0 = time.time(); scene = rt.Scene(sd); t1 = time.time()
    i = scene.info()
    print(f"[{'host' if host else 'device'} tree] create {1e3 * (t1 - t0):.1f} ms (prep {i.prep_ms:.1f}, upload+build {i.upload_ms:.1f}, GPU build {i.bvh_build_ms:.2f}); nodes {i.n_bvh_nodes}, depth {i.bvh_depth}, on_device {i.bvh_on_device}", flush=True)
    rgb, _, st = scene.render(256, 256, 8, integrator=rt.RT_INTEGRATOR_HW6, want_rgb8=False, counters=True)
    frames[host] = rgb
    print(f"    256x256x8 counted: node visits / sample {st.node_visits / st.samples:.1f}, triangle tests / sample {st.triangle_tests / st.samples:.1f}", flush=True)
    best = min(scene.render(1024, 1024, a.spp, integrator=rt.RT_INTEGRATOR_HW6, want_rgb8=False)[2].kernel_ms for _ in range(2))
    print(f"    1024x1024x{a.spp}: {best:.1f} ms = {1024 * 1024 * a.spp / best / 1e3:.1f} Msamples/s", flush=True)
    scene.close()
print("frames identical:", bool(np.array_equal(frames[False], frames[True])))
