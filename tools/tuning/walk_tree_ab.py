#!/usr/bin/env python3
"""A/B on the headline scene in one process: walkers on the GPU-built tree (default) vs on the reference topology (RTAMD_HOST_BVH=1),
replay mode with exact box decisions, same pixels.  usage: walk_tree_ab.py [--spp N] [--reps R]"""
import argparse, importlib, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
rt = importlib.import_module("raytracing-course-hw_amd")
import gen_synth_room
ap = argparse.ArgumentParser()
ap.add_argument("--spp", type=int, default=64)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
gltf, _ = gen_synth_room.generate(tempfile.mkdtemp(), 64, 50, 43)
sd = rt.load_gltf(gltf)
scenes = {}
for name in ("device", "host"):
    if name == "host": os.environ["RTAMD_HOST_BVH"] = "1"
    else: os.environ.pop("RTAMD_HOST_BVH", None)
    scenes[name] = rt.Scene(sd)
os.environ.pop("RTAMD_HOST_BVH", None)
frames = {}
for rep in range(a.reps):
    for name, scene in scenes.items():
        rgb, _, st = scene.render(1920, 1080, a.spp, want_rgb8=False)
        frames[name] = rgb
        i = scene.info()
        print(f"rep {rep} [{name} walk tree, depth {i.bvh_depth}] {st.kernel_ms:.1f} ms = {st.samples / st.kernel_ms / 1e3:.1f} Msamples/s; exact walks {st.exact_closest_hits}+{st.exact_light_sums}", flush=True)
print("frames identical:", bool(np.array_equal(frames["device"], frames["host"], equal_nan=True)))
for s in scenes.values(): s.close()
