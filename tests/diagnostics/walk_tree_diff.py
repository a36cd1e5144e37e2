#!/usr/bin/env python3
"""Which variants of the closest-hit walk agree with the oracle on the headline scene?  Renders the full frame with each variant (a set
of RTAMD_* settings applied at scene creation), collects the pixels on which any two variants differ and asks the CPU oracle about those.
(diagnostic; the oracle is test infrastructure)  usage: walk_tree_diff.py [--spp N] "NAME:ENV=v ENV2=w" ..."""
import argparse, importlib, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
rt = importlib.import_module("raytracing-course-hw_amd")
import gen_synth_room, oracle_lib
ap = argparse.ArgumentParser()
ap.add_argument("--spp", type=int, default=32)
ap.add_argument("--max-pixels", type=int, default=60)
ap.add_argument("variants", nargs="*", default=["device:", "host:RTAMD_HOST_BVH=1"])
a = ap.parse_args()
gltf, _ = gen_synth_room.generate(tempfile.mkdtemp(), 64, 50, 43)
sd = rt.load_gltf(gltf)
W, H = 1920, 1080
frames = {}
for v in a.variants:
    name, _, envs = v.partition(":")
    keys = []
    for kv in envs.split():
        k, val = kv.split("=", 1); os.environ[k] = val; keys.append(k)
    sc = rt.Scene(sd)
    frames[name], _, st = sc.render(W, H, a.spp, want_rgb8=False)
    print(f"{name}: {st.kernel_ms:.1f} ms, exact walks {st.exact_closest_hits}+{st.exact_light_sums}", flush=True)
    sc.close()
    for k in keys: os.environ.pop(k, None)
names = list(frames)
mask = np.zeros((H, W), bool)
for n in names[1:]:
    mask |= np.any(frames[n] != frames[names[0]], axis=2)
diff = np.argwhere(mask)
print(f"{len(diff)} pixels on which the variants differ at {a.spp} spp", flush=True)
orc = oracle_lib.Hw8Oracle(sd)
wrong = {n: 0 for n in names}
for (y, x) in diff[:a.max_pixels]:
    ref, _, _ = orc.render(W, H, a.spp, rect=(int(x), int(y), 1, 1))
    r = ref.reshape(-1, 3)[0]
    ok = {n: bool(np.array_equal(frames[n][y, x], r)) for n in names}
    for n in names: wrong[n] += not ok[n]
    print(f"pixel ({x},{y}): " + "  ".join(f"{n} {'ok' if ok[n] else 'WRONG ' + str(np.abs(frames[n][y, x] - r).max())}" for n in names), flush=True)
print("pixels (of those checked) differing from the oracle:", wrong)
