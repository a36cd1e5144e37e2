#!/usr/bin/env python3
"""hw6 practice6_2 (BASELINE.json configs[2]) at full size: the persistent pipeline on the GPU-built tree against the same pipeline on
the host-built tree (every pixel), and against the CPU oracle on a large crop.  (diagnostic; the oracle is test infrastructure)
usage: hw6_fullframe_check.py [--spp N] [--crop S]"""
import argparse, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
rt = importlib.import_module("raytracing-course-hw_amd")
import pin_cases, oracle_lib
ap = argparse.ArgumentParser()
ap.add_argument("--spp", type=int, default=16)
ap.add_argument("--crop", type=int, default=384)
ap.add_argument("--x0", type=int, default=-1)
ap.add_argument("--y0", type=int, default=-1)
a = ap.parse_args()
sd = pin_cases.load_hw6("practice6_2")
W = H = 1024
frames = {}
for name in ("device", "host"):
    if name == "host": os.environ["RTAMD_HOST_BVH"] = "1"
    else: os.environ.pop("RTAMD_HOST_BVH", None)
    sc = rt.Scene(sd)
    frames[name], _, st = sc.render(W, H, a.spp, integrator=rt.RT_INTEGRATOR_HW6, want_rgb8=False)
    print(f"{name} tree: {st.kernel_ms:.1f} ms", flush=True)
    sc.close()
os.environ.pop("RTAMD_HOST_BVH", None)
print("pixels on which the two trees differ:", int(np.any(frames["device"] != frames["host"], axis=2).sum()), "of", W * H, flush=True)
c = a.crop
x0 = a.x0 if a.x0 >= 0 else (W - c) // 2
y0 = a.y0 if a.y0 >= 0 else (H - c) // 2
t0 = time.time()
ref, _, _ = oracle_lib.Hw6Oracle(sd).render(W, H, a.spp, rect=(x0, y0, c, c))
got = frames["device"][y0:y0 + c, x0:x0 + c]
diff = np.any(got != ref, axis=2)
err = np.abs(got.astype(np.float64) - ref).max()
print(f"oracle crop {c}x{c} at ({x0},{y0}), {a.spp} spp, {time.time() - t0:.1f} s: {int(diff.sum())} of {c * c} pixels not bit-exact, max abs difference {err:.3e}")
for (y, x) in np.argwhere(diff)[:10]:
    print(f"    pixel ({x0 + x},{y0 + y}): gpu {got[y, x]} oracle {ref[y, x]}")
