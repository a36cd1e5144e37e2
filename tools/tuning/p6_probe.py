#!/usr/bin/env python3
"""Probe of the hw6 pipelines on BASELINE.json configs[2] (practice6_2): usage p6_probe.py [--spp N] [--size S] "ENV=a ENV2=b" ..."""
import argparse, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
rt = importlib.import_module("raytracing-course-hw_amd")
import pin_cases
ap = argparse.ArgumentParser()
ap.add_argument("--spp", type=int, default=64)
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--counters", action="store_true")
ap.add_argument("settings", nargs="*", default=[""])
a = ap.parse_args()
sd = pin_cases.load_hw6("practice6_2")
scene = rt.Scene(sd)
for setting in a.settings:
    keys = []
    for kv in setting.split():
        k, v = kv.split("=", 1); os.environ[k] = v; keys.append(k)
    best = None
    for _ in range(2):
        _, _, st = scene.render(a.size, a.size, a.spp, integrator=rt.RT_INTEGRATOR_HW6, want_rgb8=False, flags=rt.RT_FLAG_COUNTERS if a.counters else 0)
        best = st if best is None or st.kernel_ms < best.kernel_ms else best
    print(f"[{setting or 'defaults'}] {a.size}x{a.size}x{a.spp}: {best.kernel_ms:.1f} ms, {best.samples / best.kernel_ms / 1e3:.1f} Msamples/s, pipeline {best.pipeline}, launches {best.launches}, queries {best.closest_hit_queries}+{best.light_pdf_queries}", flush=True)
    for k in keys:
        os.environ.pop(k, None)
scene.close()
