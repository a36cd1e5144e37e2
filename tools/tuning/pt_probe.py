#!/usr/bin/env python3
"""Probe of the persistent pipeline on the headline scene: one scene, many renders with different RTAMD_* settings.
usage: pt_probe.py [--spp N] [--width W --height H] [--shards K] "ENV1=a ENV2=b" "ENV1=c" ...   ("" = defaults)"""
import argparse, importlib, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
rt = importlib.import_module("raytracing-course-hw_amd")
import gen_synth_room

ap = argparse.ArgumentParser()
ap.add_argument("--spp", type=int, default=64)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--shards", type=int, default=1)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--counters", action="store_true")
ap.add_argument("settings", nargs="*", default=[""])
a = ap.parse_args()
gltf, _ = gen_synth_room.generate(tempfile.mkdtemp(), 64, 50, 43)
sd = rt.load_gltf(gltf)
scene = rt.Scene(sd)
p = rt.make_params(a.width, a.height, a.spp, shard_index=0, shard_count=a.shards, tile=32, flags=rt.RT_FLAG_COUNTERS if a.counters else 0)
n = rt.lib.rt_output_elems(p)
out = torch.zeros(n, dtype=torch.float32, device="cuda")
out8 = torch.zeros(n, dtype=torch.uint8, device="cuda")
for setting in a.settings:
    keys = []
    for kv in setting.split():
        k, v = kv.split("=", 1); os.environ[k] = v; keys.append(k)
    best = None
    for _ in range(a.reps):
        st = scene.render_device(p, out.data_ptr(), out8.data_ptr())
        best = st if best is None or st.kernel_ms < best.kernel_ms else best
    print(f"[{setting or 'defaults'}] {a.width}x{a.height}x{a.spp} shard 1/{a.shards}: {best.kernel_ms:.1f} ms, {best.samples / best.kernel_ms / 1e3:.1f} Msamples/s, pipeline {best.pipeline}, "
          f"queries {best.closest_hit_queries}+{best.light_pdf_queries}, exact {best.exact_closest_hits}+{best.exact_light_sums}", flush=True)
    for k in keys:
        os.environ.pop(k, None)
scene.close()
