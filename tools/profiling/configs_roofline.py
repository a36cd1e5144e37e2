#!/usr/bin/env python3
"""Roofline figures for BASELINE.json configs[1] (hw3 practice3_5, 800x600x64) and configs[2] (hw6 practice6_2, 1024x1024x256) on
one MI355X: Msamples/s of the full-size render and the algorithmic bytes of SURVEY.md 8(d) against the 8 TB/s HBM peak.

    configs[1]: B = S * N_prim * 44 B with S = ray_depth queries per sample (closed box: every path runs to the depth limit)
    configs[2]: B = S [(ceil(log2 N_tri) + 1) 32 + 36 + 32] + P [(ceil(log2 N_light) + 1) 32 + 36] + 12 / spp,
                S / P = closest-hit / light-pdf queries per camera sample, counted by the timed render

Both kernels work out of registers / LDS / L2 (8 primitives; a 9 MB scene), so "fraction of the HBM roofline" is nominal, as for
the headline.  usage: python tools/profiling/configs_roofline.py [out.json]"""
import importlib
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
rt = importlib.import_module("raytracing-course-hw_amd")
import pin_cases

out = {}
# ---- configs[1]
sd, w, h, spp, depth = rt.load_txt(os.path.join(ROOT, "tests", "golden", "scenes", "txt", "hw3_practice3_5_800x600x64.txt"), rt.RT_INTEGRATOR_HW3)
scene = rt.Scene(sd)
best = min(scene.render(w, h, spp, integrator=rt.RT_INTEGRATOR_HW3, ray_depth=depth, want_rgb8=False)[2].kernel_ms for _ in range(5))
scene.close()
n_prim = len(sd.primitives)
b = depth * n_prim * 44.0
rate = w * h * spp / best / 1e3
out["configs[1] hw3 practice3_5 800x600x64"] = {"kernel": "render_hw3_kernel", "kernel_ms": round(best, 3), "msamples_per_s": round(rate, 1),
                                                "queries_per_sample": depth, "primitives": n_prim, "bytes_per_sample": b,
                                                "achieved_GBps": round(b * rate * 1e6 / 1e9, 1), "frac_of_8TBps": round(b * rate * 1e6 / 8e12, 4)}
# ---- configs[2]
sd = pin_cases.load_hw6("practice6_2")
scene = rt.Scene(sd)
info = scene.info()
st = min((scene.render(1024, 1024, 256, integrator=rt.RT_INTEGRATOR_HW6, want_rgb8=False)[2] for _ in range(3)), key=lambda t: t.kernel_ms)
s_bar, p_bar = st.closest_hit_queries / st.samples, st.light_pdf_queries / st.samples   # counted by the timed render itself
os.environ["RTAMD_KERNEL"] = "mega"
mega = scene.render(1024, 1024, 256, integrator=rt.RT_INTEGRATOR_HW6, want_rgb8=False)[2]
del os.environ["RTAMD_KERNEL"]
scene.close()
q = (math.ceil(math.log2(info.n_triangles)) + 1) * 32 + 36 + 32
lq = (math.ceil(math.log2(max(2, info.n_lights))) + 1) * 32 + 36
b = s_bar * q + p_bar * lq + 12.0 / 256
rate = st.samples / st.kernel_ms / 1e3
out["configs[2] hw6 practice6_2 1024x1024x256"] = {"kernel": "p6_persistent_kernel<false>" if st.pipeline == rt.RT_PIPELINE_PERSISTENT else "render_hw6_kernel<true>",
                                                   "launches": int(st.launches), "megakernel_ms (RTAMD_KERNEL=mega, render_hw6_kernel<true>)": round(mega.kernel_ms, 1), "kernel_ms": round(st.kernel_ms, 1), "msamples_per_s": round(rate, 2),
                                                   "s_bar": round(s_bar, 3), "p_bar": round(p_bar, 3), "triangles": int(info.n_triangles), "lights": int(info.n_lights),
                                                   "bytes_per_closest_hit_query": q, "bytes_per_light_query": lq, "bytes_per_sample": round(b, 1),
                                                   "achieved_GBps": round(b * rate * 1e6 / 1e9, 1), "frac_of_8TBps": round(b * rate * 1e6 / 8e12, 4)}
print(json.dumps(out, indent=1))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
