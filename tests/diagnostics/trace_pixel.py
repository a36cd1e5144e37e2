#!/usr/bin/env python3
"""Query-level comparison of one pixel of the headline scene: the hit records the GPU's shader consumed (persistent pipeline, counting
build, RTAMD_TRACE_PIXEL) against the closest-hit queries of the CPU oracle's replay of the same pixel, in order.
(diagnostic; the oracle is test infrastructure)  usage: trace_pixel.py x y [--spp N] [--env "RTAMD_HOST_BVH=1 ..."]"""
import argparse, ctypes as C, importlib, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
rt = importlib.import_module("raytracing-course-hw_amd")
import gen_synth_room, oracle_lib
ap = argparse.ArgumentParser()
ap.add_argument("x", type=int); ap.add_argument("y", type=int)
ap.add_argument("--spp", type=int, default=32)
ap.add_argument("--env", default="")
a = ap.parse_args()
gltf, _ = gen_synth_room.generate(tempfile.mkdtemp(), 64, 50, 43)
sd = rt.load_gltf(gltf)
W, H = 1920, 1080
for kv in a.env.split():
    k, v = kv.split("=", 1); os.environ[k] = v
out = os.path.join(tempfile.mkdtemp(), "trace.bin")
os.environ["RTAMD_TRACE_PIXEL"] = f"{a.x},{a.y}"; os.environ["RTAMD_TRACE_OUT"] = out
os.environ["RTAMD_PT_NO_REBALANCE"] = "1"   # one launch: one dump
sc = rt.Scene(sd)
rgb, _, st = sc.render(W, H, a.spp, want_rgb8=False, counters=True)
sc.close()
raw = np.fromfile(out, dtype=np.float32)
n = int(raw[:1].view(np.uint32)[0])
rec = raw[4:4 + 16 * n].reshape(n, 16)
hitw = rec[:, 11].copy().view(np.uint32); packed = rec[:, 15].copy().view(np.uint32)
sample, depth = (packed >> 4) & 0x01FFFFFF if False else None, None
# packed word of rt_wavefront.h: depth in the low 4 bits, sample index above (WF_SAMPLE_MASK), verified flag in bit 31
depth = packed & 15; verified = packed >> 31
entries = {}
order = []
for i in range(n):
    key = (tuple(rec[i, 0:3]), tuple(rec[i, [3, 4, 5]]))
    if key not in entries: order.append(key)
    entries[key] = i                                   # a flagged hit comes back verified: keep the last record of a ray
gpu = [entries[k] for k in order]
flagged = [i for i in range(n) if hitw[i] != 0xFFFFFFFF and verified[i]]
print(f"{len(flagged)} of {n} consumed records come from the exact walk")
for i in flagged[:8]:
    print(f"    flagged: o {rec[i, 0:3]} d {rec[i, 3:6]} t {rec[i, 8]!r} figure {hitw[i] & 0x00FFFFFF}")
orc = oracle_lib.Hw8Oracle(sd)
L = oracle_lib.lib()
L.rto_hw8_trace_pixel.argtypes = [C.c_void_p] + [C.c_int] * 6 + [C.c_void_p, C.c_int]
buf = np.zeros(12 * 4096, np.float32)
m = L.rto_hw8_trace_pixel(orc._h, W, H, a.spp, 0, a.x, a.y, buf.ctypes.data, buf.size) // 12
ref = buf[:12 * m].reshape(m, 12)
ref_px, _, _ = orc.render(W, H, a.spp, rect=(a.x, a.y, 1, 1))
print(f"pixel ({a.x},{a.y}) {a.spp} spp: GPU {rgb[a.y, a.x]} oracle {ref_px.reshape(3)}; GPU consumed {len(gpu)} rays ({n} records), oracle made {m} queries")
# match by ray (origin, direction): the GPU consumes hits in path order per sample, but samples of one pixel are sequential, so the order is the oracle's
bad = 0
gmap = {order[j]: gpu[j] for j in range(len(gpu))}
for j in range(m):
    key = (tuple(ref[j, 0:3]), tuple(ref[j, 3:6]))
    if key not in gmap:
        print(f"query {j}: ray not among the GPU's consumed hits (the paths diverged before): o {ref[j, 0:3]} d {ref[j, 3:6]}"); bad += 1
        if bad > 3: break
        continue
    i = gmap[key]
    w = int(hitw[i]); miss = w == 0xFFFFFFFF
    g_idx = -1 if miss else (w & 0x00FFFFFF); g_t = -1.0 if miss else float(rec[i, 8])
    same = (g_idx == int(ref[j, 7])) and (miss or np.float32(g_t) == np.float32(ref[j, 6]))
    if not same:
        bad += 1
        print(f"query {j} (remaining depth {int(ref[j, 11])}): o {ref[j, 0:3]} d {ref[j, 3:6]}")
        print(f"    oracle: t {ref[j, 6]!r} figure {int(ref[j, 7])} inside {int(ref[j, 8])} uv {ref[j, 9:11]}")
        print(f"    GPU   : t {g_t!r} figure {g_idx} gap code {(w >> 24) & 63 if not miss else 0} inside {(w >> 30) & 1 if not miss else 0} u,v {rec[i, 9:11]} verified {int(verified[i])}")
        if bad > 3: break
print("mismatching queries:", bad)
