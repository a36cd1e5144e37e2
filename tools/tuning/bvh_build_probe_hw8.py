#!/usr/bin/env python3
"""Headline scene (synth_room_v1, 268,816 triangles) with the scene tree built on the GPU (RT_BUILD_DEVICE_BVH) vs the host replay of the
reference's builder: scene preparation time, tree shape, node visits per sample, Msamples/s (throughput mode K=8 and replay mode)."""
import argparse, importlib, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
rt = importlib.import_module("raytracing-course-hw_amd")
import gen_synth_room
ap = argparse.ArgumentParser()
ap.add_argument("--spp", type=int, default=64)
a = ap.parse_args()
gltf, _ = gen_synth_room.generate(tempfile.mkdtemp(), 64, 50, 43)
sd = rt.load_gltf(gltf)
for flags in (rt.RT_BUILD_DEVICE_BVH, 0):
    rt.Scene(sd, build_flags=flags).close()
    t0 = time.time(); scene = rt.Scene(sd, build_flags=flags); t1 = time.time()
    i = scene.info()
    print(f"[{'device' if flags else 'host'} tree] create {1e3 * (t1 - t0):.1f} ms (host prep {i.prep_ms:.1f}, upload+build {i.upload_ms:.1f}, GPU build {i.bvh_build_ms:.2f}); nodes {i.n_bvh_nodes}, depth {i.bvh_depth}", flush=True)
    _, _, st = scene.render(480, 270, 8, want_rgb8=False, counters=True)
    print(f"    480x270x8 counted: node visits / sample {st.node_visits / st.samples:.1f}, triangle tests / sample {st.triangle_tests / st.samples:.1f}, queries / sample {st.closest_hit_queries / st.samples:.2f}+{st.light_pdf_queries / st.samples:.2f}", flush=True)
    for k in (8, 0):
        best = min(scene.render(1920, 1080, a.spp, want_rgb8=False, want_float=False, sample_streams=k)[2].kernel_ms for _ in range(2))
        print(f"    1920x1080x{a.spp} {'throughput mode K=8' if k else 'replay mode'}: {best:.1f} ms = {1920 * 1080 * a.spp / best / 1e3:.1f} Msamples/s", flush=True)
    scene.close()
