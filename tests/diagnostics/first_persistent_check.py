"""First contact of the persistent pipeline with the GPU: small renders against the oracle, each pipeline side by side."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
rt = importlib.import_module("raytracing-course-hw_amd")
import oracle_lib, pin_cases

def check(tag, sd, w, h, spp, **kw):
    ref, ref8, _ = oracle_lib.Hw8Oracle(sd).render(w, h, spp, **{k: v for k, v in kw.items() if k == "ray_depth"})
    for kernel in ("persistent", "wavefront"):
        os.environ["RTAMD_KERNEL"] = kernel
        scene = rt.Scene(sd)
        t = time.time()
        rgb, rgb8, st = scene.render(w, h, spp, **kw)
        scene.close()
        rmse = float(np.sqrt(np.nanmean((rgb.astype(np.float64) - ref) ** 2)))
        nbad = int((rgb.view(np.uint32) != ref.view(np.uint32)).any(axis=2).sum())
        print(f"{tag}[{kernel}] {w}x{h}x{spp}: pipeline {st.pipeline} launches {st.launches} kernel {st.kernel_ms:.2f} ms rmse {rmse:.3e} px_not_bit_exact {nbad} "
              f"queries {st.closest_hit_queries}+{st.light_pdf_queries} exact {st.exact_closest_hits}+{st.exact_light_sums} wall {time.time() - t:.2f}s", flush=True)

check("sphere", pin_cases.load_sphere(), 64, 64, 4)
check("sphere", pin_cases.load_sphere(), 200, 120, 16)
check("soup", pin_cases.random_triangle_scene(n=600, seed=3), 96, 72, 6)
import gen_synth_room, tempfile
path, _ = gen_synth_room.generate(tempfile.mkdtemp(), 8, 12, 9, tex_size=64)
check("room", rt.load_gltf(path), 160, 90, 12)
