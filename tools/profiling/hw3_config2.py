import sys, importlib, os
sys.path.insert(0, os.getcwd())
rt = importlib.import_module("raytracing-course-hw_amd")
sd, w, h, spp, depth = rt.load_txt(os.path.join(os.getcwd(), "tests", "golden", "scenes", "txt", "hw3_practice3_5_800x600x64.txt"), rt.RT_INTEGRATOR_HW3)
scene = rt.Scene(sd)
best = 1e9
for _ in range(5):
    rgb, _, st = scene.render(w, h, spp, integrator=rt.RT_INTEGRATOR_HW3, ray_depth=depth, want_rgb8=False)
    best = min(best, st.kernel_ms)
print(f"hw3 config 2 ({w}x{h}x{spp}): best of 5 {best:.2f} ms = {w * h * spp / best / 1e3:.0f} Msamples/s")
