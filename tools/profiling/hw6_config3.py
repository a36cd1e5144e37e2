import sys, importlib, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
rt = importlib.import_module("raytracing-course-hw_amd")
import pin_cases
sd = pin_cases.load_hw6("practice6_2")
scene = rt.Scene(sd)
rgb, _, st = scene.render(1024, 1024, 256, integrator=rt.RT_INTEGRATOR_HW6, want_rgb8=False)
print("hw6 config 3 (practice6_2 1024x1024x256):", round(st.kernel_ms, 1), "ms", round(1024*1024*256/st.kernel_ms/1e3, 2), "Msamples/s")
