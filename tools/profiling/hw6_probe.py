import sys, importlib
import os; sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
rt = importlib.import_module("raytracing-course-hw_amd")
import pin_cases
sd = pin_cases.load_hw6("practice6_2")
scene = rt.Scene(sd)
rgb, _, st = scene.render(512, 512, 32, integrator=rt.RT_INTEGRATOR_HW6, want_rgb8=False)
print("hw6 512x512x32:", st.kernel_ms, "ms", 512*512*32/st.kernel_ms/1e3, "Msamples/s")
