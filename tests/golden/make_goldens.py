"""Regenerates tests/golden/pins_*.npz from the compiled reference (oracle/_ref, needs /root/reference).
Run:  python tests/golden/make_goldens.py      (after `make -C oracle`)"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib  # noqa: E402
import pin_cases  # noqa: E402


def main():
    L8 = C.CDLL(oracle_lib.ref_path("libref_hw8.so"))
    L8.ref8_brdf.argtypes = [C.c_float] + [C.c_void_p] * 5 + [C.c_float, C.c_float, C.c_void_p]
    L8.ref8_tonemap.argtypes = [C.c_void_p, C.c_void_p]
    # per-function pins through the reference's hw8 headers + primitives.cpp + color.cpp
    for name, sd, seed in (("sphere", pin_cases.load_sphere(), 11), ("soup", pin_cases.random_triangle_scene(), 23)):
        out = pin_cases.eval_functions(oracle_lib.Ref8(sd), sd, seed)
        np.savez_compressed(os.path.join(HERE, f"pins_hw8_functions_{name}.npz"), **out)
        print(name, {k: v.shape for k, v in out.items()})
    bi = pin_cases.brdf_inputs()
    br = np.stack([oracle_lib.brdf(L8, "ref8_", float(bi["base_metallic"][i]), bi["base_color"][i], bi["l"][i], bi["v"][i], bi["n"][i],
                                   bi["color"][i], float(bi["metallic"][i]), float(bi["alpha"][i])) for i in range(len(bi["alpha"]))])
    ti = pin_cases.tonemap_inputs()
    tm = np.stack([oracle_lib.tonemap(L8, "ref8_tonemap", ti[i]) for i in range(len(ti))])
    np.savez_compressed(os.path.join(HERE, "pins_hw8_brdf_tonemap.npz"), brdf=br, tonemap=tm)
    # whole-integrator pins through the reference's hw7 scene.cpp (Scene::getPixel)
    cases = {"practice7_1": (pin_cases.load_hw7("practice7_1"), 48, 48, 8), "practice7_4": (pin_cases.load_hw7("practice7_4"), 48, 48, 8),
             "sphere_as_hw7": (pin_cases.as_hw7(pin_cases.load_sphere()), 40, 40, 6), "soup_as_hw7": (pin_cases.as_hw7(pin_cases.random_triangle_scene()), 40, 32, 6)}
    out = {}
    for name, (sd, w, h, spp) in cases.items():
        rgb, rgb8, _ = oracle_lib.Ref7(sd).render(w, h, spp)
        out[name + "_rgb"], out[name + "_rgb8"] = rgb, rgb8
        print(name, rgb.shape, float(rgb.mean()))
    np.savez_compressed(os.path.join(HERE, "pins_hw7_render.npz"), **out)
    # whole-integrator pins through the reference's hw6 scene.cpp (dielectric recursion, Mix{Cosine, lights})
    out = {}
    for name, (mk, w, h, spp) in pin_cases.HW6_CASES.items():
        rgb, rgb8, _ = oracle_lib.Ref6(mk()).render(w, h, spp)
        out[name + "_rgb"], out[name + "_rgb8"] = rgb, rgb8
        print(name, rgb.shape, float(rgb.mean()))
    np.savez_compressed(os.path.join(HERE, "pins_hw6_render.npz"), **out)
    # loader arithmetic through the reference's own transition.h (node chains, inverse-transpose normals, tangents)
    import tempfile as _tf
    with _tf.TemporaryDirectory() as td:
        _, lc = pin_cases.loader_case(td)
    L8.ref8_transform_chain.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 6
    out = {}
    n = lc["pos"].shape[0]
    for tag, levels, chain, matrix in (("chain", 3, lc["chain"], None), ("matrix", 1, np.array([0, 0, 0, 0, 0, 0, 1, 1, 1, 1], np.float32), lc["matrix"])):
        op, on, ot = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
        ch = np.ascontiguousarray(chain, np.float32)
        L8.ref8_transform_chain(levels, ch.ctypes.data, matrix.ctypes.data if matrix is not None else None, n, lc["pos"].ctypes.data,
                                lc["nrm"].ctypes.data, lc["tan"].ctypes.data, op.ctypes.data, on.ctypes.data, ot.ctypes.data)
        out[tag + "_pos"], out[tag + "_nrm"], out[tag + "_tan"] = op, on, ot
    np.savez_compressed(os.path.join(HERE, "pins_loader_transforms.npz"), **out)
    print("loader transforms", {k: v.shape for k, v in out.items()})
    # byte-level pins through the reference's own unmodified CLI programs (hw1, hw3)
    import hashlib
    import subprocess
    import tempfile
    txt = os.path.join(HERE, "scenes", "txt")
    out = {}
    for name, prog in (("hw1_sample", "hw1_main"), ("hw1_sample_256", "hw1_main"), ("hw3_practice3_5_64x48x8", "hw3_main"), ("hw3_mixed_materials", "hw3_main")):
        with tempfile.TemporaryDirectory() as td:
            ppm = os.path.join(td, "o.ppm")
            subprocess.run([oracle_lib.ref_path(prog), os.path.join(txt, name + ".txt"), ppm], check=True, stderr=subprocess.DEVNULL)
            data = open(ppm, "rb").read()
        out[name + "_md5"] = np.frombuffer(hashlib.md5(data).hexdigest().encode(), np.uint8)
        if len(data) < 100000:
            out[name + "_ppm"] = np.frombuffer(data, np.uint8)
        print(name, hashlib.md5(data).hexdigest(), len(data))
    np.savez_compressed(os.path.join(HERE, "pins_txt_programs.npz"), **out)
    # hw2: float radiance through the reference's own loader + Scene::getPixel, and the program's PPM md5
    out = {}
    for name in pin_cases.HW2_CASES:
        path = os.path.join(txt, name + ".txt")
        out[name + "_rgb"] = oracle_lib.RefTxt(2, path).render()
        with tempfile.TemporaryDirectory() as td:
            ppm = os.path.join(td, "o.ppm")
            subprocess.run([oracle_lib.ref_path("hw2_main"), path, ppm], check=True, stderr=subprocess.DEVNULL)
            out[name + "_md5"] = np.frombuffer(hashlib.md5(open(ppm, "rb").read()).hexdigest().encode(), np.uint8)
        print(name, out[name + "_rgb"].shape, float(out[name + "_rgb"].mean()), bytes(out[name + "_md5"]).decode())
    with tempfile.TemporaryDirectory() as td:
        ppm = os.path.join(td, "o.ppm")
        subprocess.run([oracle_lib.ref_path("hw2_main"), os.path.join(txt, "hw2_sample.txt"), ppm], check=True, stderr=subprocess.DEVNULL)
        out["hw2_sample_md5"] = np.frombuffer(hashlib.md5(open(ppm, "rb").read()).hexdigest().encode(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "pins_hw2_render.npz"), **out)
    # hw4: float radiance of the reference's sequential stream (fresh process per scene: the engine is a file-static)
    out = {}
    for name in pin_cases.HW4_CASES:
        out[name + "_rgb"] = oracle_lib.ref_txt_render_fresh(4, os.path.join(txt, name + ".txt"))
        print(name, out[name + "_rgb"].shape, float(out[name + "_rgb"].mean()))
    np.savez_compressed(os.path.join(HERE, "pins_hw4_render.npz"), **out)
    # hw5: float radiance through the reference's own loader + Scene::getPixel(rng(y*W+x)), and the program's PPM md5
    out = {}
    for name in pin_cases.HW5_CASES:
        path = os.path.join(txt, name + ".txt")
        out[name + "_rgb"] = oracle_lib.RefTxt(5, path).render()
        with tempfile.TemporaryDirectory() as td:
            ppm = os.path.join(td, "o.ppm")
            subprocess.run([oracle_lib.ref_path("hw5_main"), path, ppm], check=True, stderr=subprocess.DEVNULL)
            out[name + "_md5"] = np.frombuffer(hashlib.md5(open(ppm, "rb").read()).hexdigest().encode(), np.uint8)
        print(name, out[name + "_rgb"].shape, float(out[name + "_rgb"].mean()), bytes(out[name + "_md5"]).decode())
    np.savez_compressed(os.path.join(HERE, "pins_hw5_render.npz"), **out)


if __name__ == "__main__":
    main()
