"""Writes tests/golden/scenes/txt/hw5_mixed_figures.txt: a .txt scene in the hw5 grammar that exercises every figure type
(ELLIPSOID, BOX, PLANE, TRIANGLE with POSITION/ROTATION), all three light kinds (box, ellipsoid, triangle), and the three
materials.  The reference ships no hw5 scene; this one is deterministic (seeded)."""
import math
import os
import random

random.seed(20241223)
out = []
w = out.append
w("DIMENSIONS 72 54"); w("RAY_DEPTH 5"); w("SAMPLES 6"); w("BG_COLOR 0.05 0.06 0.09")
w("CAMERA_POSITION 0 1.4 7"); w("CAMERA_RIGHT 1 0 0"); w("CAMERA_UP 0 1 0"); w("CAMERA_FORWARD 0 -0.12 -1"); w("CAMERA_FOV_X 1.05")


def prim(lines):
    w("NEW_PRIMITIVE")
    for l in lines:
        w(l)


prim(["PLANE 0 1 0", "POSITION 0 -1 0", "COLOR 0.75 0.75 0.7"])
prim(["PLANE 0 0 2", "POSITION 0 0 -5", "COLOR 0.35 0.55 0.75"])          # non-unit normal: hw5 keeps it as parsed
prim(["PLANE 1 0 0", "POSITION -4.5 0 0", "COLOR 0.8 0.4 0.4"])
prim(["BOX 0.9 0.08 0.7", "POSITION 0 3.4 0.5", "ROTATION 0.06 0 0.04 0.997", "EMISSION 7 6 5"])
prim(["ELLIPSOID 0.3 0.45 0.3", "POSITION 2.6 0.2 1.2", "ROTATION 0.2 0.1 0 0.97", "EMISSION 2.5 1 0.6"])
prim(["ELLIPSOID 1.0 1.0 1.0", "POSITION -1.6 0 0.3", "COLOR 0.95 0.95 0.95", "DIELECTRIC", "IOR 1.5"])
prim(["BOX 0.55 0.8 0.55", "POSITION 1.3 -0.2 -1.2", "ROTATION 0 0.25 0 0.968", "COLOR 0.85 0.75 0.35", "METALLIC"])
prim(["ELLIPSOID 0.45 0.3 0.45", "POSITION 0.3 -0.7 2.0", "COLOR 0.8 0.3 0.3"])
# a fan of small triangles with their own positions / rotations (48 of them), four of them emissive
for i in range(48):
    a = 2 * math.pi * i / 48
    r = 2.2 + 0.6 * random.random()
    pos = (r * math.cos(a), -0.6 + 1.8 * random.random(), -1.5 + r * math.sin(a) * 0.6)
    v = [tuple(round(random.uniform(-0.35, 0.35), 4) for _ in range(3)) for _ in range(3)]
    q = [random.uniform(-0.3, 0.3) for _ in range(3)] + [0.9]
    n = math.sqrt(sum(c * c for c in q))
    q = [c / n for c in q]
    lines = ["TRIANGLE " + " ".join(f"{c}" for p in v for c in p),
             "POSITION " + " ".join(f"{c:.4f}" for c in pos),
             "ROTATION " + " ".join(f"{c:.5f}" for c in q)]
    if i % 12 == 5:
        lines.append(f"EMISSION {3 + i % 5} {2 + i % 3} {1 + i % 7}")
    else:
        lines.append("COLOR " + " ".join(f"{random.uniform(0.2, 0.95):.3f}" for _ in range(3)))
        if i % 7 == 3:
            lines.append("METALLIC")
        if i % 11 == 4:
            lines += ["DIELECTRIC", "IOR 1.4"]
    prim(lines)
# two larger triangles forming an emissive quad without rotation
prim(["TRIANGLE -3.5 2.2 -3 -2.3 2.2 -3 -2.3 3.0 -3", "EMISSION 1.5 2.5 4"])
prim(["TRIANGLE -3.5 2.2 -3 -2.3 3.0 -3 -3.5 3.0 -3", "EMISSION 1.5 2.5 4"])
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "scenes", "txt", "hw5_mixed_figures.txt")
open(path, "w").write("\n".join(out) + "\n")
print(path, len(out), "lines")
