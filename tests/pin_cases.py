"""Deterministic input sets for pinning the CPU oracle against the compiled reference.

Each case is evaluated once through the reference harnesses (tests/golden/make_goldens.py, only where
/root/reference exists) and stored in tests/golden/pins_*.npz; tests/test_oracle_pins.py evaluates the
same inputs through the oracle and demands bit-identical outputs."""
import importlib
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = os.path.join(ROOT, "tests", "golden", "scenes")
rt = importlib.import_module("raytracing-course-hw_amd")


def unit(v):
    v = np.asarray(v, np.float64)
    return (v / np.linalg.norm(v, axis=-1, keepdims=True)).astype(np.float32)


def load_sphere():
    return rt.load_gltf(os.path.join(SCENES, "hw8_sphere", "sphere_emissive.gltf"))


def load_hw7(name):
    """hw7 example scene with hw7's material rule max(roughness, 0.04) (hw7/src/sceneio.cpp:168) applied."""
    sd = rt.load_gltf(os.path.join(SCENES, "hw7", name + ".gltf"))
    for i in range(sd.n_materials):
        sd.materials[i].roughness_factor = max(sd.materials[i].roughness_factor, np.float32(0.04))
    return sd


def as_hw7(sd):
    """Reinterpret a scene the way hw7 would load it: roughness floor, no textures."""
    for i in range(sd.n_materials):
        m = sd.materials[i]
        m.roughness_factor = max(m.roughness_factor, np.float32(0.04))
        m.base_color_texture = m.emissive_texture = m.metallic_roughness_texture = m.normal_texture = -1
    return sd


def random_triangle_scene(n=400, seed=7, n_emissive_mats=2):
    """Triangle soup with duplicated/coplanar triangles (exact ties in t) and several emissive ones."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(-2, 2, (n, 1, 3))
    tri = (c + rng.normal(0, 0.35, (n, 3, 3))).astype(np.float32)
    tri[n // 2: n // 2 + 20] = tri[:20]  # exact duplicates -> ties in t, resolved by figure order
    nrm = unit(rng.normal(0, 1, (n, 3, 3)))
    tan = np.concatenate([unit(rng.normal(0, 1, (n, 3, 3))), np.sign(rng.normal(0, 1, (n, 3, 1))).astype(np.float32)], axis=2)
    uv = rng.uniform(-1, 2, (n, 3, 2)).astype(np.float32)
    mats = []
    for i in range(6):
        m = rt.rt_material()
        m.base_color = tuple(rng.uniform(0.1, 1, 3).astype(np.float32))
        m.emission = tuple((rng.uniform(0.5, 4, 3) if i < n_emissive_mats else np.zeros(3)).astype(np.float32))
        m.metallic_factor = float(np.float32(rng.uniform(0, 1)))
        m.roughness_factor = float(np.float32(rng.uniform(0.05, 1)))
        m.base_color_texture = m.emissive_texture = m.metallic_roughness_texture = m.normal_texture = -1
        mats.append(m)
    mat_idx = rng.integers(0, 6, n).astype(np.uint32)
    cam = rt.rt_camera()
    cam.position, cam.right, cam.up, cam.forward = (0, 0, 6), (1, 0, 0), (0, 1, 0), (0, 0, -1)
    cam.fov_y = 0.9
    return rt.SceneData(tri.reshape(n, 9), uv.reshape(n, 6), nrm.reshape(n, 9), tan.reshape(n, 12), mat_idx, mats, camera=cam)


def rays_for(sd, n, seed):
    """Rays aimed at random points of random triangles, from outside and from surface-like origins."""
    rng = np.random.default_rng(seed)
    tris = sd.positions.reshape(-1, 3, 3)
    pick = rng.integers(0, len(tris), n)
    w = rng.dirichlet((1, 1, 1), n).astype(np.float32)
    target = (tris[pick] * w[:, :, None]).sum(axis=1)
    origin = np.where(rng.uniform(size=(n, 1)) < 0.5, rng.uniform(-6, 6, (n, 3)),
                      (tris[rng.integers(0, len(tris), n)].mean(axis=1) + rng.normal(0, 1e-3, (n, 3)))).astype(np.float32)
    d = unit(target - origin)
    d[: n // 10] = unit(rng.normal(0, 1, (n // 10, 3)))  # some arbitrary directions (misses)
    return origin, d


def eval_functions(impl, sd, seed):
    """impl: Hw8Oracle or Ref8.  Returns dict of output arrays."""
    o, d = rays_for(sd, 600, seed)
    hit_idx = np.zeros(len(o), np.int64)
    hit_val = np.zeros((len(o), 14), np.float32)
    order = impl.figure_order().astype(np.int64)
    for i in range(len(o)):
        idx, val = impl.closest_hit(o[i], d[i])
        hit_idx[i] = order[idx] if idx >= 0 else -1  # LOAD-order index
        hit_val[i] = val if idx >= 0 else 0
    lo, ld = rays_for(sd, 400, seed + 1)
    lpdf = np.array([impl.light_pdf(lo[i], ld[i]) for i in range(len(lo))], np.float32)
    rng = np.random.default_rng(seed + 2)
    k = 500
    xs = rng.uniform(-2, 2, (k, 3)).astype(np.float32)
    ns = unit(rng.normal(0, 1, (k, 3)))
    ns[:20] = np.array([0, 0, 1], np.float32)
    ns[20:40] = np.array([0, 0, -1], np.float32)
    vs = unit(rng.normal(0, 1, (k, 3)))
    vs = np.where((np.sum(vs * ns, axis=1, keepdims=True) > 0) & (rng.uniform(size=(k, 1)) < 0.8), -vs, vs).astype(np.float32)
    alphas = (rng.uniform(0.08, 1, k) ** 2).astype(np.float32)
    mix = np.stack([impl.mix_sample_pdf(1000 + i, xs[i], ns[i], vs[i], float(alphas[i])) for i in range(k)])
    return {"figure_order": order, "hit_idx": hit_idx, "hit_val": hit_val, "light_pdf": lpdf, "mix": mix}


def brdf_inputs(seed=5, k=800):
    rng = np.random.default_rng(seed)
    return dict(base_metallic=rng.choice([0.0, 0.3, 1.0], k).astype(np.float32), base_color=rng.uniform(0, 1, (k, 3)).astype(np.float32),
                l=unit(rng.normal(0, 1, (k, 3))), v=unit(rng.normal(0, 1, (k, 3))), n=unit(rng.normal(0, 1, (k, 3))),
                color=rng.uniform(0, 1, (k, 3)).astype(np.float32), metallic=rng.choice([0.0, 0.5, 1.0], k).astype(np.float32),
                alpha=(rng.uniform(0.08, 1, k) ** 2).astype(np.float32))


def tonemap_inputs(seed=6, k=3000):
    rng = np.random.default_rng(seed)
    x = np.concatenate([rng.uniform(0, 1.5, (k, 3)), rng.exponential(2.0, (k // 3, 3)), np.zeros((1, 3)), np.full((1, 3), 1e-8)])
    return x.astype(np.float32)


def load_hw6(name):
    return rt.load_gltf(os.path.join(SCENES, "hw6", name + ".gltf"), rt.RT_INTEGRATOR_HW6)


def hw6_soup(n=300, seed=17):
    """Triangle soup with all three hw6 material kinds (dielectric recursion included) and emissive triangles."""
    sd = random_triangle_scene(n, seed, n_emissive_mats=1)
    kinds = [0, 0, 1, 2, 2, 0]  # RT_MAT_DIFFUSE / METALLIC / DIELECTRIC
    for i in range(sd.n_materials):
        sd.materials[i].kind = kinds[i]
        sd.materials[i].ior = 1.5
    return rt.SceneData(sd.positions, None, None, None, sd.material_index, list(sd.materials)[:sd.n_materials], camera=sd.camera, bg=(0.1, 0.2, 0.3))


HW6_CASES = {"practice6_1": (lambda: load_hw6("practice6_1"), 48, 36, 4), "practice6_2": (lambda: load_hw6("practice6_2"), 20, 20, 2),
             "hw6_soup": (hw6_soup, 48, 40, 8)}


def loader_case(tmpdir):
    """A small glTF with a three-level TRS chain (non-uniform scale, general rotations) on one mesh and a `matrix` node on
    another; returns (gltf path, per-mesh dicts with the raw vertex data and the node chain)."""
    import json
    rng = np.random.default_rng(99)
    n = 24
    pos = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    nrm = unit(rng.normal(0, 1, (n, 3)))
    tan = np.concatenate([unit(rng.normal(0, 1, (n, 3))), np.ones((n, 1), np.float32)], axis=1).astype(np.float32)
    uv = rng.uniform(0, 1, (n, 2)).astype(np.float32)
    idx = np.arange(n, dtype=np.uint16)
    blob = b"".join(a.tobytes() for a in (pos, nrm, uv, tan, idx))
    offs = np.cumsum([0, pos.nbytes, nrm.nbytes, uv.nbytes, tan.nbytes])
    views = [{"buffer": 0, "byteOffset": int(offs[i]), "byteLength": int(a.nbytes)} for i, a in enumerate((pos, nrm, uv, tan, idx))]
    acc = [{"bufferView": 0, "componentType": 5126, "count": n, "type": "VEC3"}, {"bufferView": 1, "componentType": 5126, "count": n, "type": "VEC3"},
           {"bufferView": 2, "componentType": 5126, "count": n, "type": "VEC2"}, {"bufferView": 3, "componentType": 5126, "count": n, "type": "VEC4"},
           {"bufferView": 4, "componentType": 5123, "count": n, "type": "SCALAR"}]
    def q(axis, ang):
        a = unit(np.array(axis, np.float64)) * np.sin(ang / 2)
        return [float(np.float32(a[0])), float(np.float32(a[1])), float(np.float32(a[2])), float(np.float32(np.cos(ang / 2)))]
    chain = [{"translation": [0.5, -1.25, 2.0], "rotation": q((0, 1, 0), 0.7), "scale": [1.5, 1.5, 1.5]},
             {"translation": [-0.3, 0.1, 0.2], "rotation": q((1, 2, 3), 1.1), "scale": [0.5, 2.0, 1.25]},
             {"translation": [0.0, 0.75, -0.5], "rotation": q((-1, 0.5, 0.2), 2.3), "scale": [1.0, 0.8, 1.7]}]
    m = np.eye(4)
    m[:3, :3] = np.array([[0.9, -0.2, 0.1], [0.3, 1.1, -0.4], [0.05, 0.2, 0.7]])
    m[:3, 3] = [1.0, 2.0, -3.0]
    matrix = [float(np.float32(x)) for x in m.T.reshape(-1)]  # column-major
    prim = {"attributes": {"POSITION": 0, "NORMAL": 1, "TEXCOORD_0": 2, "TANGENT": 3}, "indices": 4, "material": 0}
    nodes = [dict(chain[0], children=[1]), dict(chain[1], children=[2]), dict(chain[2], mesh=0), {"matrix": matrix, "mesh": 1},
             {"camera": 0, "translation": [0, 0, 5]}]
    g = {"asset": {"version": "2.0"}, "nodes": nodes, "meshes": [{"primitives": [prim]}, {"primitives": [prim]}],
         "materials": [{"pbrMetallicRoughness": {"metallicFactor": 0}}], "accessors": acc, "bufferViews": views,
         "buffers": [{"byteLength": len(blob), "uri": "loader_case.bin"}], "images": [], "textures": [],
         "cameras": [{"type": "perspective", "perspective": {"yfov": 0.8}}]}
    os.makedirs(tmpdir, exist_ok=True)
    open(os.path.join(tmpdir, "loader_case.bin"), "wb").write(blob)
    path = os.path.join(tmpdir, "loader_case.gltf")
    json.dump(g, open(path, "w"))
    flat = lambda c: np.array(c["translation"] + c["rotation"] + c["scale"], np.float32)
    return path, dict(pos=pos, nrm=nrm, tan=tan, chain=np.stack([flat(c) for c in chain]), matrix=np.array(matrix, np.float32))


# hw2 scenes pinned through the reference's own loader + getPixel (float radiance) and program (PPM md5).
HW2_CASES = ("hw2_sample_166x128", "hw2_glass_stack")

# hw4 scenes (reference practice scenes at 64x48x8 and a scene with two box lights + an ellipsoid light), pinned in the
# reference's sequential single-engine order.
HW4_CASES = ("hw4_practice3_3_64x48x8", "hw4_practice3_5_64x48x8", "hw4_box_and_ellipsoid_lights")

# hw5 scenes: a reference practice scene through the hw5 grammar, and tools/gen_hw5_fixture.py's mixed-figure scene
# (48+2 TRIANGLE figures with own position/rotation, box / ellipsoid / triangle lights, three planes).
HW5_CASES = ("hw5_practice3_5_64x48x8", "hw5_mixed_figures")
