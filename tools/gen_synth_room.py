#!/usr/bin/env python3
"""synth_room_v1 — the synthetic stand-in for hw8/examples/room (whose room.bin is missing).

Writes <out>/synth_room.gltf + synth_room.bin + three 512x512 PNG textures, so the scene goes through
the same glTF loader as any other input.  Deterministic: every random choice comes from the LCG
x <- (1664525 x + 1013904223) mod 2^32 seeded with 20241223 (u = x / 2^32).

Default: a closed box room (12 triangles), 2 emissive ceiling quads, 64 UV spheres of 4,200 triangles each
(50 segments x 43 rings; one of them emissive) = 268,816 triangles (room.gltf has 269,966).  Sphere materials:
40 % rough dielectric, 30 % metal, 30 % textured (checker albedo + value-noise metallic-roughness + sine-bump
normal map), roughness in [0.1, 1].  Every primitive carries POSITION/NORMAL/TEXCOORD_0/TANGENT and u32
indices; nodes are TRS only; camera yfov 0.7696 (room.gltf's), meant for 16:9."""
import argparse
import json
import math
import os
import struct
import zlib

import numpy as np


class Lcg:
    def __init__(self, seed):
        self.x = seed & 0xFFFFFFFF

    def u(self):
        self.x = (1664525 * self.x + 1013904223) & 0xFFFFFFFF
        return self.x / 4294967296.0


def write_png(path, rgb):
    h, w, _ = rgb.shape
    raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def textures(n=512):
    y, x = np.mgrid[0:n, 0:n]
    check = ((x // 64 + y // 64) % 2).astype(np.float32)
    albedo = np.stack([0.85 - 0.55 * check, 0.8 - 0.35 * check, 0.75 - 0.6 * (1 - check)], axis=2)
    # value noise from a hashed 16x16 lattice, bilinear
    lat = ((np.arange(17 * 17, dtype=np.uint64) * 2654435761 % 4294967296) / 4294967296.0).reshape(17, 17).astype(np.float32)
    lat[16, :] = lat[0, :]
    lat[:, 16] = lat[:, 0]
    fx, fy = x / 32.0, y / 32.0
    ix, iy = fx.astype(int), fy.astype(int)
    tx, ty = fx - ix, fy - iy
    noise = (lat[iy, ix] * (1 - tx) + lat[iy, ix + 1] * tx) * (1 - ty) + (lat[iy + 1, ix] * (1 - tx) + lat[iy + 1, ix + 1] * tx) * ty
    mr = np.stack([np.zeros_like(noise), 0.15 + 0.8 * noise, (noise > 0.55).astype(np.float32)], axis=2)  # g = roughness, b = metallic
    bx = 0.35 * np.cos(2 * math.pi * x / 32.0)
    by = 0.35 * np.cos(2 * math.pi * y / 48.0)
    nz = np.sqrt(np.maximum(0.0, 1 - bx * bx - by * by))
    normal = np.stack([0.5 + 0.5 * bx, 0.5 + 0.5 * by, 0.5 + 0.5 * nz], axis=2)
    to8 = lambda a: np.clip(np.rint(a * 255), 0, 255).astype(np.uint8)
    return to8(albedo), to8(mr), to8(normal)


def uv_sphere(segs, rings):
    """Unit sphere: (rings+1) x (segs+1) vertices, 2*segs*(rings-1) triangles."""
    v = np.linspace(0, 1, rings + 1)[:, None]
    u = np.linspace(0, 1, segs + 1)[None, :]
    theta, phi = math.pi * v, 2 * math.pi * u
    pos = np.stack([np.sin(theta) * np.cos(phi), np.cos(theta) * np.ones_like(phi), np.sin(theta) * np.sin(phi)], axis=2)
    tan = np.stack([-np.sin(phi) * np.ones_like(theta), np.zeros_like(theta * phi), np.cos(phi) * np.ones_like(theta)], axis=2)
    uv = np.stack([u * np.ones_like(v), v * np.ones_like(u)], axis=2)
    idx = []
    for r in range(rings):
        for s in range(segs):
            a, b = r * (segs + 1) + s, r * (segs + 1) + s + 1
            c, d = a + segs + 1, b + segs + 1
            if r != 0:
                idx += [a, b, c]
            if r != rings - 1:
                idx += [b, d, c]
    pos = pos.reshape(-1, 3).astype(np.float32)
    tan4 = np.concatenate([tan.reshape(-1, 3), np.ones((pos.shape[0], 1))], axis=1).astype(np.float32)
    return pos, pos.copy(), uv.reshape(-1, 2).astype(np.float32), tan4, np.array(idx, np.uint32)


def quad(p0, ex, ey, normal, uv_scale=1.0):
    """Quad p0 + s*ex + t*ey, two triangles, normal given (tangent = ex direction)."""
    p0, ex, ey = (np.array(a, np.float32) for a in (p0, ex, ey))
    pos = np.array([p0, p0 + ex, p0 + ex + ey, p0 + ey], np.float32)
    nrm = np.tile(np.array(normal, np.float32), (4, 1))
    uv = np.array([[0, 0], [uv_scale, 0], [uv_scale, uv_scale], [0, uv_scale]], np.float32)
    t = ex / np.linalg.norm(ex)
    tan = np.tile(np.array([t[0], t[1], t[2], 1.0], np.float32), (4, 1))
    return pos, nrm, uv, tan, np.array([0, 1, 2, 0, 2, 3], np.uint32)


def generate(out_dir, n_spheres=64, segs=50, rings=43, seed=20241223, tex_size=512, name="synth_room"):
    os.makedirs(out_dir, exist_ok=True)
    rnd = Lcg(seed)
    blob = bytearray()
    views, accessors = [], []

    def add(arr, typ, ctype):
        arr = np.ascontiguousarray(arr)
        while len(blob) % 4:
            blob.append(0)
        views.append({"buffer": 0, "byteOffset": len(blob), "byteLength": arr.nbytes})
        blob.extend(arr.tobytes())
        accessors.append({"bufferView": len(views) - 1, "componentType": ctype, "count": int(arr.shape[0]), "type": typ})
        return len(accessors) - 1

    def add_geom(pos, nrm, uv, tan, idx):
        return {"attributes": {"POSITION": add(pos, "VEC3", 5126), "NORMAL": add(nrm, "VEC3", 5126), "TEXCOORD_0": add(uv, "VEC2", 5126),
                               "TANGENT": add(tan, "VEC4", 5126)}, "indices": add(idx, "SCALAR", 5125)}

    for tname, img in zip(("albedo", "mr", "normal"), textures(tex_size)):
        write_png(os.path.join(out_dir, f"{name}_{tname}.png"), img)
    images = [{"uri": f"{name}_{t}.png"} for t in ("albedo", "mr", "normal")]
    tex = [{"sampler": 0, "source": i} for i in range(3)]
    materials = [
        {"name": "walls", "pbrMetallicRoughness": {"baseColorFactor": [0.9, 0.9, 0.9, 1], "baseColorTexture": {"index": 0}, "metallicFactor": 0, "roughnessFactor": 0.9}},
        {"name": "ceiling_light", "pbrMetallicRoughness": {"baseColorFactor": [0, 0, 0, 1], "metallicFactor": 0, "roughnessFactor": 1}, "emissiveFactor": [1, 1, 1],
         "extensions": {"KHR_materials_emissive_strength": {"emissiveStrength": 14}}},
        {"name": "lamp_sphere", "pbrMetallicRoughness": {"baseColorFactor": [0, 0, 0, 1], "metallicFactor": 0, "roughnessFactor": 1}, "emissiveFactor": [1, 0.8, 0.5],
         "extensions": {"KHR_materials_emissive_strength": {"emissiveStrength": 6}}},
    ]
    nodes, meshes = [], []
    X, Y, Z = 8.0, 6.0, 10.0  # room half-width, height, half-depth
    walls = [quad((-X, 0, Z), (2 * X, 0, 0), (0, 0, -2 * Z), (0, 1, 0), 8), quad((-X, Y, -Z), (2 * X, 0, 0), (0, 0, 2 * Z), (0, -1, 0), 8),
             quad((-X, 0, -Z), (2 * X, 0, 0), (0, Y, 0), (0, 0, 1), 4), quad((X, 0, Z), (-2 * X, 0, 0), (0, Y, 0), (0, 0, -1), 4),
             quad((-X, 0, Z), (0, 0, -2 * Z), (0, Y, 0), (1, 0, 0), 4), quad((X, 0, -Z), (0, 0, 2 * Z), (0, Y, 0), (-1, 0, 0), 4)]
    meshes.append({"primitives": [dict(add_geom(*w), material=0) for w in walls]})
    nodes.append({"mesh": 0, "name": "room"})
    lights = [quad((-5, Y - 0.01, -4), (3, 0, 0), (0, 0, 3), (0, -1, 0)), quad((2, Y - 0.01, 1), (3, 0, 0), (0, 0, 3), (0, -1, 0))]
    meshes.append({"primitives": [dict(add_geom(*q), material=1) for q in lights]})
    nodes.append({"mesh": 1, "name": "ceiling_lights"})
    sphere = add_geom(*uv_sphere(segs, rings))
    for i in range(n_spheres):
        r = 0.35 + 0.75 * rnd.u()
        c = [(-X + 1.2) + (2 * X - 2.4) * rnd.u(), r + (Y - 2 * r - 0.6) * rnd.u() * rnd.u(), (-Z + 1.2) + (2 * Z - 5.0) * rnd.u()]
        ang, kind, rough = 2 * math.pi * rnd.u(), rnd.u(), 0.1 + 0.9 * rnd.u()
        col = [0.25 + 0.7 * rnd.u(), 0.25 + 0.7 * rnd.u(), 0.25 + 0.7 * rnd.u(), 1]
        if i == 0:
            mat, r, c = 2, 0.5, [0.0, 4.2, -2.0]
        else:
            if kind < 0.4:
                m = {"pbrMetallicRoughness": {"baseColorFactor": col, "metallicFactor": 0, "roughnessFactor": rough}}
            elif kind < 0.7:
                m = {"pbrMetallicRoughness": {"baseColorFactor": col, "metallicFactor": 1, "roughnessFactor": rough}}
            else:
                m = {"pbrMetallicRoughness": {"baseColorFactor": [1, 1, 1, 1], "baseColorTexture": {"index": 0}, "metallicFactor": 1, "roughnessFactor": 1,
                                              "metallicRoughnessTexture": {"index": 1}}, "normalTexture": {"index": 2}}
            m["name"] = f"sphere_{i}"
            materials.append(m)
            mat = len(materials) - 1
        meshes.append({"primitives": [dict(sphere, material=mat)]})
        nodes.append({"mesh": len(meshes) - 1, "name": f"sphere_{i}", "translation": c, "scale": [r, r, r],
                      "rotation": [0, math.sin(ang / 2), 0, math.cos(ang / 2)]})
    nodes.append({"camera": 0, "name": "camera", "translation": [0, 2.6, Z - 0.4]})
    gltf = {"asset": {"version": "2.0", "generator": "synth_room_v1"}, "scene": 0, "scenes": [{"nodes": list(range(len(nodes)))}],
            "nodes": nodes, "cameras": [{"type": "perspective", "perspective": {"yfov": 0.7696, "znear": 0.1, "aspectRatio": 16 / 9}}],
            "materials": materials, "meshes": meshes, "textures": tex, "images": images, "samplers": [{}],
            "accessors": accessors, "bufferViews": views, "buffers": [{"byteLength": len(blob), "uri": f"{name}.bin"}],
            "extensionsUsed": ["KHR_materials_emissive_strength"]}
    with open(os.path.join(out_dir, f"{name}.bin"), "wb") as f:
        f.write(bytes(blob))
    path = os.path.join(out_dir, f"{name}.gltf")
    with open(path, "w") as f:
        json.dump(gltf, f)
    n_tris = 12 + 4 + n_spheres * 2 * segs * (rings - 1)
    return path, n_tris


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("out_dir")
    ap.add_argument("--spheres", type=int, default=64)
    ap.add_argument("--segs", type=int, default=50)
    ap.add_argument("--rings", type=int, default=43)
    a = ap.parse_args()
    p, n = generate(a.out_dir, a.spheres, a.segs, a.rings)
    print(p, n, "triangles")
