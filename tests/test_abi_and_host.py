"""CPU-side checks: the C-ABI library loads and exports every symbol include/rtamd.h declares, the host
front-end (glTF/PNG loader, PPM writer, shard layout) behaves, and GPU entry points fail loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(rt):
    header = open(os.path.join(ROOT, "include", "rtamd.h")).read()
    declared = set(re.findall(r"^(?:int|void|size_t|const char \*|const rt_scene_desc \*)\s*(rt_[a-z0-9_]+)\(", header, re.M))
    assert len(declared) >= 17
    assert declared == set(rt.ABI_SYMBOLS), declared ^ set(rt.ABI_SYMBOLS)
    for name in declared:
        assert hasattr(rt.lib, name), f"librtamd.so does not export {name}"
    assert rt.lib.rt_abi_version() == 6


def test_struct_sizes_match_header(rt):
    # sizes the C side checks through struct_size; a mismatch means the ctypes mirror drifted
    assert C.sizeof(rt.rt_material) == 56 and C.sizeof(rt.rt_render_params) == 64
    assert C.sizeof(rt.rt_scene_desc) % 8 == 0 and C.sizeof(rt.rt_stats) == 96


def test_gltf_loader_sphere(rt, sphere_scene):
    sd = sphere_scene
    assert sd.positions.shape == (970, 9) and sd.n_materials == 2 and len(sd.images) == 1
    assert sd.images[0].shape == (1024, 1024, 3)
    assert sd.materials[1].emissive_texture == 0 and list(sd.materials[1].emission) == [5.0, 5.0, 5.0]  # strength 5 applied
    assert abs(sd.camera.fov_y - 1.0521329641342163) < 1e-7 and list(sd.camera.position) == [0.0, 0.0, pytest.approx(5.149219512939453)]
    n = sd.normals.reshape(-1, 3)
    assert np.allclose(np.linalg.norm(n, axis=1), 1, atol=1e-5)


def test_loader_errors_are_reported_not_crashes(rt, tmp_path):
    with pytest.raises(rt.RtError):
        rt.load_gltf(str(tmp_path / "missing.gltf"))
    bad = tmp_path / "bad.gltf"
    bad.write_text('{"buffers": [{"byteLength": 4, "uri": "nope.bin"}]}')
    with pytest.raises(rt.RtError):
        rt.load_gltf(str(bad))
    bad.write_text("{ not json")
    with pytest.raises(rt.RtError):
        rt.load_gltf(str(bad))


def test_ppm_writer_and_png_decoder(rt, tmp_path):
    img = (np.arange(5 * 7 * 3) % 251).astype(np.uint8).reshape(5, 7, 3)
    p = tmp_path / "o.ppm"
    rt.write_ppm(str(p), img)
    data = p.read_bytes()
    assert data.startswith(b"P6\n7 5\n255\n") and data[len(b"P6\n7 5\n255\n"):] == img.tobytes()  # hw8/src/sceneio.cpp:383-385
    tex = rt.decode_png(os.path.join(ROOT, "tests", "golden", "scenes", "hw8_sphere", "sphere_emission.png"))
    assert tex.shape == (1024, 1024, 3) and tex.dtype == np.uint8 and tex.max() > 0


def test_shard_layout_roundtrip(rt):
    W, H = 70, 45
    full = np.random.default_rng(0).integers(0, 255, (H, W, 3)).astype(np.uint8)
    acc = np.zeros_like(full)
    for world in (2, 3):
        acc[:] = 0
        for r in range(world):
            p = rt.make_params(W, H, 1, shard_index=r, shard_count=world, tile=16)
            n = rt.lib.rt_output_elems(p)
            tiles_x = (W + 15) // 16
            buf = np.zeros(n, np.uint8)
            st = 0
            for t in range(r, tiles_x * ((H + 15) // 16), world):
                x0, y0 = (t % tiles_x) * 16, (t // tiles_x) * 16
                tile = np.zeros((16, 16, 3), np.uint8)
                w, h = min(16, W - x0), min(16, H - y0)
                tile[:h, :w] = full[y0:y0 + h, x0:x0 + w]
                buf[st * 768:(st + 1) * 768] = tile.reshape(-1)
                st += 1
            acc += rt.unshard(p, buf)
        assert np.array_equal(acc, full)


def test_multi_device_entry_points_fail_loudly_without_a_gpu(rt, sphere_scene):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert rt.lib.rt_device_count() == 0
    with pytest.raises(rt.RtError) as e:
        rt.MultiScene(sphere_scene, [0, 1])
    assert e.value.code == rt.RT_ERR_NO_DEVICE


def test_no_gpu_means_loud_failure_not_fallback(rt, sphere_scene):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(rt.RtError) as e:
        rt.Scene(sphere_scene)
    assert e.value.code == rt.RT_ERR_NO_DEVICE


def test_gltf_interleaved_vertex_buffer_and_u8_indices_load_like_the_tight_layout(rt, tmp_path):
    """Extension beyond the reference: one interleaved bufferView with byteStride (pos|nrm|uv|tan = 48 B per
    vertex) and 8-bit indices must load to exactly the arrays of the same mesh stored as four tight views with 16-bit
    indices — the layout the reference understands, whose arithmetic is pinned in test_oracle_pins."""
    import json
    import pin_cases
    tight_path, lc = pin_cases.loader_case(str(tmp_path / "tight"))
    g = json.load(open(tight_path))
    tight_blob = open(os.path.join(str(tmp_path / "tight"), "loader_case.bin"), "rb").read()
    n = 24
    pos = np.frombuffer(tight_blob, np.float32, n * 3, 0).reshape(n, 3)
    nrm = np.frombuffer(tight_blob, np.float32, n * 3, pos.nbytes).reshape(n, 3)
    uv = np.frombuffer(tight_blob, np.float32, n * 2, pos.nbytes + nrm.nbytes).reshape(n, 2)
    tan = np.frombuffer(tight_blob, np.float32, n * 4, pos.nbytes + nrm.nbytes + uv.nbytes).reshape(n, 4)
    inter = np.concatenate([pos, nrm, uv, tan], axis=1).astype(np.float32)          # 12 floats = 48 B per vertex
    idx8 = np.arange(n, dtype=np.uint8)
    blob = inter.tobytes() + idx8.tobytes()
    g["bufferViews"] = [{"buffer": 0, "byteOffset": 0, "byteLength": inter.nbytes, "byteStride": 48},
                        {"buffer": 0, "byteOffset": inter.nbytes, "byteLength": n}]
    g["accessors"] = [{"bufferView": 0, "byteOffset": 0, "componentType": 5126, "count": n, "type": "VEC3"},
                      {"bufferView": 0, "byteOffset": 12, "componentType": 5126, "count": n, "type": "VEC3"},
                      {"bufferView": 0, "byteOffset": 24, "componentType": 5126, "count": n, "type": "VEC2"},
                      {"bufferView": 0, "byteOffset": 32, "componentType": 5126, "count": n, "type": "VEC4"},
                      {"bufferView": 1, "componentType": 5121, "count": n, "type": "SCALAR"}]
    g["buffers"] = [{"byteLength": len(blob), "uri": "inter.bin"}]
    d = tmp_path / "inter"
    d.mkdir()
    (d / "inter.bin").write_bytes(blob)
    json.dump(g, open(d / "inter.gltf", "w"))
    a, b = rt.load_gltf(tight_path), rt.load_gltf(str(d / "inter.gltf"))
    assert a.positions.shape == b.positions.shape == (16, 9)
    for name in ("positions", "normals", "texcoords", "tangents"):
        assert np.array_equal(getattr(a, name).view(np.uint32), getattr(b, name).view(np.uint32)), name


def test_jpeg_textures_decode_close_to_libjpeg(rt, tmp_path):
    """Extension: baseline JPEG textures.  The decoder follows stb_image's pipeline (the reference's
    loader), which is not available here, so its texel bytes are PARITY UNPINNED; this test only checks it against libjpeg
    (PIL) on 4:4:4 / 4:2:2 / 4:2:0 / restart-interval / grayscale files: IDCT and chroma-filter rounding may differ by a
    few levels, nothing more."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:93, 0:130]
    img = np.stack([128 + 100 * np.sin(xx / 9.0) * np.cos(yy / 13.0), 128 + 90 * np.cos(xx / 17.0 + yy / 5.0), 40 + yy * 2], axis=2).clip(0, 255).astype(np.uint8)
    img[20:40, 30:60] = rng.integers(0, 255, (20, 30, 3))
    cases = [("444", dict(quality=95, subsampling=0), 3), ("422", dict(quality=90, subsampling=1), 8), ("420", dict(quality=85, subsampling=2), 3),
             ("420_restart", dict(quality=60, subsampling=2, restart_marker_blocks=4), 3), ("gray", dict(quality=92), 2)]
    for name, kw, tol in cases:
        path = str(tmp_path / f"{name}.jpg")
        Image.fromarray(img[:, :, 0] if name == "gray" else img).save(path, "JPEG", **kw)
        mine = rt.decode_png(path)
        ref = np.asarray(Image.open(path).convert("RGB"))
        d = np.abs(mine.astype(int) - ref.astype(int))
        assert mine.shape == ref.shape and d.max() <= tol and d.mean() < 0.3, (name, int(d.max()), float(d.mean()))
    Image.fromarray(img).save(str(tmp_path / "prog.jpg"), "JPEG", progressive=True)
    with pytest.raises(rt.RtError):
        rt.decode_png(str(tmp_path / "prog.jpg"))


def test_host_preparation_reproduces_reference_figure_and_light_orders(rt, sphere_scene):
    """The parity-critical host logic (std::sort / std::partition replay of the reference's BVH build and light list, hw8
    and hw5 flavours, including the threaded subtree build) checked without a GPU against the oracle, whose orders are pinned
    against the compiled reference (tests/test_oracle_pins.py)."""
    import oracle_lib
    import pin_cases
    for sd in (sphere_scene, pin_cases.random_triangle_scene(n=700, seed=11), pin_cases.random_triangle_scene(n=40000, seed=12, n_emissive_mats=3)):
        fo, lo = rt.host_prepare_orders(sd, rt.RT_INTEGRATOR_HW8)
        orc = oracle_lib.Hw8Oracle(sd)
        assert np.array_equal(fo, orc.figure_order()) and np.array_equal(lo, orc.light_order())
    for name, (mk, _, _, _) in pin_cases.HW6_CASES.items():
        sd6 = mk()
        fo, lo = rt.host_prepare_orders(sd6, rt.RT_INTEGRATOR_HW6)
        orc6 = oracle_lib.Hw6Oracle(sd6)
        assert np.array_equal(fo, orc6.figure_order()) and np.array_equal(lo, orc6.light_order()), name
    sd5, *_ = rt.load_txt(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scenes", "txt", "hw5_mixed_figures.txt"), rt.RT_INTEGRATOR_HW5)
    fo, lo = rt.host_prepare_orders(sd5, rt.RT_INTEGRATOR_HW5)
    ofo, olo = oracle_lib.Hw5Oracle(sd5).orders()
    assert np.array_equal(fo, ofo) and np.array_equal(lo, olo) and len(lo) == 8


def test_gltf_index_accessor_byte_offset_is_honoured(rt, tmp_path):
    """Extension beyond the reference (which ignores the index accessor's byteOffset, hw8/src/sceneio.cpp:258-269): two index
    accessors packed into ONE bufferView — mesh 0 uses the first 12 indices, mesh 1 the last 12 — must load exactly the
    triangles of the same file written with one tight index view per mesh."""
    import json
    import pin_cases
    tight_path, _ = pin_cases.loader_case(str(tmp_path / "base"))
    g = json.load(open(tight_path))
    blob = open(os.path.join(str(tmp_path / "base"), "loader_case.bin"), "rb").read()
    idx_view = g["bufferViews"][4]
    prim = g["meshes"][0]["primitives"][0]

    def variant(name, views, accessors):
        h = json.loads(json.dumps(g))
        h["bufferViews"] = g["bufferViews"][:4] + views
        h["accessors"] = g["accessors"][:4] + accessors
        h["meshes"] = [{"primitives": [dict(prim, indices=4)]}, {"primitives": [dict(prim, indices=5)]}]
        d = tmp_path / name
        d.mkdir()
        (d / "loader_case.bin").write_bytes(blob)
        json.dump(h, open(d / "scene.gltf", "w"))
        return rt.load_gltf(str(d / "scene.gltf"))

    off, half = idx_view["byteOffset"], 12
    shared = variant("shared", [dict(idx_view)],
                     [{"bufferView": 4, "byteOffset": 0, "componentType": 5123, "count": half, "type": "SCALAR"},
                      {"bufferView": 4, "byteOffset": 2 * half, "componentType": 5123, "count": half, "type": "SCALAR"}])
    split = variant("split", [{"buffer": 0, "byteOffset": off, "byteLength": 2 * half}, {"buffer": 0, "byteOffset": off + 2 * half, "byteLength": 2 * half}],
                    [{"bufferView": 4, "componentType": 5123, "count": half, "type": "SCALAR"},
                     {"bufferView": 5, "componentType": 5123, "count": half, "type": "SCALAR"}])
    assert shared.positions.shape == split.positions.shape == (8, 9)
    for name in ("positions", "normals", "texcoords", "tangents"):
        assert np.array_equal(getattr(shared, name).view(np.uint32), getattr(split, name).view(np.uint32)), name
    assert not np.array_equal(shared.positions[:4], shared.positions[4:])  # the two meshes really use different triangles
