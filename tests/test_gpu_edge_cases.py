"""Edge cases of the hw8 path on the GPU, each against the oracle: empty / single-triangle / degenerate geometry, no lights,
extreme depths and frame sizes, more shards than tiles."""
import numpy as np
import pytest

import oracle_lib
import pin_cases

pytestmark = pytest.mark.gpu


def _mat(rt, emission=(0, 0, 0), color=(0.8, 0.8, 0.8), metallic=0.0, rough=0.6):
    m = rt.rt_material()
    m.base_color, m.emission, m.metallic_factor, m.roughness_factor = color, emission, metallic, rough
    m.base_color_texture = m.emissive_texture = m.metallic_roughness_texture = m.normal_texture = -1
    return m


def _scene(rt, tris, mats, mat_idx, bg=(0.05, 0.1, 0.2)):
    tris = np.asarray(tris, np.float32).reshape(-1, 3, 3)
    n = len(tris)
    e1, e2 = tris[:, 0] - tris[:, 2], tris[:, 1] - tris[:, 2]
    nrm = np.cross(e1, e2)
    ln = np.linalg.norm(nrm, axis=1, keepdims=True)
    nrm = np.where(ln > 0, nrm / np.maximum(ln, 1e-30), np.array([0, 0, 1.0]))
    nrm3 = np.repeat(nrm[:, None, :], 3, axis=1).astype(np.float32)
    tan = np.tile(np.array([1, 0, 0, 1], np.float32), (n, 3, 1))
    uv = np.zeros((n, 3, 2), np.float32)
    cam = rt.rt_camera()
    cam.position, cam.right, cam.up, cam.forward, cam.fov_y = (0, 0, 5), (1, 0, 0), (0, 1, 0), (0, 0, -1), 0.9
    return rt.SceneData(tris.reshape(n, 9), uv.reshape(n, 6), nrm3.reshape(n, 9), tan.reshape(n, 12), np.asarray(mat_idx, np.uint32), mats, camera=cam, bg=bg)


def _check(rt, sd, w, h, spp, depth=0, tag=""):
    for kernel in ("persistent", "wavefront", "mega"):
        import os
        os.environ["RTAMD_KERNEL"] = kernel
        scene = rt.Scene(sd)
        rgb, rgb8, _ = scene.render(w, h, spp, ray_depth=depth)
        scene.close()
        ref, ref8, _ = oracle_lib.Hw8Oracle(sd).render(w, h, spp, ray_depth=depth)
        ok = np.array_equal(rgb, ref, equal_nan=True)
        rmse = float(np.sqrt(np.nanmean((rgb.astype(np.float64) - ref) ** 2)))
        print(f"{tag}[{kernel}] {w}x{h}x{spp} depth {depth or 6}: bit_exact {ok} rmse {rmse:.2e} nan_px {int(np.isnan(ref).any(axis=2).sum())}")
        assert np.array_equal(np.isnan(rgb), np.isnan(ref))
        if kernel == "persistent": assert ok and np.array_equal(rgb8, ref8)   # the gated default: bit for bit
        else: assert rmse < 1e-3 and (rgb8 != ref8).sum() <= 3                # wavefront / mega: the padded boxes' answer, north_star tolerance
    os.environ.pop("RTAMD_KERNEL", None)


def test_single_triangle_and_one_light(rt):
    tris = [[[-1, -1, 0], [1, -1, 0], [0, 1, 0]], [[-3, 3, -1], [3, 3, -1], [0, 3, 2]]]
    sd = _scene(rt, tris, [_mat(rt), _mat(rt, emission=(4, 4, 3))], [0, 1])
    _check(rt, sd, 40, 24, 8, tag="two triangles")
    sd1 = _scene(rt, tris[:1], [_mat(rt)], [0])
    _check(rt, sd1, 17, 9, 3, tag="single triangle, no light")


def test_degenerate_triangles(rt):
    """Zero-area triangles (two equal vertices, a point, three collinear vertices): NaN / inf arithmetic must follow the
    reference (a NaN barycentric PASSES `u < 0 || v < 0 || u + v > 1`)."""
    sd0 = pin_cases.random_triangle_scene(n=120, seed=8)
    pos = sd0.positions.copy().reshape(-1, 3, 3)
    pos[5, 1] = pos[5, 0]            # two equal vertices
    pos[9] = pos[9, 0]               # a point
    pos[13, 2] = (pos[13, 0] + pos[13, 1]) / 2  # collinear
    mi = sd0.material_index.copy()
    mi[9] = 3                        # the point is NOT emissive here (see the next test)
    sd = rt.SceneData(pos.reshape(-1, 9), sd0.texcoords, sd0.normals, sd0.tangents, mi, list(sd0.materials)[:sd0.n_materials], camera=sd0.camera)
    _check(rt, sd, 48, 36, 6, tag="degenerate")


def test_emissive_point_light_box_rounding(rt):
    """An emissive zero-area triangle makes light sampling aim rays EXACTLY at a vertex, i.e. through the corner of
    light-BVH boxes.  There the reference's own slab test (6 divisions on a re-centred box, hw8/src/primitives.cpp:29-53)
    rejects a box by one ulp although the ray hits a triangle inside it, while a conservative padded test keeps it.
    The persistent pipeline (the default) sends such hits — and only those — through walks with the reference's own box
    arithmetic and must reproduce the oracle bit for bit; so does the round pipeline under RTAMD_ROUNDS_EXACT=1 (off by default
    there: its exact re-walks sit on every round's critical path).  The megakernel and the plain round pipeline keep the padded
    test's answer (a handful of samples differ, the random stream stays in sync)."""
    import os
    sd0 = pin_cases.random_triangle_scene(n=120, seed=8)
    pos = sd0.positions.copy().reshape(-1, 3, 3)
    pos[9] = pos[9, 0]
    mi = sd0.material_index.copy()
    mi[9] = 0
    sd = rt.SceneData(pos.reshape(-1, 9), sd0.texcoords, sd0.normals, sd0.tangents, mi, list(sd0.materials)[:sd0.n_materials], camera=sd0.camera)
    ref, _, _ = oracle_lib.Hw8Oracle(sd).render(48, 36, 6)
    imgs = {}
    for kernel in ("wavefront", "mega"):
        os.environ["RTAMD_KERNEL"] = kernel
        scene = rt.Scene(sd)
        imgs[kernel], _, _ = scene.render(48, 36, 6)
        scene.close()
    assert np.array_equal(imgs["wavefront"], imgs["mega"], equal_nan=True)
    bad = int((np.abs(imgs["mega"].astype(np.float64) - ref).max(axis=2) > 1e-3).sum())
    print(f"emissive point light: {bad} of {48 * 36} pixels differ from the oracle with the padded box test")
    assert 0 < bad <= 10
    os.environ["RTAMD_ROUNDS_EXACT"] = "1"
    for kernel, pipeline in (("persistent", rt.RT_PIPELINE_PERSISTENT), ("wavefront", rt.RT_PIPELINE_ROUNDS)):
        os.environ["RTAMD_KERNEL"] = kernel
        scene = rt.Scene(sd)
        rgb, _, st = scene.render(48, 36, 6, counters=True)
        scene.close()
        print(f"{kernel}: exact closest hits {st.exact_closest_hits}, exact light sums {st.exact_light_sums} of {st.closest_hit_queries} + {st.light_pdf_queries} queries")
        assert st.pipeline == pipeline and st.exact_light_sums > 0
        assert np.array_equal(rgb, ref, equal_nan=True)
    os.environ.pop("RTAMD_KERNEL", None)
    os.environ.pop("RTAMD_ROUNDS_EXACT", None)


@pytest.mark.parametrize("w,h,spp,depth", [(1, 1, 5, 0), (9, 7, 2, 1), (8, 8, 3, 2), (33, 5, 2, 16), (24, 16, 1, 6)])
def test_sizes_and_depths(rt, sphere_scene, w, h, spp, depth):
    _check(rt, sphere_scene, w, h, spp, depth, tag="sphere")


def test_more_shards_than_tiles(rt, sphere_scene):
    scene = rt.Scene(sphere_scene)
    full, full8, _ = scene.render(40, 24, 3)
    acc = np.zeros_like(full)
    for r in range(7):  # 3x2 tiles of 16x16 -> shard 6 owns nothing
        p = rt.make_params(40, 24, 3, shard_index=r, shard_count=7, tile=16)
        n = rt.lib.rt_output_elems(p)
        if n == 0:
            continue
        buf, _, _ = scene.render(40, 24, 3, shard_index=r, shard_count=7, tile=16)
        acc += rt.unshard(p, buf)
    assert np.array_equal(acc, full)
    scene.close()


def test_bad_parameters_are_rejected(rt, sphere_scene):
    scene = rt.Scene(sphere_scene)
    for kw in (dict(width=0, height=4, samples=1), dict(width=4, height=4, samples=0), dict(width=4, height=4, samples=1, ray_depth=17),
               dict(width=4, height=4, samples=1, shard_index=3, shard_count=2), dict(width=4, height=4, samples=1, shard_count=2, tile=12),
               dict(width=65536, height=32768, samples=1)):  # y*W+x would leave the 31-bit seed range of minstd_rand
        with pytest.raises(rt.RtError):
            scene.render(kw.pop("width"), kw.pop("height"), kw.pop("samples"), **kw)
    scene.close()


def test_spill_variant_of_the_traversal_kernel(rt, sphere_scene, monkeypatch):
    """Trees deeper than the 32-entry LDS stacks run a bounds-checked kernel variant whose deep stack entries live in a
    global overflow area.  RTAMD_WF_LDS_STACK=3 pretends the LDS stacks hold three entries, so ordinary scenes exercise the
    overflow path on nearly every ray; pixels must not change."""
    import pin_cases
    for sd, (w, h, spp) in ((sphere_scene, (64, 48, 5)), (pin_cases.random_triangle_scene(n=700, seed=21), (72, 48, 5))):
        scene = rt.Scene(sd)
        assert scene.info().bvh_depth > 3
        monkeypatch.setenv("RTAMD_KERNEL", "wavefront")  # the round pipeline owns the spill variant (and serves deeper trees)
        ref, ref8, st0 = scene.render(w, h, spp)
        monkeypatch.setenv("RTAMD_WF_LDS_STACK", "3")
        rgb, rgb8, st = scene.render(w, h, spp)
        monkeypatch.delenv("RTAMD_WF_LDS_STACK")
        assert st.launches == st0.launches > 1 and np.array_equal(rgb, ref) and np.array_equal(rgb8, ref8)
        orc, _, _ = oracle_lib.Hw8Oracle(sd).render(w, h, spp)
        assert np.array_equal(rgb, orc)
        scene.close()


def test_empty_scenes_render_the_background(rt, tmp_path):
    """No figures at all: every integrator returns the background colour for every sample (hw8/src/scene.cpp:90-92,
    hw3/src/scene.cpp:36-38, hw1/src/scene.cpp:8)."""
    cam = rt.rt_camera()
    cam.position, cam.right, cam.up, cam.forward = (0, 0, 2), (1, 0, 0), (0, 1, 0), (0, 0, -1)
    cam.fov_y, cam.fov_x = 0.9, 1.1
    z = np.zeros((0, 9), np.float32)
    sd = rt.SceneData(z, np.zeros((0, 6), np.float32), z, np.zeros((0, 12), np.float32), np.zeros(0, np.uint32), [], camera=cam, bg=(0.25, 0.5, 0.75))
    scene = rt.Scene(sd)
    rgb, rgb8, _ = scene.render(40, 24, 3)
    assert np.allclose(rgb, np.array([0.25, 0.5, 0.75], np.float32), rtol=0, atol=1e-6) and rgb8.std(axis=(0, 1)).max() == 0
    scene.close()
    txt = tmp_path / "empty.txt"
    txt.write_text("DIMENSIONS 24 16\nBG_COLOR 0.2 0.4 0.6\nCAMERA_POSITION 0 0 0\nCAMERA_RIGHT 1 0 0\nCAMERA_UP 0 1 0\nCAMERA_FORWARD 0 0 -1\nCAMERA_FOV_X 1.0\nRAY_DEPTH 4\nSAMPLES 2\n")
    for flavor in (rt.RT_INTEGRATOR_HW1, rt.RT_INTEGRATOR_HW2, rt.RT_INTEGRATOR_HW3, rt.RT_INTEGRATOR_HW4, rt.RT_INTEGRATOR_HW5):
        sdt, w, h, spp, depth = rt.load_txt(str(txt), flavor)
        scene = rt.Scene(sdt)
        rgb, _, _ = scene.render(w, h, max(1, spp), integrator=flavor, ray_depth=depth, want_rgb8=False)
        assert np.allclose(rgb, np.array([0.2, 0.4, 0.6], np.float32), rtol=0, atol=1e-6), flavor
        scene.close()


@pytest.mark.parametrize("pipes", [2, 3, 4])
def test_independent_pipelines_do_not_change_pixels(rt, sphere_scene, monkeypatch, pipes):
    """Large frames are rendered as several independent pipelines (disjoint path slots, one stream each) whose kernels overlap;
    RTAMD_WF_PIPELINES forces the cut on small frames.  Plain, sharded, spill-variant and throughput-mode renders must not change."""
    import pin_cases
    sd = pin_cases.random_triangle_scene(n=400, seed=33)
    scene = rt.Scene(sd)
    w, h, spp = 88, 56, 6
    monkeypatch.setenv("RTAMD_KERNEL", "wavefront")
    monkeypatch.setenv("RTAMD_WF_PIPELINES", "1")
    ref, ref8, st1 = scene.render(w, h, spp, counters=True)
    shard_ref, _, _ = scene.render(w, h, spp, shard_index=1, shard_count=3, tile=16)
    thr_ref, _, _ = scene.render(w, h, spp, sample_streams=3)
    monkeypatch.setenv("RTAMD_WF_PIPELINES", str(pipes))
    rgb, rgb8, st = scene.render(w, h, spp, counters=True)
    assert st.launches == pipes * st1.launches
    assert np.array_equal(rgb, ref, equal_nan=True) and np.array_equal(rgb8, ref8)
    assert (st.closest_hit_queries, st.light_pdf_queries, st.samples) == (st1.closest_hit_queries, st1.light_pdf_queries, st1.samples)
    shard, _, _ = scene.render(w, h, spp, shard_index=1, shard_count=3, tile=16)
    assert np.array_equal(shard, shard_ref, equal_nan=True)
    thr, _, _ = scene.render(w, h, spp, sample_streams=3)
    assert np.array_equal(thr, thr_ref, equal_nan=True)
    monkeypatch.setenv("RTAMD_WF_LDS_STACK", "3")
    spill, _, _ = scene.render(w, h, spp)
    assert np.array_equal(spill, ref, equal_nan=True)
    orc, _, _ = oracle_lib.Hw8Oracle(sd).render(w, h, spp)
    assert np.array_equal(rgb, orc, equal_nan=True)
    scene.close()


def test_more_rounds_than_the_per_launch_event_pool(rt, sphere_scene, monkeypatch):
    """12,000 spp at depth 6 = 72,000 wavefront rounds: beyond 65,536 rounds the driver stops bracketing every traverse launch with
    HIP events (rt_stats.dominant_kernel_ms then falls back to the whole kernel time); pixels are unaffected."""
    monkeypatch.setenv("RTAMD_KERNEL", "wavefront")
    scene = rt.Scene(sphere_scene)
    rgb, rgb8, st = scene.render(8, 8, 12000)
    scene.close()
    monkeypatch.delenv("RTAMD_KERNEL")
    scene = rt.Scene(sphere_scene)
    prgb, prgb8, pst = scene.render(8, 8, 12000)  # the persistent pipeline: one launch, the same 12,000 serial samples per pixel
    scene.close()
    assert pst.launches == 1 and pst.pipeline == rt.RT_PIPELINE_PERSISTENT and np.array_equal(prgb, rgb, equal_nan=True) and np.array_equal(prgb8, rgb8)
    ref, ref8, _ = oracle_lib.Hw8Oracle(sphere_scene).render(8, 8, 12000)
    assert st.launches == 1 + 2 * 72000 and st.dominant_kernel_launches == st.launches and st.dominant_kernel_ms == st.kernel_ms
    assert np.array_equal(rgb, ref, equal_nan=True) and np.array_equal(rgb8, ref8)


def test_persistent_pipeline_phases_and_deals_do_not_change_pixels(rt, monkeypatch):
    """The persistent pipeline renders a frame in two launches when it can re-deal the 8x8 sub-tiles between its workgroups
    (first 1/16 of the samples with the round-robin deal, then longest-processing-time-first by the measured cost, every pixel
    resuming from its record).  Forced here on a small frame (few workgroups, first phase of 2 samples): pixels, shards and
    throughput mode must be what one launch gives, and that is the oracle's frame."""
    import pin_cases
    sd = pin_cases.random_triangle_scene(n=400, seed=33)
    w, h, spp = 136, 88, 7
    ref, ref8, _ = oracle_lib.Hw8Oracle(sd).render(w, h, spp)
    scene = rt.Scene(sd)
    monkeypatch.setenv("RTAMD_PT_NO_REBALANCE", "1")
    one, one8, st1 = scene.render(w, h, spp)
    shard1, _, _ = scene.render(w, h, spp, shard_index=1, shard_count=2, tile=16)
    thr1, _, _ = scene.render(w, h, 12, sample_streams=3)
    monkeypatch.delenv("RTAMD_PT_NO_REBALANCE")
    assert st1.pipeline == rt.RT_PIPELINE_PERSISTENT and st1.launches == 1
    assert np.array_equal(one, ref, equal_nan=True) and np.array_equal(one8, ref8)
    for blocks, phase0 in ((8, 2), (3, 1), (16, 6)):
        monkeypatch.setenv("RTAMD_PT_BLOCKS", str(blocks))
        monkeypatch.setenv("RTAMD_PT_PHASE0", str(phase0))
        two, two8, st2 = scene.render(w, h, spp)
        assert st2.launches == 2 and st2.dominant_kernel_launches == 2
        assert np.array_equal(two, one, equal_nan=True) and np.array_equal(two8, one8)
        assert (st2.closest_hit_queries, st2.light_pdf_queries) == (st1.closest_hit_queries, st1.light_pdf_queries)
        shard2, _, _ = scene.render(w, h, spp, shard_index=1, shard_count=2, tile=16)
        assert np.array_equal(shard2, shard1, equal_nan=True)
        thr2, _, stt = scene.render(w, h, 12, sample_streams=3)
        assert np.array_equal(thr2, thr1, equal_nan=True)
    scene.close()


def test_persistent_pipeline_in_several_passes(rt, monkeypatch):
    """A frame with more path slots than the workgroups' LDS bitmaps hold (1,280 workgroups x 5,120 paths) is rendered in several
    passes over disjoint slot ranges; with RTAMD_PT_BLOCKS=14 a 400x300 frame (2,080 sub-tiles of 64 slots) already needs two
    (and each pass its own two phases).  Replay mode, shards and throughput mode (whose stream index comes from the global slot) must not change."""
    import pin_cases
    sd = pin_cases.random_triangle_scene(n=300, seed=12)
    w, h, spp = 400, 300, 4
    scene = rt.Scene(sd)
    one, one8, st1 = scene.render(w, h, spp)
    thr1, _, _ = scene.render(w, h, 8, sample_streams=2)
    shard1, _, _ = scene.render(w, h, spp, shard_index=2, shard_count=3, tile=32)
    monkeypatch.setenv("RTAMD_PT_BLOCKS", "14")
    monkeypatch.setenv("RTAMD_PT_PHASE0", "1")
    two, two8, st2 = scene.render(w, h, spp)
    thr2, _, stt = scene.render(w, h, 8, sample_streams=2)
    shard2, _, _ = scene.render(w, h, spp, shard_index=2, shard_count=3, tile=32)
    scene.close()
    assert st1.launches == 1 and st2.launches == 4 and stt.launches >= 4 + 1   # 2 passes x 2 phases (+ the stream reduction)
    assert np.array_equal(two, one, equal_nan=True) and np.array_equal(two8, one8)
    assert np.array_equal(thr2, thr1, equal_nan=True) and np.array_equal(shard2, shard1, equal_nan=True)
    assert (st2.closest_hit_queries, st2.light_pdf_queries) == (st1.closest_hit_queries, st1.light_pdf_queries)
    crop_ref, _, _ = oracle_lib.Hw8Oracle(sd).render(w, h, spp, rect=(180, 130, 40, 40))
    assert np.array_equal(two[130:170, 180:220], crop_ref, equal_nan=True)
