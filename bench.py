#!/usr/bin/env python3
"""Headline benchmark: Msamples/s of the replay path tracer at 1920x1080x256 spp on synth_room_v1.

    python bench.py --gpus N --steps K --warmup W

One "step" = one complete render of the workload frame (every pixel, every sample, tonemap included) with the
scene already resident in HBM.

N > 1.  `python bench.py --gpus N` works from a plain shell: the un-launched parent (which never touches the GPU)
starts `python -m torch.distributed.run --nproc-per-node N bench.py <same args>` as a CHILD process, relays rank 0's
JSON line and exits with the child's return code; started under torch.distributed.run already (the driver's way),
it is simply one of the ranks.  One rank per GPU; the frame is cut into 32x32 tiles dealt round-robin to the ranks
(no data-path collective while rendering) and the one exchange step — a gather of the rendered u8 tiles to rank 0
over RCCL — is inside the timed region.

What N > 1 measures (reference seam: hw8/src/sceneio.cpp:387-396, the pixel loop being sharded):
  default `--scaling strong`: the NAMED workload split N ways — `--gpus N` renders the 1920x1080x256 frame of the
      metric on N GPUs, `--gpus 8 --workload synth_room_v1_3840x2160x1024` is BASELINE.json configs[4] exactly;
  `--scaling weak`: every GPU keeps one base frame's worth of pixels, the frame grows by sqrt(N) per side; the
      `metric` string then names the grown frame (it is NOT the headline metric).

Prints ONE JSON line on rank 0 (contract in the task statement): metric/value/... plus `roofline` and `cpu_baseline`.
"""
import argparse
import importlib
import json
import math
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

WORKLOADS = {
    # BASELINE.json configs[3] (the config the metric is quoted on) and configs[4]
    "synth_room_v1_1920x1080x256": dict(width=1920, height=1080, spp=256, spheres=64, segs=50, rings=43),
    "synth_room_v1_3840x2160x1024": dict(width=3840, height=2160, spp=1024, spheres=64, segs=50, rings=43),
    # small variant for quick functional checks (NOT a valid benchmark number)
    "synth_room_small_320x180x16": dict(width=320, height=180, spp=16, spheres=8, segs=12, rings=9),
}
HEADLINE = "synth_room_v1_1920x1080x256"
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured with a float4 copy)
TILE = 32


def algorithmic_bytes_per_sample(n_tris, n_lights, s_bar, p_bar, t_bar, spp):
    """root-to-leaf lower bound, independent of the BVH actually built."""
    b_node, b_pos = 32, 36
    b_hit = 108 + 32 + 12 * t_bar
    scene_q = (math.ceil(math.log2(max(n_tris, 2))) + 1) * b_node + b_pos + b_hit
    light_q = (math.ceil(math.log2(max(n_lights, 2))) + 1) * b_node + b_pos
    return s_bar * scene_q + p_bar * light_q + 12.0 / spp


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)  # the first render also allocates the path-state buffers
    ap.add_argument("--workload", default=HEADLINE, choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (profiling only: the result is not the headline metric)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="N>1: strong (default) = the named workload's frame split N ways; weak = per-GPU pixel count fixed, the frame grows (labelled as such)")
    ap.add_argument("--emulate-shards", type=int, default=0, help="diagnostic: render only shard 0 of N on this one GPU (what each GPU does at --gpus N, without the gather)")
    ap.add_argument("--sample-streams", type=int, default=0, help="diagnostic: throughput mode with K random streams per pixel (NOT the reference's pixel stream, hence not the headline metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the bounded baseline sample")
    return ap.parse_args(argv)


def launch_ranks(args):
    """The un-launched parent of an N>1 run: start the ranks as a child process tree and relay rank 0's JSON line.
    Nothing here imports torch or touches HIP, and the ranks are started with subprocess (never exec): a process that has
    initialised the GPU must not be replaced."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in child.stdout.splitlines() if l.startswith("{")]
    for l in child.stdout.splitlines():
        if not l.startswith("{"):
            print(l, file=sys.stderr)  # launcher chatter goes to stderr so that stdout stays ONE JSON line
    if lines:
        print(lines[-1], flush=True)
    return child.returncode if child.returncode != 0 or lines else 1


def cpu_identity():
    """CPU model string, logical CPUs usable by this process, physical cores behind them (/proc/cpuinfo)."""
    logical = sorted(os.sched_getaffinity(0))
    model, cores, phys, core, cpu = "unknown", set(), None, None, None
    try:
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "processor":
                cpu, phys, core = int(v), None, None
            elif k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            if cpu in logical and phys is not None and core is not None:
                cores.add((phys, core))
    except OSError:
        pass
    return model, len(logical), (len(cores) or len(logical))


class FakeScene:
    """RTAMD_BENCH_FAKE_RENDER=1 (CPU test of the launcher / gather / timing plumbing only, never a measurement): fills the
    shard buffer with a function of the global pixel position instead of rendering, so the assembled frame can be checked."""

    def __init__(self, rt, W, H, rank, world):
        import numpy as np
        self.np, self.W, self.H, self.rank, self.world = np, W, H, rank, world

    def fill(self, out8):
        np = self.np
        tiles_x, tiles_y = (self.W + TILE - 1) // TILE, (self.H + TILE - 1) // TILE
        buf = np.zeros((len(range(self.rank, tiles_x * tiles_y, self.world)), TILE, TILE, 3), np.uint8)
        for st, t in enumerate(range(self.rank, tiles_x * tiles_y, self.world)):
            x0, y0 = (t % tiles_x) * TILE, (t // tiles_x) * TILE
            w, h = min(TILE, self.W - x0), min(TILE, self.H - y0)
            yy, xx = np.mgrid[y0:y0 + h, x0:x0 + w]
            buf[st, :h, :w, 0] = xx & 255
            buf[st, :h, :w, 1] = yy & 255
            buf[st, :h, :w, 2] = (xx + yy) & 255
        import torch
        out8.copy_(torch.from_numpy(buf.reshape(-1)))

    @staticmethod
    def expected(np, W, H):
        yy, xx = np.mgrid[0:H, 0:W]
        return np.stack([xx & 255, yy & 255, (xx + yy) & 255], axis=2).astype(np.uint8)


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        sys.exit(launch_ranks(args))  # parent: has not imported torch, has not touched the GPU
    world = int(world_env or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import numpy as np
    import torch

    fake = os.environ.get("RTAMD_BENCH_FAKE_RENDER") == "1"
    if not fake and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    # Rehearsal on a one-GPU box: RTAMD_BENCH_REHEARSAL=1 puts every rank on cuda:0 and gathers over gloo, which
    # exercises all of the N>1 logic except RCCL itself (RCCL refuses two ranks on one device).
    rehearsal = fake or os.environ.get("RTAMD_BENCH_REHEARSAL") == "1"
    dev_index = 0 if rehearsal else local_rank
    if not fake:
        torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))

    rt = importlib.import_module("raytracing-course-hw_amd")
    rtd = importlib.import_module("raytracing-course-hw_amd.distributed")
    import gen_synth_room

    wl = WORKLOADS[args.workload]
    W, H, SPP = wl["width"], wl["height"], (args.spp if args.spp > 0 else wl["spp"])
    base_w, base_h = W, H
    scaling = args.scaling if world > 1 else "n/a"  # at N = 1 there is nothing to scale: the workload is the named frame
    if world > 1 and args.scaling == "weak":
        # Weak scaling: every GPU keeps one base frame's worth of pixels; the frame grows by sqrt(N) per side.
        # A pixel's samples are serial in replay mode, so pixels are the only axis that can grow.
        k = math.sqrt(world)
        W, H = int(round(W * k / 8.0)) * 8, int(round(H * k / 8.0)) * 8
    dev = "cpu" if fake else "cuda"
    emu = args.emulate_shards if (args.emulate_shards > 1 and world == 1) else 0
    params = rt.make_params(W, H, SPP, shard_index=rank, shard_count=(emu or world), tile=TILE, sample_streams=args.sample_streams)
    n_elems = rt.lib.rt_output_elems(params)
    out_rgb8 = torch.zeros(n_elems, dtype=torch.uint8, device=dev)
    gatherer = rtd.FrameGatherer(dist, W, H, SPP, rank, world, TILE, torch.uint8, "cpu" if rehearsal else "cuda") if world > 1 else None

    if fake:
        scene, sd, info, t_load = FakeScene(rt, W, H, rank, world), None, None, 0.0
        stream = None
    else:
        tmp = tempfile.mkdtemp(prefix=f"synth_room_r{rank}_")
        t0 = time.time()
        gltf, n_tris = gen_synth_room.generate(tmp, wl["spheres"], wl["segs"], wl["rings"])
        sd = rt.load_gltf(gltf)
        t_load = time.time() - t0
        scene = rt.Scene(sd)
        info = scene.info()
        stream = torch.cuda.current_stream()
        params.stream = stream.cuda_stream
        out_rgb = torch.zeros(n_elems, dtype=torch.float32, device="cuda")
    kernel_ms, dom_ms, dom_launches, closest_q, light_q = [], [], [], [], []
    frame = None

    def step():
        nonlocal frame
        if fake:
            scene.fill(out_rgb8)
            st = None
        else:
            st = scene.render_device(params, out_rgb.data_ptr(), out_rgb8.data_ptr())
            kernel_ms.append(st.kernel_ms)
            dom_ms.append(st.dominant_kernel_ms)
            dom_launches.append(st.dominant_kernel_launches)
            closest_q.append(st.closest_hit_queries)
            light_q.append(st.light_pdf_queries)
        if world > 1:  # the one exchange step: tonemapped tiles to rank 0 over RCCL/xGMI, assembled into the frame there
            frame = gatherer.gather(out_rgb8)  # stays on rank 0's GPU (buffers were allocated before the timed region)
        return st

    def sync():
        if world > 1:
            dist.barrier()
        if not fake:
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    kernel_ms.clear(); dom_ms.clear(); dom_launches.clear(); closest_q.clear(); light_q.clear()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st = step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    total_samples = W * H * SPP * args.steps
    value = total_samples / elapsed / 1e6

    headline = (args.workload == HEADLINE and (W, H) == (base_w, base_h) and args.spp <= 0 and args.sample_streams <= 1 and not fake)
    if headline:
        metric = "Msamples/sec at 1920x1080x256spp"
    else:
        why = []
        if (W, H) != (base_w, base_h):
            why.append(f"weak scaling: frame grown to {W}x{H}")
        if args.sample_streams > 1:
            why.append(f"throughput mode with {args.sample_streams} streams per pixel")
        if fake:
            why.append("FAKE RENDER: plumbing test, not a measurement")
        metric = f"Msamples/sec at {W}x{H}x{SPP}spp ({args.workload}" + "".join("; " + w for w in why) + "; NOT the headline config)"
    base = {"metric": metric, "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic"}

    if fake:
        if rank == 0:
            ok = True
            if world > 1:
                ok = bool(np.array_equal(frame.cpu().numpy(), FakeScene.expected(np, W, H)))
            base["config"] = {"workload": args.workload, "width": W, "height": H, "spp": SPP, "gathered_frame_ok": ok}
            base["roofline"] = None
            base["cpu_baseline"] = None
            print(json.dumps(base), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    if emu and rank == 0:
        print(json.dumps({"diagnostic": f"shard 0 of {emu} on one GPU", "ms_per_step": round(elapsed / args.steps * 1e3, 3),
                          "shard_samples": int(st.samples), "shard_msamples_per_s": round(st.samples * args.steps / elapsed / 1e6, 3),
                          "projected_aggregate_if_all_shards_equal": round(st.samples * emu * args.steps / elapsed / 1e6, 3)}), flush=True)
        scene.close()
        return

    if rank == 0:
        samples_per_render = st.samples  # this rank's share
        # Work counters: closest-hit / light-pdf queries per camera sample (S-bar, P-bar) come from the TIMED renders' own
        # stats when the kernel reports them without the counting variant; node visits / triangle tests (diagnostics only) from
        # an untimed counting run at reduced spp.
        cnt_spp = min(SPP, 4)
        cparams = rt.make_params(W, H, cnt_spp, shard_index=rank, shard_count=world, tile=TILE, flags=rt.RT_FLAG_COUNTERS, stream=stream.cuda_stream)
        cst = scene.render_device(cparams, out_rgb.data_ptr(), None)
        torch.cuda.synchronize()
        if sum(closest_q) > 0:
            s_bar = sum(closest_q) / (len(closest_q) * max(1, samples_per_render))
            p_bar = sum(light_q) / (len(light_q) * max(1, samples_per_render))
            counters_from = "the timed renders"
        else:
            s_bar = cst.closest_hit_queries / max(1, cst.samples)
            p_bar = cst.light_pdf_queries / max(1, cst.samples)
            counters_from = f"an untimed counting render at {cnt_spp} spp"
        t_bar = 1.0
        bps = algorithmic_bytes_per_sample(info.n_triangles, info.n_lights, s_bar, p_bar, t_bar, SPP)
        k_ms = sum(kernel_ms) / max(1, len(kernel_ms))
        render_gbs = bps * samples_per_render / (k_ms * 1e-3) / 1e9
        kernel_name = rt.dominant_kernel_name(st)
        n_l = max(1, sum(dom_launches))
        launch_ms = sum(dom_ms) / n_l                                      # average launch duration, HIP events on the launch stream
        q_bytes = (math.ceil(math.log2(max(info.n_triangles, 2))) + 1) * 32 + 36
        lq_bytes = (math.ceil(math.log2(max(info.n_lights, 2))) + 1) * 32 + 36
        queries_per_launch = s_bar * samples_per_render * len(dom_ms) / n_l
        light_queries_per_launch = p_bar * samples_per_render * len(dom_ms) / n_l
        if kernel_name == "wf_traverse_kernel":
            # per-round traversal launch: its algorithmic bytes are the root-to-leaf parts of SURVEY 8(d): per closest-hit query
            # (ceil(log2 N_tri)+1) nodes x 32 B + 36 B positions, per light-pdf query the same with N_light.
            achieved = (q_bytes * queries_per_launch + lq_bytes * light_queries_per_launch) / (launch_ms * 1e-3) / 1e9
        else:
            # one launch renders the frame: all of SURVEY 8(d)'s bytes per sample x the samples of the launch
            achieved = bps * samples_per_render * len(dom_ms) / n_l / (launch_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None, "traffic_measured_in_this_run": False,
                    "kernel": kernel_name,
                    "kernel_avg_launch_ms": round(launch_ms, 4), "kernel_launches_per_step": int(n_l / max(1, len(dom_ms))),
                    "kernel_bytes_per_query": q_bytes, "queries_per_launch": round(queries_per_launch, 1),
                    "kernel_bytes_per_light_query": lq_bytes, "light_queries_per_launch": round(light_queries_per_launch, 1),
                    "whole_render": {"achieved": round(render_gbs, 3), "frac": round(render_gbs / HBM_PEAK_GBS, 6), "gpu_ms": round(k_ms, 3),
                                     "bytes_per_sample": round(bps, 1)},
                    "s_bar": round(s_bar, 3), "p_bar": round(p_bar, 3), "s_bar_p_bar_from": counters_from,
                    "node_visits_per_sample": round(cst.node_visits / max(1, cst.samples), 2),
                    "triangle_tests_per_sample": round(cst.triangle_tests / max(1, cst.samples), 2)}
        # HBM-side traffic of the dominant kernel comes from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE need separate
        # passes and cannot be read from inside this process): tools/pmc_traffic.py condenses such a run of THIS command
        # into profiles/traffic_latest.json, which is replayed here with its provenance when workload and kernel match
        # (`traffic_measured_in_this_run` stays false: it is a committed measurement, not one taken by this run).
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("workload") == args.workload and world == 1 and tj.get("kernel") == roofline["kernel"]:
                    roofline["traffic"] = int(tj["bytes_per_frame"] / max(1, roofline["kernel_launches_per_step"])) if "bytes_per_frame" in tj else tj["bytes_per_launch"]
                    roofline["traffic_source"] = tj.get("source", "profiles/traffic_latest.json")
            except (OSError, ValueError, KeyError):
                pass
        # What the kernel is really limited by (the working set is cache-resident, "hbm" above is the nominal roofline of SURVEY 8(d)):
        # the SQ / TCC counters of the same profiling run, condensed by tools/pmc_limiter.py into profiles/limiter_latest.json.
        lpath = os.path.join(ROOT, "profiles", "limiter_latest.json")
        if os.path.exists(lpath):
            try:
                lj = json.load(open(lpath))
                if lj.get("workload") == args.workload and lj.get("kernel") == roofline["kernel"]:
                    roofline["limiter"] = lj
            except (OSError, ValueError):
                pass
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(args, rt, np, sd, W, H, SPP)
        base["config"] = {"workload": args.workload if (W, H) == (base_w, base_h) else f"{args.workload} grown to {W}x{H} ({world} x {base_w}x{base_h} pixels)",
                          "scene": "synth_room_v1 (seed 20241223)", "triangles": int(info.n_triangles),
                          "emissive_triangles": int(info.n_lights), "width": W, "height": H, "spp": SPP, "ray_depth": 6,
                          "parallelism": f"pixel tiles {TILE}x{TILE} round-robin over {world} GPU(s)" + ((", gloo gather of u8 tiles (one-GPU rehearsal)" if rehearsal else ", RCCL gather of u8 tiles") if world > 1 else ""),
                          "bvh_nodes": int(info.n_bvh_nodes), "bvh_depth": int(info.bvh_depth), "light_bvh_depth": int(info.light_bvh_depth), "scene_prep_ms": round(info.prep_ms, 1), "scene_upload_ms": round(info.upload_ms, 1),
                          "scene_load_ms": round(t_load * 1e3, 1), "device_bytes": int(info.device_bytes)}
        if world == 1 and not args.no_cpu_baseline:  # a lean run (profilers use --no-cpu-baseline) launches nothing but the timed renders and the counting render
            # Untimed side measurement: the same scene with its tree built on the GPU (RT_BUILD_DEVICE_BVH).  Not the
            # headline: the figure order becomes the load order, so these frames follow the reference's estimator, not its pixels.
            try:
                t0 = time.perf_counter()
                fast = rt.Scene(sd, build_flags=rt.RT_BUILD_DEVICE_BVH)
                create_ms = (time.perf_counter() - t0) * 1e3
                fi = fast.info()
                fp = rt.make_params(W, H, min(SPP, 64), tile=TILE, stream=stream.cuda_stream, sample_streams=8)
                fst = min((fast.render_device(fp, out_rgb.data_ptr(), out_rgb8.data_ptr()) for _ in range(2)), key=lambda t: t.kernel_ms)
                fcp = rt.make_params(W, H, min(SPP, 4), tile=TILE, flags=rt.RT_FLAG_COUNTERS, stream=stream.cuda_stream)
                fct = fast.render_device(fcp, out_rgb.data_ptr(), None)
                fast.close()
                base["config"]["device_built_tree"] = {"scene_create_ms": round(create_ms, 1), "host_prep_ms": round(fi.prep_ms, 1), "gpu_build_ms": round(fi.bvh_build_ms, 2),
                                                       "bvh_nodes": int(fi.n_bvh_nodes), "bvh_depth": int(fi.bvh_depth),
                                                       "node_visits_per_sample": round(fct.node_visits / max(1, fct.samples), 2),
                                                       "msamples_per_s_throughput_mode_k8": round(fst.samples / fst.kernel_ms / 1e3, 1),
                                                       "note": "untimed side measurement; frames of such a scene are statistically, not pixel-wise, the reference's"}
            except Exception as e:  # the side measurement must not take the bench line down
                base["config"]["device_built_tree"] = {"error": str(e)}
            # What the exactness gate costs on this box: the same frame, one untimed render each, without it.
            try:
                variants = {}
                os.environ["RTAMD_KERNEL"] = "wavefront"           # round 1's round pipeline: the conservative walk's answer stands
                try:
                    vst = scene.render_device(params, out_rgb.data_ptr(), out_rgb8.data_ptr())
                finally:
                    os.environ.pop("RTAMD_KERNEL", None)
                variants["round_pipeline_without_exactness_gate_msamples_per_s"] = round(vst.samples / vst.kernel_ms / 1e3, 1)
                os.environ["RTAMD_NO_EXACT_BOXES"] = "1"           # read at scene creation
                try:
                    plain = rt.Scene(sd)
                finally:
                    os.environ.pop("RTAMD_NO_EXACT_BOXES", None)
                vst = plain.render_device(params, out_rgb.data_ptr(), out_rgb8.data_ptr())
                plain.close()
                variants["persistent_pipeline_without_exactness_gate_msamples_per_s"] = round(vst.samples / vst.kernel_ms / 1e3, 1)
                gated = min(kernel_ms) if kernel_ms else None       # the timed renders of this run (kernel time), same box
                if gated:
                    variants["exactness_gate_cost_pct"] = round(100.0 * (gated / vst.kernel_ms - 1.0), 2)  # persistent pipeline, kernel time with the gate vs without
                variants["note"] = "untimed single renders (kernel time) on the same box; the headline runs with the gate: every pixel the reference's"
                base["config"]["without_exactness_gate"] = variants
            except Exception as e:
                base["config"]["without_exactness_gate"] = {"error": str(e)}
        base["roofline"] = roofline
        base["cpu_baseline"] = cpu
        print(json.dumps(base), flush=True)
    scene.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(args, rt, np, sd, W, H, SPP):
    """The oracle (kind "port") timed on this box's host cores on a bounded sample of the same workload, plus — where
    oracle/_ref holds it — the reference's own hw7 sources on the same scene without textures."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib  # the checker, used here only as the timed CPU baseline
    model, cores, cores_phys = cpu_identity()
    orc = oracle_lib.Hw8Oracle(sd)
    cw, ch = min(W, 256), min(H, 144)
    rect = ((W - cw) // 2, (H - ch) // 2, cw, ch)
    t1 = time.perf_counter()
    orc.render(W, H, 1, rect=rect, threads=cores)
    probe = time.perf_counter() - t1
    cspp = max(1, min(SPP, int(args.cpu_seconds / max(probe, 1e-3))))
    t1 = time.perf_counter()
    ref_rgb, _, _ = orc.render(W, H, cspp, rect=rect, threads=cores)
    dt = time.perf_counter() - t1
    # the same code on ONE thread (BASELINE.md section 3 asks for both): a 64x36 block, a few spp, ~3 s
    sw, sh_ = min(W, 64), min(H, 36)
    srect = ((W - sw) // 2, (H - sh_) // 2, sw, sh_)
    t1 = time.perf_counter()
    orc.render(W, H, 1, rect=srect, threads=1)
    probe1 = time.perf_counter() - t1
    sspp = max(1, min(SPP, int(3.0 / max(probe1, 1e-3))))
    t1 = time.perf_counter()
    orc.render(W, H, sspp, rect=srect, threads=1)
    dt1 = time.perf_counter() - t1
    cpu = {"value": round(cw * ch * cspp / dt / 1e6, 5), "unit": "Msamples/s", "cores": cores, "cores_physical": cores_phys, "cpu_model": model,
           "kind": "port",
           "sample": f"oracle (CPU restatement, OpenMP dynamic,8, {cores} threads = logical CPUs) on the centre {cw}x{ch} pixel block of the same {W}x{H} frame at {cspp} spp, {dt:.1f} s",
           "value_1_thread": round(sw * sh_ * sspp / dt1 / 1e6, 5),
           "sample_1_thread": f"centre {sw}x{sh_} block at {sspp} spp, {dt1:.1f} s"}
    # The reference ITSELF where it can be compiled: hw7 (hw8's integrator before textures) is built from the
    # reference's own sources into oracle/_ref/libref_hw7.so.  It renders the same geometry / lights / camera with
    # the textures stripped, on the same pixel block; the GPU renders that scene with RT_INTEGRATOR_HW7 and the two
    # are compared on the block.  Skipped silently when the library is not there.
    try:
        if oracle_lib.ref_path("libref_hw7.so") and SPP >= 8:
            sd7 = rt.SceneData.from_desc(sd.desc)
            for i in range(sd7.n_materials):
                m = sd7.materials[i]
                m.base_color_texture = m.emissive_texture = m.metallic_roughness_texture = m.normal_texture = -1
            sd7._build_desc()
            t1 = time.perf_counter()
            ref7 = oracle_lib.Ref7(sd7)                      # includes the reference's own BVH build
            t_build7 = time.perf_counter() - t1
            rspp = max(1, min(SPP, cspp // 2))
            t1 = time.perf_counter()
            ref7_rgb, _, _ = ref7.render(W, H, rspp, rect=rect, threads=cores)
            dt7 = time.perf_counter() - t1
            scene7 = rt.Scene(sd7)
            g7, _, st7 = scene7.render(W, H, rspp, integrator=rt.RT_INTEGRATOR_HW7, want_rgb8=False)
            crop = g7[rect[1]:rect[1] + ch, rect[0]:rect[0] + cw]
            scene7.close()
            cpu["reference_hw7"] = {"value": round(cw * ch * rspp / dt7 / 1e6, 5), "unit": "Msamples/s", "cores": cores, "cores_physical": cores_phys, "kind": "reference",
                                    "sample": f"the reference's hw7 sources (textures stripped from the scene) on the same {cw}x{ch} block at {rspp} spp, {dt7:.1f} s; its own scene build took {t_build7:.1f} s",
                                    "gpu_same_scene_msamples_per_s": round(W * H * rspp / st7.kernel_ms / 1e3, 3),
                                    "rmse_gpu_vs_reference_on_block": float(np.sqrt(np.mean((crop.astype(np.float64) - ref7_rgb) ** 2)))}
    except Exception as e:  # the reference harness is optional equipment
        cpu["reference_hw7"] = {"skipped": repr(e)[:200]}
    return cpu


if __name__ == "__main__":
    main()
