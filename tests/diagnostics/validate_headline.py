#!/usr/bin/env python3
"""One-off parity sweep at the headline size (BASELINE.json configs[3]): render the full 1920x1080x256 frame of synth_room_v1 on
the GPU and let the CPU oracle replay N randomly placed 32x32 tiles of it (seeded), reporting how many tiles / pixels agree bit for
bit, the desynchronised pixels (|difference| > 1e-3 in any channel) and the RMSE over all replayed pixels.

    python tools/validate_headline.py --tiles 200 --out gpurun_out/headline_parity.json

Test infrastructure (it calls the oracle); the pytest version of this check (tests/test_gpu_scenes.py) replays 100 crops, 4.9 % of the frame."""
import argparse
import importlib
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=200)
    ap.add_argument("--seed", type=int, default=20241223)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    rt = importlib.import_module("raytracing-course-hw_amd")
    import gen_synth_room
    import oracle_lib
    W, H, SPP = 1920, 1080, 256
    path, ntris = gen_synth_room.generate(tempfile.mkdtemp(prefix="synth_room_"), 64, 50, 43)
    sd = rt.load_gltf(path)
    scene = rt.Scene(sd)
    rgb, rgb8, st = scene.render(W, H, SPP)
    scene.close()
    orc = oracle_lib.Hw8Oracle(sd)
    rng = np.random.default_rng(args.seed)
    all_tiles = [(tx * 32, ty * 32) for ty in range((H + 31) // 32) for tx in range(W // 32)]
    pick = rng.choice(len(all_tiles), size=min(args.tiles, len(all_tiles)), replace=False)
    exact_tiles = exact_px = desync_px = byte_mismatch = n_px = 0
    se = 0.0
    worst = (0.0, None)
    t0 = time.time()
    for n, i in enumerate(sorted(pick)):
        x0, y0 = all_tiles[i]
        h = min(32, H - y0)
        ref, ref8, _ = orc.render(W, H, SPP, rect=(x0, y0, 32, h))
        crop, crop8 = rgb[y0:y0 + h, x0:x0 + 32], rgb8[y0:y0 + h, x0:x0 + 32]
        d = np.abs(crop.astype(np.float64) - ref)
        same = (crop == ref).all(axis=2)
        exact_tiles += int(same.all()); exact_px += int(same.sum()); n_px += same.size
        desync_px += int((d.max(axis=2) > 1e-3).sum()); byte_mismatch += int((crop8 != ref8).sum())
        se += float((d ** 2).sum())
        r = float(np.sqrt((d ** 2).mean()))
        if r > worst[0]:
            worst = (r, (x0, y0))
        if n % 20 == 19:
            print(f"{n + 1} tiles, {time.time() - t0:.0f} s, bit-exact pixels {exact_px}/{n_px}", flush=True)
    res = {"workload": "synth_room_v1_1920x1080x256", "gpu_kernel_ms": round(st.kernel_ms, 1), "tiles_replayed": len(pick), "pixels_replayed": n_px,
           "share_of_frame": round(n_px / (W * H), 4), "bit_exact_tiles": exact_tiles, "bit_exact_pixels": exact_px,
           "desynchronised_pixels_abs_gt_1e-3": desync_px, "byte_mismatches_rgb8": byte_mismatch,
           "rmse_over_replayed_pixels": float(np.sqrt(se / (3 * n_px))), "worst_tile_rmse": worst[0], "worst_tile_at": worst[1],
           "tolerance_rmse": 1e-3, "oracle_seconds": round(time.time() - t0, 1), "tile_seed": args.seed}
    print(json.dumps(res))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
