#!/usr/bin/env python3
"""One rank's share of a strong-scaling frame when b frames are shaded per pass of launches (the blocking packed entry,
one batch at a time): ms per frame, slowest of the N ranks.  python3 scripts/share_batch_probe.py"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import scenes
from raytracer_amd.tiles import max_owned_tiles
W, H = 1920, 1080
sc = scenes.atrium(262144)
g = rt.RtHip(sc, semantics="array", accel=rt.ACCEL_PLOC, bounces=2, do_shadows=1, shadow_t_max=1.0)
d_n = torch.zeros(W * H * 4, dtype=torch.float32, device="cuda")
d_p = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda")
g.render_gbuffer_device(sc.fov, W, H, d_n, d_p); g.sync()
packed = rt.RtHip.render_tiles_packed_device.__wrapped__
res = {}
for n in (2, 4, 8):
    chunk = max_owned_tiles(W, H, 64, n) * 64 * 64 * 4
    for b in (1, 2, 4, 8):
        if (chunk // 4) * b * 2 > (1 << 23):
            continue
        outs = [torch.zeros(chunk, dtype=torch.uint8, device="cuda") for _ in range(b)]
        worst = 0.0
        for rank in range(n):
            for _ in range(3):
                packed(g, [d_p] * b, [d_n] * b, W, H, 64, rank, n, outs)
            g.sync()
            reps = max(4, 48 // b)
            t0 = time.perf_counter()
            for _ in range(reps):
                packed(g, [d_p] * b, [d_n] * b, W, H, 64, rank, n, outs)
            g.sync()
            worst = max(worst, (time.perf_counter() - t0) / (reps * b) * 1e3)
        res["N%d_batch%d" % (n, b)] = round(worst, 4)
        print("N=%d batch=%d: %.4f ms per frame (slowest rank), implied efficiency vs 0.844: %.2f" % (n, b, worst, 0.844 / (n * worst)), flush=True)
print(json.dumps(res))
