#!/usr/bin/env python3
"""Frames in flight: one context vs two contexts (own streams) rendering the bench frame from two host threads."""
import os, sys, time, threading, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import scenes
W, H, K = 1920, 1080, 40
sc = scenes.atrium(262144)
def make():
    g = rt.RtHip(sc, semantics="array", accel=rt.ACCEL_PLOC, bounces=2, do_shadows=1, shadow_t_max=1.0)
    d_n = torch.zeros(W * H * 4, dtype=torch.float32, device="cuda")
    d_p = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda")
    g.render_gbuffer_device(sc.fov, W, H, d_n, d_p); g.sync()
    out = torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda")
    return g, d_p, d_n, out
render = rt.RtHip.render_tiles_batch_device.__wrapped__
def loop(ctx, n):
    g, d_p, d_n, out = ctx
    for _ in range(n):
        render(g, [d_p], [d_n], W, H, 64, 0, 1, [out])
ctxs = [make() for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2)]
res = {}
for nthreads in range(1, len(ctxs) + 1):
    for c in ctxs[:nthreads]:
        loop(c, 3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=loop, args=(c, K // nthreads)) for c in ctxs[:nthreads]]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res["threads_%d_ms_per_frame" % nthreads] = round(dt / (K // nthreads * nthreads) * 1e3, 4)
print(json.dumps(res))
