#!/usr/bin/env python3
"""One rank's share of a strong-scaling frame with M frames in flight (M contexts on their own streams, one host thread
each): what a pipelined multi-GPU job could reach per GPU.  python3 scripts/share_inflight_probe.py [nranks] [max_m]"""
import os, sys, time, threading, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import scenes
from raytracer_amd.tiles import max_owned_tiles
W, H, K = 1920, 1080, 96
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
maxm = int(sys.argv[2]) if len(sys.argv) > 2 else 4
sc = scenes.atrium(262144)
chunk = max_owned_tiles(W, H, 64, n) * 64 * 64 * 4
def make():
    g = rt.RtHip(sc, semantics="array", accel=rt.ACCEL_PLOC, bounces=2, do_shadows=1, shadow_t_max=1.0)
    d_n = torch.zeros(W * H * 4, dtype=torch.float32, device="cuda")
    d_p = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda")
    g.render_gbuffer_device(sc.fov, W, H, d_n, d_p); g.sync()
    out = torch.zeros(chunk, dtype=torch.uint8, device="cuda")
    return g, d_p, d_n, out
packed = rt.RtHip.render_tiles_packed_device.__wrapped__
def loop(ctx, k, rank):
    g, d_p, d_n, out = ctx
    for _ in range(k):
        packed(g, [d_p], [d_n], W, H, 64, rank, n, [out])
ctxs = [make() for _ in range(maxm)]
res = {"nranks": n}
for m in range(1, maxm + 1):
    for c in ctxs[:m]:
        loop(c, 3, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=loop, args=(c, K // m, 0)) for c in ctxs[:m]]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res["in_flight_%d_ms_per_share" % m] = round(dt / (K // m * m) * 1e3, 4)
print(json.dumps(res))
