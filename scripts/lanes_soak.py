#!/usr/bin/env python3
"""Soak of the frame pipeline's lanes at full size: N frames of the bench workload on 3 lanes, three different G-buffers in
rotation, every finished image compared with the sequentially rendered one.  python3 scripts/lanes_soak.py [frames] [lanes]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import raytracer_amd as rt
from raytracer_amd import scenes
W, H = 1920, 1080
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
L = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sc = scenes.atrium(262144)
g = rt.RtHip(sc, semantics="array", accel=rt.ACCEL_PLOC, bounces=2, do_shadows=1, shadow_t_max=1.0)
gb, exp = [], []
for f in range(3):
    d_n = torch.zeros(W * H * 4, dtype=torch.float32, device="cuda"); d_p = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda")
    g.render_gbuffer_device(sc.fov - 0.05 * f, W, H, d_n, d_p)
    out = torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda")
    g.render_frame_device(d_p, d_n, W, H, out)
    gb.append((d_p, d_n)); exp.append((out, g.last_frame_rays()))
g.set_tuning(frame_streams=L)
lim = max(2, L)
outs = [torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda") for _ in range(lim)]
enq = rt.RtHip.frame_enqueue_device.__wrapped__
torch.cuda.synchronize()
bad, pend, done = 0, [], 0
t0 = time.perf_counter()
for k in range(N):
    if len(pend) == lim:
        j = pend.pop(0)
        rays = g.frame_finish()
        if rays != exp[j % 3][1] or not torch.equal(outs[j % lim], exp[j % 3][0]):
            bad += 1
        done += 1
    enq(g, gb[k % 3][0], gb[k % 3][1], W, H, outs[k % lim]); pend.append(k)
while pend:
    j = pend.pop(0)
    rays = g.frame_finish()
    if rays != exp[j % 3][1] or not torch.equal(outs[j % lim], exp[j % 3][0]):
        bad += 1
    done += 1
g.sync()
print("%d frames on %d lanes, %.3f ms per frame (with the comparisons), %d differ" % (done, L, (time.perf_counter() - t0) / done * 1e3, bad))
sys.exit(1 if bad else 0)
