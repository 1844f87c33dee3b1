#!/usr/bin/env python3
"""Soak of the multi-GPU pipeline's entries on a one-rank communicator (the route of N ranks: packed tiles, all-gather on the
communication stream, one unpack per frame): N entries of 1..2 full-size frames (two fit the sync-free driver's bounds when one
rank owns the whole frame) on 3 lanes, three G-buffers in rotation, every finished image compared with the sequentially
rendered one.  python3 scripts/entries_soak.py [entries] [lanes]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import raytracer_amd as rt
from raytracer_amd import scenes
W, H = 1920, 1080
N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
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
g.comm_init(0, 1)
g.set_tuning(frame_streams=L)
lim = max(2, L)
outs = [[torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda") for _ in range(2)] for _ in range(lim)]
enq = rt.RtHip.frames_multi_enqueue_device.__wrapped__
torch.cuda.synchronize()
bad, pend, frames = 0, [], 0


def check(entry):
    global bad, frames
    k, ids = entry
    rays = g.frame_finish()
    if rays != sum(exp[i][1] for i in ids):
        bad += 1
    for slot, i in enumerate(ids):
        if not torch.equal(outs[k % lim][slot], exp[i][0]):
            bad += 1
        frames += 1


t0 = time.perf_counter()
for k in range(N):
    if len(pend) == lim:
        check(pend.pop(0))
    ids = [k % 3] if k % 4 == 3 else [k % 3, (k + 1 + k // 5) % 3]
    enq(g, [gb[i][0] for i in ids], [gb[i][1] for i in ids], W, H, 64, outs[k % lim][:len(ids)])
    pend.append((k, ids))
while pend:
    check(pend.pop(0))
g.sync()
print("%d entries (%d frames) on %d lanes, %.3f ms per frame (with the comparisons), %d differ" % (N, frames, L, (time.perf_counter() - t0) / frames * 1e3, bad))
g.comm_destroy()
sys.exit(1 if bad else 0)
