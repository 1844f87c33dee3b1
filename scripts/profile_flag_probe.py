#!/usr/bin/env python3
"""Frame time of the bench workload with config.profile on and off (does timing the trace launches through their
dispatch packets cost anything?): python3 scripts/profile_flag_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import raytracer_amd as rt
from raytracer_amd import scenes
W, H = 1920, 1080
sc = scenes.atrium(262144)
g = rt.RtHip(sc, semantics="array", accel=rt.ACCEL_PLOC, bounces=2, do_shadows=1, shadow_t_max=1.0, profile=1)
d_n = torch.zeros(W * H * 4, dtype=torch.float32, device="cuda"); d_p = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda")
g.render_gbuffer_device(sc.fov, W, H, d_n, d_p)
out = torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda")
enq = type(g).frame_enqueue_device.__wrapped__ if hasattr(type(g).frame_enqueue_device, "__wrapped__") else type(g).frame_enqueue_device
def run(n):
    enq(g, d_p, d_n, W, H, out)
    for _ in range(n - 1):
        enq(g, d_p, d_n, W, H, out)
        g.frame_finish()
    g.frame_finish(); g.sync()
for prof in (1, 0, 1, 0):
    g.set_config(profile=prof)
    run(50)
    t0 = time.perf_counter(); run(400); dt = time.perf_counter() - t0
    print("profile=%d  %.4f ms/frame" % (prof, dt / 400 * 1e3), flush=True)
g.close()
