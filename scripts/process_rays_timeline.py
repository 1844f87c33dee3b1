#!/usr/bin/env python3
"""rt_hip_process_gbuffer_device (RayTracer::processRays' entry) on every covered texel of the 1080p G-buffer, a few calls: run under
rocprofv3 --kernel-trace to see where the call's time goes.  python3 scripts/process_rays_timeline.py [calls]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import raytracer_amd as rt
from raytracer_amd import scenes
W, H = 1920, 1080
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sc = scenes.atrium(262144)
g = rt.RtHip(sc, accel=rt.ACCEL_PLOC)
d_n = torch.zeros(W * H * 4, dtype=torch.float32, device="cuda"); d_p = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda")
g.render_gbuffer_device(sc.fov, W, H, d_n, d_p); g.sync()
nr = d_n.view(-1, 4); cov = nr[:, 3] > 0
pos = d_p.view(-1, 3)[cov].contiguous()
nrm = (nr[cov, :3].double() * 2.0 - 1.0).float().contiguous()
n = pos.shape[0]
d_res = torch.zeros(n * 16, dtype=torch.float32, device="cuda")
d_off = torch.zeros(2 * 4096, dtype=torch.int32, device="cuda")
call = rt.RtHip.process_gbuffer_device.__wrapped__
for _ in range(3):
    call(g, pos, nrm, n, d_res, d_off)
g.sync()
t0 = time.perf_counter()
for _ in range(calls):
    call(g, pos, nrm, n, d_res, d_off)
g.sync()
print("%d rays, %.4f ms per call" % (n, (time.perf_counter() - t0) / calls * 1e3))
