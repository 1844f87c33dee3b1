#!/usr/bin/env python3
"""RayTracer::processRays with device-resident inputs and outputs (rt_hip_process_gbuffer_device) on the 1080p
atrium-262k G-buffer: time per call.  Run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import scenes
W, H = 1920, 1080
sc = scenes.atrium(262144)
g = rt.RtHip(sc, accel=rt.ACCEL_PLOC)
d_n = torch.zeros(W * H * 4, dtype=torch.float32, device="cuda")
d_p = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda")
g.render_gbuffer_device(sc.fov, W, H, d_n, d_p); g.sync()
nr = d_n.view(-1, 4); sel = nr[:, 3] > 0
d_pos = d_p.view(-1, 3)[sel].contiguous()
d_nrm = (nr[sel, :3].double() * 2.0 - 1.0).float().contiguous()
n = d_pos.shape[0]
d_res = torch.zeros(n * 16, dtype=torch.float32, device="cuda")
d_off = torch.zeros(2 * 4096, dtype=torch.int32, device="cuda")
for _ in range(3):
    r = g.process_gbuffer_device(d_pos, d_nrm, n, d_res, d_off)
K = 20
t0 = time.perf_counter()
for _ in range(K):
    r = g.process_gbuffer_device(d_pos, d_nrm, n, d_res, d_off)
dt = (time.perf_counter() - t0) / K
print(json.dumps({"rays": n, "result": r if not hasattr(r, "shape") else None, "ms": round(dt * 1e3, 4), "mrays_s": round(n / dt / 1e6, 1)}))
