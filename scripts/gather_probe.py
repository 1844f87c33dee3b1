#!/usr/bin/env python3
"""Where the photon gather's time goes: full run vs no-selection (count = huge) on the C5 frame's points."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import scenes
from raytracer_amd.photons import build_photon_map
W, H = 1920, 1080
per_light = int(sys.argv[1]) if len(sys.argv) > 1 else 250000
sc = scenes.atrium(262144)
g = rt.RtHip(sc, semantics="array", accel=rt.ACCEL_PLOC, use_photon_map=1, photon_radius=1.0, seed=42)
n = build_photon_map(g, per_light, len(sc.lights), "cuda")
d_n = torch.zeros(W * H * 4, dtype=torch.float32, device="cuda")
d_p = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda")
g.render_gbuffer_device(sc.fov, W, H, d_n, d_p); g.sync()
nr = d_n.view(-1, 4); cov = nr[:, 3] > 0
pos = d_p.view(-1, 3)[cov].contiguous(); nrm = (nr[cov, :3] * 2 - 1).contiguous()
out = torch.zeros_like(pos)
res = {"photons": n, "points": int(pos.shape[0])}
for name, count in (("k100", 100), ("k_all", 1 << 30)):
    g.radiance_estimate_device(pos, nrm, pos.shape[0], count, 1.0, out); g.sync()
    t0 = time.perf_counter()
    g.radiance_estimate_device(pos, nrm, pos.shape[0], count, 1.0, out); g.sync()
    res[name + "_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
print(json.dumps(res))
