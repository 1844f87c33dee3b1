#!/usr/bin/env python3
"""Distribution of the number of photons within the gather radius at the C5 frame's shading points
(sample), to size the gather's load balancing."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import scenes
from raytracer_amd.photons import build_photon_map
from scipy.spatial import cKDTree
W, H = 1920, 1080
per_light = int(sys.argv[1]) if len(sys.argv) > 1 else 2500000
sc = scenes.atrium(262144)
g = rt.RtHip(sc, semantics="array", accel=rt.ACCEL_PLOC, use_photon_map=1, photon_radius=1.0, seed=42)
first, count = 0, per_light * len(sc.lights)
buf = torch.empty((count, 9), dtype=torch.float32, device="cuda")
stored = g.photon_emit_device(per_light, first, count, buf, count)
ph = buf[:stored, :3].cpu().numpy()
d_n = torch.zeros(W * H * 4, dtype=torch.float32, device="cuda")
d_p = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda")
g.render_gbuffer_device(sc.fov, W, H, d_n, d_p); g.sync()
cov = (d_n.view(-1, 4)[:, 3] > 0).cpu().numpy()
pos = d_p.view(-1, 3).cpu().numpy()[cov]
rng = np.random.RandomState(0)
pick = rng.choice(len(pos), 20000, replace=False)
tree = cKDTree(ph)
cnt = tree.query_ball_point(pos[pick], 1.0, return_length=True, workers=16)
q = [0, 10, 50, 90, 99, 99.9, 100]
print(json.dumps({"photons": int(stored), "points": int(len(pos)), "in_radius_percentiles": dict(zip(map(str, q), np.percentile(cnt, q).tolist())),
                  "mean": float(cnt.mean()), "share_of_work_top1pct": float(np.sort(cnt)[-200:].sum() / cnt.sum())}))
