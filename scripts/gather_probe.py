#!/usr/bin/env python3
"""Where the photon gather's time goes on the C5 frame: per density class of the shading points (photons within the
radius, counted on the host for a sample), for the coarse-grid / full-pass and the fine-grid / shell-only wave kernels.
    gather_probe.py [photons_per_light]"""
import os, sys, time, json
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
n = build_photon_map(g, per_light, len(sc.lights), "cuda")
ph = g.photon_get()[:, :3]
d_n = torch.zeros(W * H * 4, dtype=torch.float32, device="cuda")
d_p = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda")
g.render_gbuffer_device(sc.fov, W, H, d_n, d_p); g.sync()
nr = d_n.view(-1, 4); cov = nr[:, 3] > 0
pos = d_p.view(-1, 3)[cov].contiguous(); nrm = (nr[cov, :3] * 2 - 1).contiguous()
res = {"photons": n, "points": int(pos.shape[0])}


def timed(p, q, reps=3):
    out = torch.zeros_like(p)
    g.radiance_estimate_device(p, q, p.shape[0], 100, 1.0, out); g.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.radiance_estimate_device(p, q, p.shape[0], 100, 1.0, out)
    g.sync()
    return (time.perf_counter() - t0) / reps * 1e3


for fine in (0, 1):
    g.set_tuning(gather_fine=fine)
    res["all_points_fine%d_ms" % fine] = round(timed(pos, nrm), 2)
# density classes on a sample
rng = np.random.RandomState(0)
pick = rng.choice(pos.shape[0], 20000, replace=False)
tree = cKDTree(ph)
cnt = tree.query_ball_point(pos[pick].cpu().numpy(), 1.0, return_length=True, workers=16)
res["share_of_points"], res["share_of_photon_visits"], res["us_per_point"] = {}, {}, {}
for lo, hi in ((0, 1), (1, 300), (300, 1000), (1000, 3000), (3000, 10000), (10000, 30000), (30000, 100000), (100000, 10**9)):
    sel = np.nonzero((cnt >= lo) & (cnt < hi))[0]
    key = "%d-%d" % (lo, hi)
    res["share_of_points"][key] = round(len(sel) / len(cnt), 4)
    res["share_of_photon_visits"][key] = round(float(cnt[sel].sum() / cnt.sum()), 4)
    if len(sel) == 0:
        continue
    idx = torch.from_numpy(pick[sel][np.arange(65536) % len(sel)]).cuda()
    p, q = pos[idx].contiguous(), nrm[idx].contiguous()
    row = {}
    for fine in (0, 1):
        g.set_tuning(gather_fine=fine)
        row["fine%d" % fine] = round(timed(p, q) / 65536 * 1e3, 3)
    res["us_per_point"][key] = row
print(json.dumps(res, indent=1))
