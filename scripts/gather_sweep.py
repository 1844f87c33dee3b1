#!/usr/bin/env python3
"""C5 gather time on the frame's level-0 points for combinations of the gather tuning knobs.
    gather_sweep.py photons_per_light  name=value,name=value  [name=value,...] ..."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import scenes
from raytracer_amd.photons import build_photon_map
W, H = 1920, 1080
per_light = int(sys.argv[1])
sc = scenes.atrium(262144)
g = rt.RtHip(sc, semantics="array", accel=rt.ACCEL_PLOC, use_photon_map=1, photon_radius=1.0, seed=42, profile=1)
n = build_photon_map(g, per_light, len(sc.lights), "cuda")
d_n = torch.zeros(W * H * 4, dtype=torch.float32, device="cuda")
d_p = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda")
g.render_gbuffer_device(sc.fov, W, H, d_n, d_p); g.sync()
nr = d_n.view(-1, 4)
pp = d_p.view(-1, 3)
if os.environ.get("SWEEP_TILE_ORDER", "1") != "0":   # 8 x 8 tiles, the order of the frame's level-0 points
    idx = torch.arange(W * H, device="cuda").view(H // 8, 8, W // 8, 8).permute(0, 2, 1, 3).reshape(-1)
    nr, pp = nr[idx], pp[idx]
cov = nr[:, 3] > 0
pos = pp[cov].contiguous(); nrm = (nr[cov, :3] * 2 - 1).contiguous()
out = torch.zeros_like(pos)
base = g.tuning()
for spec in sys.argv[2:]:
    kw = dict(base)
    for item in spec.split(","):
        if item:
            k, v = item.split("=")
            kw[k] = type(base[k])(float(v))
    g.set_tuning(**kw)
    if "gather_fine_cells" in spec:
        n = build_photon_map(g, per_light, len(sc.lights), "cuda")   # the grid is built when the set is installed
    g.radiance_estimate_device(pos, nrm, pos.shape[0], 100, 1.0, out); g.sync()
    g.reset_stats()
    for _ in range(3):
        g.radiance_estimate_device(pos, nrm, pos.shape[0], 100, 1.0, out)
    st = g.stats()
    fin = torch.isfinite(out)
    rec = {"tuning": spec or "default", "photons": n, "points": int(pos.shape[0]), "gather_ms": round(st["gather_ms"] / 3, 3),
           "checksum": [float(out[fin].double().sum()), int((~fin).sum())]}
    if hasattr(rt.load_library(), "rt_hip_debug_gather_probe"):   # probe builds (-DRT_GATHER_PROBE): per gather
        import ctypes
        buf = (ctypes.c_ulonglong * 16)()
        rt.load_library().rt_hip_debug_gather_probe(buf, 1)
        names = ["calls", "calls_nonempty", "rounds", "elements", "long_runs", "long_elements", "short_runs", "-"]
        rec["probe_per_gather"] = {k: v // 4 for k, v in zip(names, buf) if k != "-"}
    print(json.dumps(rec), flush=True)
