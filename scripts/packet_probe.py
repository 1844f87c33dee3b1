#!/usr/bin/env python3
"""Statistics of the wave-packet shadow walk (tuning.shadow_packet) on a bench workload: per budget, the packets walked,
visits and leaf visits per packet, the share of rays handed to the per-lane traversal, the frame time -- and that the
image is the per-lane one bit for bit.   RT_HIP_PACKET_STATS=1 python3 scripts/packet_probe.py [workload] [budgets]"""
import os
import sys
import time

os.environ.setdefault("RT_HIP_PACKET_STATS", "1")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import raytracer_amd as rt  # noqa: E402
from raytracer_amd import scenes  # noqa: E402
from bench import WORKLOADS  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "atrium-262k-1080p"
budgets = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 32, 64, 96, 128, 192, 256, 1024]
wl = WORKLOADS[name]
W, H = wl["w"], wl["h"]
dev = torch.device("cuda", 0)
scene = scenes.cornell() if wl["tris"] == 0 else scenes.atrium(wl["tris"])
g = rt.RtHip(scene, semantics="array", accel=rt.ACCEL_PLOC, device=0, profile=1, bounces=2, do_shadows=1, shadow_t_max=1.0, seed=42)
npix = W * H
d_n = torch.zeros(npix * 4, dtype=torch.float32, device=dev)
d_p = torch.zeros(npix * 3, dtype=torch.float32, device=dev)
g.render_gbuffer_device(scene.fov, W, H, d_n, d_p)
g.sync()
ref = torch.zeros(npix * 4, dtype=torch.uint8, device=dev)
out = torch.zeros(npix * 4, dtype=torch.uint8, device=dev)
g.set_tuning(shadow_packet=0)
g.render_frame_device(d_p, d_n, W, H, ref)
g.sync()
for b in budgets:
    g.set_tuning(shadow_packet=b)
    g.set_config(census=0, profile=1)
    for _ in range(3):
        g.render_frame_device(d_p, d_n, W, H, out)
    g.sync()
    g.reset_stats()
    t0 = time.perf_counter()
    for _ in range(50):
        g.render_frame_device(d_p, d_n, W, H, out)
    g.sync()
    ms = (time.perf_counter() - t0) / 50 * 1e3
    st = g.stats()
    same = bool(torch.equal(out, ref))
    line = "budget %5d  frame %.4f ms  k_shadow %.4f ms/launch  image identical %s" % (b, ms, st["any_ms"] / max(st["any_launches"], 1), same)
    if b > 0:
        g.set_config(census=1, profile=0)
        g.reset_stats()
        g.render_frame_device(d_p, d_n, W, H, out)
        g.sync()
        ps = list(g.stats()["packet_stats"])
        rays = g.last_frame_rays()
        if ps[0]:
            line += "  | packets %d  visits/packet %.1f  leaves/packet %.1f  per-lane rays %d (%.2f %% of %d)" % (
                ps[0], ps[1] / ps[0], ps[2] / ps[0], ps[3], 100.0 * ps[3] / max(rays, 1), rays)
    print(line, flush=True)
g.close()
