#!/usr/bin/env python3
"""One virtual rank's share of a strong-scaling frame, repeated (for a kernel trace of the split frame's timeline):
    python3 scripts/share_probe.py [nranks] [rank] [reps] [workload]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import raytracer_amd as rt
from raytracer_amd import scenes
from raytracer_amd.tiles import max_owned_tiles
from bench import WORKLOADS

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
r = int(sys.argv[2]) if len(sys.argv) > 2 else 0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
wl = WORKLOADS[sys.argv[4] if len(sys.argv) > 4 else "atrium-262k-1080p"]
W, H = wl["w"], wl["h"]
sc = scenes.atrium(wl["tris"])
g = rt.RtHip(sc, semantics="array", accel=rt.ACCEL_PLOC, bounces=2, do_shadows=1, shadow_t_max=1.0)
d_n = torch.zeros(W * H * 4, dtype=torch.float32, device="cuda")
d_p = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda")
g.render_gbuffer_device(sc.fov, W, H, d_n, d_p)
chunk = max_owned_tiles(W, H, 64, n) * 64 * 64 * 4
dst = [torch.zeros(chunk, dtype=torch.uint8, device="cuda")]
packed = type(g).render_tiles_packed_device.__wrapped__
for _ in range(5):
    packed(g, [d_p], [d_n], W, H, 64, r, n, dst)
g.sync()
t0 = time.perf_counter()
for _ in range(reps):
    packed(g, [d_p], [d_n], W, H, 64, r, n, dst)
g.sync()
print("rank %d of %d: %.4f ms per call, %d rays" % (r, n, (time.perf_counter() - t0) / reps * 1e3, g.last_frame_rays()))
g.close()
