#!/usr/bin/env python3
"""One rank's share of a split frame through the library's pipeline with F frames in flight (tuning.frame_streams = F,
rt_hip_tiles_packed_enqueue_device): python3 scripts/share_lanes_probe.py [nranks] [rank] [workload]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import raytracer_amd as rt
from raytracer_amd import scenes
from raytracer_amd.tiles import max_owned_tiles
from bench import WORKLOADS
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
r = int(sys.argv[2]) if len(sys.argv) > 2 else 0
wl = WORKLOADS[sys.argv[3] if len(sys.argv) > 3 else "atrium-262k-1080p"]
W, H = wl["w"], wl["h"]
sc = scenes.atrium(wl["tris"])
g = rt.RtHip(sc, semantics="array", accel=rt.ACCEL_PLOC, bounces=2, do_shadows=1, shadow_t_max=1.0)
d_n = torch.zeros(W * H * 4, dtype=torch.float32, device="cuda"); d_p = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda")
g.render_gbuffer_device(sc.fov, W, H, d_n, d_p)
chunk = max_owned_tiles(W, H, 64, n) * 64 * 64 * 4
enq = type(g).tiles_packed_enqueue_device.__wrapped__
for F in (1, 2, 3, 4, 6, 8):
    g.set_tuning(frame_streams=F)
    lim = max(2, F)
    dsts = [torch.zeros(chunk, dtype=torch.uint8, device="cuda") for _ in range(lim)]
    def piped(k):
        pend = 0
        for i in range(k):
            if pend == lim:
                g.frame_finish(); pend -= 1
            enq(g, d_p, d_n, W, H, 64, r, n, dsts[i % lim]); pend += 1
        while pend:
            g.frame_finish(); pend -= 1
        g.sync()
    piped(16)
    t0 = time.perf_counter(); piped(200); dt = (time.perf_counter() - t0) / 200 * 1e3
    print("rank %d of %d, %d streams: %.4f ms per frame share" % (r, n, F, dt), flush=True)
g.close()
