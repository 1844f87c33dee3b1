#!/usr/bin/env python3
"""Probe build (-DRT_WAVE_CENSUS, profiles/r3_variants/wave_census.patch): wave-level statistics of the any-hit traversals of
one frame: RT_HIP_LIBRARY=variants_tmp/librt_wc.so python3 scripts/wave_census_probe.py [workload]"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import raytracer_amd as rt
from raytracer_amd import scenes
from bench import WORKLOADS
wl = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "atrium-262k-1080p"]
W, H = wl["w"], wl["h"]
sc = scenes.atrium(wl["tris"])
g = rt.RtHip(sc, semantics="array", accel=rt.ACCEL_PLOC, bounces=2, do_shadows=1, shadow_t_max=1.0)
d_n = torch.zeros(W * H * 4, dtype=torch.float32, device="cuda"); d_p = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda")
g.render_gbuffer_device(sc.fov, W, H, d_n, d_p)
out = torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda")
lib = rt.load_library()
buf = (ctypes.c_ulonglong * 8)()
g.render_frame_device(d_p, d_n, W, H, out); g.sync()
lib.rt_hip_debug_wave_census(buf, 1)
g.render_frame_device(d_p, d_n, W, H, out); g.sync()
lib.rt_hip_debug_wave_census(buf, 1)
w, l, u, leaf, pre, n = [int(buf[k]) for k in range(6)]
print("any-hit traversals of one frame: %d waves, %.1f wave-iterations per wave, lanes active %.1f %%, uniform iterations %.1f %% "
      "(uniform prefix %.1f per wave), iterations with a leaf %.1f %%" % (n, w / n, 100.0 * l / (64 * w), 100.0 * u / w, pre / n, 100.0 * leaf / w))
g.close()
