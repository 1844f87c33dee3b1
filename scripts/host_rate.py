#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-buffer entry points (never bench.py's `value`): the 1080p
G-buffer goes up (58 MB) and the image or the hit records come down on every call."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import scenes

W, H = 1920, 1080
sc = scenes.atrium(262144)
g = rt.RtHip(sc, semantics="array", accel=rt.ACCEL_PLOC, bounces=2, do_shadows=1, shadow_t_max=1.0)
d_n = torch.zeros(W * H * 4, dtype=torch.float32, device="cuda")
d_p = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda")
g.render_gbuffer_device(sc.fov, W, H, d_n, d_p)
g.sync()
nr = d_n.cpu().numpy().reshape(-1, 4)
po = d_p.cpu().numpy().reshape(-1, 3)
out = {}
for _ in range(2):
    g.render_frame(po, nr, W, H)
t0 = time.perf_counter()
K = 10
for _ in range(K):
    g.render_frame(po, nr, W, H)
dt = (time.perf_counter() - t0) / K
out["render_frame_host_ms"] = round(dt * 1e3, 3)
out["render_frame_host_mrays_s"] = round(g.last_frame_rays() / dt / 1e6, 1)
sel = nr[:, 3] > 0
pos = po[sel]
nrm = (nr[sel, :3].astype(np.float64) * 2.0 - 1.0).astype(np.float32)
g2 = rt.RtHip(sc, accel=rt.ACCEL_PLOC)
for _ in range(2):
    r = g2.process_rays(pos, nrm)
t0 = time.perf_counter()
for _ in range(K):
    r = g2.process_rays(pos, nrm)
dt = (time.perf_counter() - t0) / K
out["process_gbuffer_host_ms"] = round(dt * 1e3, 3)
out["process_gbuffer_host_mrays_s"] = round(pos.shape[0] / dt / 1e6, 1)
# the C call alone (the ctypes wrapper above also copies the 125 MB of records into numpy arrays)
import ctypes as C
res_p, off_p, nres, nmat = C.c_void_p(), C.c_void_p(), C.c_int(0), C.c_int(0)
fp = C.POINTER(C.c_float)
t0 = time.perf_counter()
for _ in range(K):
    g2._ck(g2.lib.rt_hip_process_gbuffer(g2.h, pos.ctypes.data_as(fp), nrm.ctypes.data_as(fp), pos.shape[0], C.byref(res_p),
                                         C.byref(off_p), C.byref(nres), C.byref(nmat)))
dt = (time.perf_counter() - t0) / K
out["process_gbuffer_c_call_ms"] = round(dt * 1e3, 3)
out["process_gbuffer_c_call_mrays_s"] = round(pos.shape[0] / dt / 1e6, 1)
out["process_gbuffer_rays"] = int(pos.shape[0])
out["process_gbuffer_hits"] = int(r["records"].shape[0])
# the same call with device-resident inputs and outputs (no PCIe): processRays proper
d_pos = torch.from_numpy(pos).cuda(); d_nrm = torch.from_numpy(nrm).cuda()
d_res = torch.zeros(pos.shape[0] * 16, dtype=torch.float32, device="cuda")
d_off = torch.zeros(2 * 4096, dtype=torch.int32, device="cuda")
for _ in range(2):
    g2.process_gbuffer_device(d_pos, d_nrm, pos.shape[0], d_res, d_off)
g2.sync()
t0 = time.perf_counter()
for _ in range(K):
    nres, nmat = g2.process_gbuffer_device(d_pos, d_nrm, pos.shape[0], d_res, d_off)
g2.sync()
dt = (time.perf_counter() - t0) / K
out["process_gbuffer_device_ms"] = round(dt * 1e3, 3)
out["process_gbuffer_device_mrays_s"] = round(pos.shape[0] / dt / 1e6, 1)
out["process_gbuffer_groups"] = int(nmat)
# the live (Gen-2) frame on the device: select -> ray-gen -> closest hit -> raypass -> composite
d_rp = torch.full((W * H, 3), 255, dtype=torch.uint8, device="cuda")
d_img = torch.zeros((W * H, 4), dtype=torch.uint8, device="cuda")
for interlace in (1, 16):
    for _ in range(2):
        g2.live_frame_device(d_p, d_n, W, H, interlace, 0, d_rp, d_img)
    t0 = time.perf_counter()
    for k in range(K):
        rays = g2.live_frame_device(d_p, d_n, W, H, interlace, k % interlace, d_rp, d_img)
    dt = (time.perf_counter() - t0) / K
    out["live_frame_interlace%d_ms" % interlace] = round(dt * 1e3, 3)
    out["live_frame_interlace%d_rays" % interlace] = rays
print(json.dumps(out))
