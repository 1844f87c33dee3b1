#!/usr/bin/env python3
"""Per-update device and host time of rt_hip_update for unchanged / moved / rebuilt scenes (atrium-262k).
Under rocprofv3 --kernel-trace --stats this gives the kernel breakdown of the refit path."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
from raytracer_amd import scenes

tris = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
accel = {"ploc": rt.ACCEL_PLOC, "lbvh": rt.ACCEL_LBVH}[sys.argv[2] if len(sys.argv) > 2 else "ploc"]
sc = scenes.atrium(tris)
g = rt.RtHip(sc, semantics="array", accel=accel, use_model_mat=1)
mats = np.stack([m.model_colmajor() for m in sc.meshes])
print("build", g.stats()["build_ms"])
for label in ("moved", "rebuild", "moved"):
    host, dev = [], []
    for i in range(12):
        mm = mats.copy(); mm[:, 12] += 1e-3 * (1 + (i & 1))
        g.set_model_mats(0, mm)
        if label == "rebuild":
            g.invalidate()
        g.sync()
        t0 = time.perf_counter()
        g.update()
        g.sync()
        host.append((time.perf_counter() - t0) * 1e3)
        st = g.stats()
        dev.append(st["build_ms"])
    print(label, "kind", st["last_update"], "host ms median %.3f" % np.median(host[2:]), "device ms median %.3f" % np.median(dev[2:]),
          "cost %.3f / built %.3f" % (st["bvh_area_cost"], st["bvh_area_cost_built"]))
g.close()
