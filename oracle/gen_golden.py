#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the COMPILED REFERENCE (oracle/_ref/libref_oracle.so).

Run in the build container only (needs /root/reference to have been compiled by `make -C oracle
ref`).  Fixtures carry data only -- seeded inputs and the reference's outputs -- never reference
source.  Scenes are the procedural ones of raytracer_amd/scenes.py; a digest of the geometry is
stored so that a drifting generator is caught instead of silently invalidating the vectors.

    python oracle/gen_golden.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pyoracle import Oracle, Ref  # noqa: E402
from raytracer_amd import scenes  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
MISS = 0xFFFFFFFF


def scene_digest(sc):
    h = hashlib.sha256()
    for m in sc.meshes:
        for a in (m.verts, m.normals, m.texcoords, m.faces, m.model):
            h.update(np.ascontiguousarray(a).tobytes())
        h.update(np.int32(m.material).tobytes())
        h.update(np.asarray(m.position, np.float32).tobytes())
    for mt in sc.materials:
        h.update(np.asarray(list(mt.diffuse) + list(mt.specular) + [mt.shininess, mt.reflectivity, mt.alpha, mt.ior], np.float32).tobytes())
    for li in sc.lights:
        h.update(np.asarray(list(li.position) + list(li.color) + [li.energy], np.float32).tobytes())
    return h.hexdigest()


SCENES = {
    "cornell": scenes.cornell,
    "atrium2k": lambda: scenes.atrium(2048),
    "atrium2k_xf": lambda: scenes.with_transforms(scenes.atrium(2048)),
    "atrium8k": lambda: scenes.atrium(8192),
    "spheres": scenes.sphere_field,
}


def test_rays(sc, orc, w, h, seed):
    d = orc.camera_rays(sc.fov, w, h)
    rng = np.random.RandomState(seed)
    o = rng.uniform(-1, 1, (d.shape[0], 3)).astype(np.float32)
    return np.concatenate([o, d], 1).astype(np.float32)


def gen_process(name, w, h):
    sc = SCENES[name]()
    orc, ref = Oracle(sc), Ref(sc)
    nr, po = orc.software_gbuffer(w, h)
    pos, nrm, _ = orc.gbuffer_select(nr, po)
    r = ref.process_rays(pos, nrm)
    hit = r["tidx"] != MISS
    # the reference leaves t/u/v of missed rays uninitialised (rt_raytracer.cpp:441-443): zero them
    for k in ("t", "u", "v"):
        r[k] = np.where(hit, r[k], 0).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "process_%s.npz" % name), digest=scene_digest(sc), pos=pos, nrm=nrm,
                        tidx=r["tidx"], t=r["t"], u=r["u"], v=r["v"], dir=r["dir"], records=r["records"],
                        offset_material=r["offset_material"], offset_end=r["offset_end"])
    print("process", name, len(pos), "rays", hit.sum(), "hits")


def gen_array(name, w, h):
    sc = SCENES[name]()
    orc, ref = Oracle(sc), Ref(sc)
    rays = test_rays(sc, orc, w, h, 0)
    r = ref.array_intersect(rays)
    occ = ref.array_occlude(rays)
    np.savez_compressed(os.path.join(OUT, "array_%s.npz" % name), digest=scene_digest(sc), rays=rays, hit=r["hit"],
                        material=r["material"], records=r["records"], occluded=occ)
    print("array", name, len(rays), "rays", r["hit"].sum(), "hits", occ.sum(), "occluded")


def gen_spheres():
    sc = SCENES["spheres"]()
    orc, ref = Oracle(sc), Ref(sc)
    d = orc.camera_rays(sc.fov, 96, 72)
    dn = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    rng = np.random.RandomState(1)
    pts = np.array([m.position for m in sc.meshes], np.float32)
    dirs = (pts[rng.randint(0, len(pts), 3000)] + rng.normal(0, 1.5, (3000, 3))).astype(np.float32)
    rays = np.concatenate([np.concatenate([np.zeros_like(dn), dn], 1),
                           np.concatenate([rng.uniform(-1, 1, (3000, 3)).astype(np.float32), dirs], 1)], 0)
    r = ref.spheres_intersect(rays)
    ro = np.concatenate([rng.uniform(-20, 20, (3000, 3)), rng.normal(0, 1, (3000, 3))], 1).astype(np.float32)
    occ = ref.spheres_occlude(ro)
    np.savez_compressed(os.path.join(OUT, "spheres.npz"), digest=scene_digest(sc), rays=rays, hit=r["hit"],
                        diffuse=r["diffuse"], records=r["records"], occ_rays=ro, occluded=occ)
    print("spheres", len(rays), r["hit"].sum(), "hits", occ.sum(), "occluded")


def gen_photons():
    sc = SCENES["cornell"]()
    orc, ref = Oracle(sc), Ref(sc)
    ph, seed = ref.photon_generate(3000, 0)
    nr, po = orc.software_gbuffer(48, 36)
    pos, nrm, _ = orc.gbuffer_select(nr, po)
    ref.photon_set(ph)
    est1 = ref.radiance_estimate(pos, nrm, 100, 1.0)
    est3 = ref.radiance_estimate(pos, nrm, 50, 3.0)
    look = ref.photon_lookup(pos, 1.0)
    np.savez_compressed(os.path.join(OUT, "photons_cornell.npz"), digest=scene_digest(sc), count=3000, seed=seed,
                        photons=ph, pos=pos, nrm=nrm, est_k100_r1=est1, est_k50_r3=est3, lookup_r1=look)
    print("photons", ph.shape, "seed", seed, "nan points", np.isnan(est1).any(1).sum())


def gen_camera():
    orc, ref = Oracle(), Ref()
    out = {}
    for (w, h) in ((640, 480), (1920, 1080)):
        d = ref.camera_rays(0.86, w, h)
        idx = np.array([0, w - 1, w * (h - 1), w * h - 1, w * (h // 2) + w // 2, 12345 % (w * h)])
        out["idx_%dx%d" % (w, h)] = idx
        out["dir_%dx%d" % (w, h)] = d[idx]
        out["xor_%dx%d" % (w, h)] = np.bitwise_xor.reduce(d.view(np.uint32), axis=0)
        out["sum_%dx%d" % (w, h)] = d.astype(np.float64).sum(0)
    np.savez_compressed(os.path.join(OUT, "camera.npz"), **out)
    print("camera ok")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    # Cornell, a 2k- and an 8k-triangle scene x 16k-64k G-buffer samples (160 x 120 = 19,200 texels)
    for n in ("cornell", "atrium2k", "atrium2k_xf", "atrium8k"):
        gen_process(n, 160, 120)
    for n in ("cornell", "atrium2k", "atrium8k"):
        gen_array(n, 80, 60)
    gen_spheres()
    gen_photons()
    gen_camera()
