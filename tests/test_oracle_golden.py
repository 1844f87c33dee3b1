"""The CPU restatement (oracle/rt_oracle.cpp) against the golden vectors that
oracle/gen_golden.py captured from the COMPILED REFERENCE.  Runs anywhere (no GPU, no
/root/reference).  Bar: bit-exact, except hit-record normals under non-trivial model matrices
(the reference inverts with Eigen's SSE path; tolerance 1e-5)."""
import os

import numpy as np
import pytest

from conftest import assert_bits_equal
from raytracer_amd import scenes

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MISS = 0xFFFFFFFF

SCENES = {
    "cornell": scenes.cornell,
    "atrium2k": lambda: scenes.atrium(2048),
    "atrium2k_xf": lambda: scenes.with_transforms(scenes.atrium(2048)),
    "atrium8k": lambda: scenes.atrium(8192),
    "spheres": scenes.sphere_field,
}


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def scene_for(name, gold):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(GOLD), "..", "oracle"))
    from oracle.gen_golden import scene_digest
    sc = SCENES[name]()
    assert scene_digest(sc) == str(gold["digest"]), "procedural scene drifted from the one the fixture was made on"
    return sc


def groups(offset_material, offset_end, records):
    out, s = {}, 0
    for m, e in zip(offset_material, offset_end):
        out[int(m)] = records[s:e]
        s = e
    return out


@pytest.mark.parametrize("name", ["cornell", "atrium2k", "atrium2k_xf", "atrium8k"])
def test_process_rays_golden(oracle_cls, name):
    g = load("process_" + name)
    sc = scene_for(name, g)
    r = oracle_cls(sc).process_rays(g["pos"], g["nrm"])
    np.testing.assert_array_equal(r["tidx"], g["tidx"])
    for k in ("t", "u", "v", "dir"):
        assert_bits_equal(r[k], g[k], k)
    # the reference orders groups by Material* address; compare group by group
    np.testing.assert_array_equal(np.sort(r["offset_material"]), np.sort(g["offset_material"]))
    ga = groups(r["offset_material"], r["offset_end"], r["records"])
    gb = groups(g["offset_material"], g["offset_end"], g["records"])
    for m in ga:
        assert len(ga[m]) == len(gb[m])
        for f in ("origin", "direction", "position", "texcoord"):
            assert_bits_equal(ga[m][f], gb[m][f], "%s material %d" % (f, m))
        if name.endswith("_xf"):
            np.testing.assert_allclose(ga[m]["normal"], gb[m]["normal"], atol=1e-5, rtol=0)
        else:
            assert_bits_equal(ga[m]["normal"], gb[m]["normal"], "normal material %d" % m)


@pytest.mark.parametrize("name", ["cornell", "atrium2k", "atrium8k"])
def test_array_golden(oracle_cls, name):
    g = load("array_" + name)
    sc = scene_for(name, g)
    orc = oracle_cls(sc)
    r = orc.array_intersect(g["rays"])
    np.testing.assert_array_equal(r["hit"], g["hit"])
    np.testing.assert_array_equal(r["material"], g["material"])
    for f in ("origin", "direction", "position", "normal", "texcoord"):
        assert_bits_equal(r["records"][f], g["records"][f], f)
    np.testing.assert_array_equal(orc.array_occlude(g["rays"]), g["occluded"])


def test_spheres_golden(oracle_cls):
    g = load("spheres")
    sc = scene_for("spheres", g)
    orc = oracle_cls(sc)
    r = orc.spheres_intersect(g["rays"])
    np.testing.assert_array_equal(r["hit"], g["hit"])
    assert_bits_equal(r["diffuse"], g["diffuse"], "srand(420) colours")
    for f in ("position", "normal"):
        assert_bits_equal(r["records"][f], g["records"][f], f)
    np.testing.assert_array_equal(orc.spheres_occlude(g["occ_rays"]), g["occluded"])


def test_photons_golden(oracle_cls):
    g = load("photons_cornell")
    sc = scene_for("cornell", g)
    orc = oracle_cls(sc)
    ph = orc.photon_generate(int(g["count"]), geometry=0, rng=0, seed=int(g["seed"]))   # libc stream, same seed
    assert_bits_equal(ph, g["photons"], "photons")
    orc.photon_set(g["photons"])
    e1 = orc.radiance_estimate(g["pos"], g["nrm"], 100, 1.0)
    assert np.isnan(e1).any(), "the empty-neighbourhood NaN case (rt_photon_map.cpp:204) must be covered"
    assert_bits_equal(e1, g["est_k100_r1"], "estimate k=100 r=1")
    assert_bits_equal(orc.radiance_estimate(g["pos"], g["nrm"], 50, 3.0), g["est_k50_r3"], "estimate k=50 r=3")
    assert_bits_equal(orc.photon_lookup(g["pos"], 1.0), g["lookup_r1"], "lookup")


def test_camera_golden(oracle_cls):
    g = load("camera")
    orc = oracle_cls()
    for (w, h) in ((640, 480), (1920, 1080)):
        d = orc.camera_rays(0.86, w, h)
        key = "%dx%d" % (w, h)
        assert_bits_equal(d[g["idx_" + key]], g["dir_" + key], "camera samples")
        np.testing.assert_array_equal(np.bitwise_xor.reduce(d.view(np.uint32), axis=0), g["xor_" + key])
