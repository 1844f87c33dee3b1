"""The HIP path straight against the golden vectors captured from the compiled reference
(tests/golden/, see oracle/gen_golden.py) -- no oracle in the loop."""
import os

import numpy as np
import pytest

from conftest import assert_bits_equal
from parity_util import assert_only_edge_ties
from test_oracle_golden import SCENES, groups, load, scene_for

pytestmark = pytest.mark.gpu
MISS = 0xFFFFFFFF
ID_MISMATCHES_VS_DACRT = {"cornell": 2, "atrium2k": 0, "atrium2k_xf": 0, "atrium8k": 0}


@pytest.mark.parametrize("name", ["cornell", "atrium2k", "atrium2k_xf", "atrium8k"])
@pytest.mark.parametrize("accel", [1, 2, 3], ids=["brute", "lbvh", "ploc"])
def test_process_rays_vs_reference_vectors(name, accel):
    """RayTracer::processRays of the compiled reference (DACRT) on 15k-19k G-buffer samples."""
    import raytracer_amd as rt
    g = load("process_" + name)
    sc = scene_for(name, g)
    h = rt.RtHip(sc, accel=accel)
    r = h.process_rays(g["pos"], g["nrm"])
    assert_bits_equal(r["dir"], g["dir"], "dir")
    hit = g["tidx"] != MISS
    np.testing.assert_array_equal(r["tidx"] != MISS, hit)                  # hit / miss: 100 %
    assert_bits_equal(r["t"][hit], g["t"][hit], "t")                        # t: bit-exact (bar: 1e-5 rel)
    same = r["tidx"] == g["tidx"]
    # ids: north_star's bar -- every mismatch is a bit-equal-t tie between triangles that share an edge
    rate = assert_only_edge_ties(sc, r["tidx"][hit], g["tidx"][hit], r["t"][hit], g["t"][hit],
                                 "golden/process_%s/accel%d vs DACRT" % (name, accel))
    # ... and the stated figures (profiles/r2_parity_id_match.json): none on the atrium fixtures, 2 of
    # 14,764 hit rays on Cornell (99.9865 %; both bit-equal-t shared-edge ties, which assert_only_edge_ties has just
    # proved).  The HIP id is the lowest of the tied ones whatever the builder, so the count is exact, not a bound.
    assert int((~same[hit]).sum()) == ID_MISMATCHES_VS_DACRT[name], (name, int((~same[hit]).sum()), rate)
    both = hit & same
    assert_bits_equal(r["u"][both], g["u"][both], "u")
    assert_bits_equal(r["v"][both], g["v"][both], "v")
    if same.all():
        ga = groups(r["offset_material"], r["offset_end"], r["records"])
        gb = groups(g["offset_material"], g["offset_end"], g["records"])
        assert sorted(ga) == sorted(gb)
        for m in ga:
            for f in ("origin", "direction", "position", "texcoord"):
                assert_bits_equal(ga[m][f], gb[m][f], f)
            np.testing.assert_allclose(ga[m]["normal"], gb[m]["normal"], atol=1e-5, rtol=0)
    h.close()


@pytest.mark.parametrize("name", ["cornell", "atrium2k", "atrium8k"])
def test_array_vs_reference_vectors(name):
    import raytracer_amd as rt
    g = load("array_" + name)
    sc = scene_for(name, g)
    h = rt.RtHip(sc, semantics="array")
    r = h.intersect(g["rays"])
    np.testing.assert_array_equal(r["hit"], g["hit"])
    np.testing.assert_array_equal(r["material"], g["material"])
    for f in ("origin", "direction", "position", "normal", "texcoord"):
        assert_bits_equal(r["records"][f], g["records"][f], f)
    np.testing.assert_array_equal(h.occlude(g["rays"]), g["occluded"])
    h.close()


def test_spheres_vs_reference_vectors():
    import raytracer_amd as rt
    g = load("spheres")
    sc = scene_for("spheres", g)
    h = rt.RtHip(sc, geometry=rt.SPHERES)
    r = h.intersect(g["rays"])
    np.testing.assert_array_equal(r["hit"], g["hit"])
    for f in ("position", "normal"):
        assert_bits_equal(r["records"][f], g["records"][f], f)
    hit = g["hit"] > 0
    cols = h.sphere_colors(len(sc.meshes))
    assert_bits_equal(cols[r["material"][hit]], g["diffuse"][hit], "srand(420) colours")
    np.testing.assert_array_equal(h.occlude(g["occ_rays"]), g["occluded"])
    h.close()
