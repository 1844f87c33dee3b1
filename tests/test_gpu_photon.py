"""GPU photon pass vs the oracle (rt_photon_map.cpp restated and pinned bit for bit to the
compiled reference, see test_oracle_golden.py::test_photons_golden).

Emission uses the counter RNG (the reference's serial libc stream cannot be sharded), so it is
compared with the oracle running the SAME counter RNG: bit-exact.  The gather is compared with
the oracle fed the identical photons: the selected set is identical, the float sum runs in grid
order instead of ascending distance, hence rtol 2e-5."""
import numpy as np
import pytest

from conftest import assert_bits_equal
from raytracer_amd import scenes
from raytracer_amd.tiles import photon_shard

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    return torch


def _emit(g, per_light, first, count):
    torch = _torch()
    out = torch.zeros(max(count, 1) * 9, dtype=torch.float32, device="cuda")
    n = g.photon_emit_device(per_light, first, count, out, count)
    return out.cpu().numpy().reshape(-1, 9)[:n]


@pytest.mark.parametrize("scene_fn,accel", [(scenes.cornell, 1), (scenes.cornell, 2), (lambda: scenes.atrium(2048), 2)],
                         ids=["cornell-brute", "cornell-lbvh", "atrium2k-lbvh"])
def test_emission_matches_oracle_counter_rng(oracle_cls, scene_fn, accel):
    import raytracer_amd as rt
    sc = scene_fn()
    per_light = 3000
    exp = oracle_cls(sc).photon_generate(per_light, geometry=0, rng=1, seed=42)
    g = rt.RtHip(sc, semantics="array", accel=accel, seed=42)
    got = _emit(g, per_light, 0, per_light * len(sc.lights))
    assert len(exp) > per_light // 2
    assert_bits_equal(got, exp, "photons")
    g.close()


def test_emission_is_shard_invariant(oracle_cls):
    import raytracer_amd as rt
    sc = scenes.atrium(2048)
    per_light = 2500
    total = per_light * len(sc.lights)
    g = rt.RtHip(sc, semantics="array", accel=2, seed=7)
    whole = _emit(g, per_light, 0, total)
    for nranks in (2, 3, 8):
        parts = [_emit(g, per_light, *photon_shard(total, r, nranks)) for r in range(nranks)]
        assert_bits_equal(np.concatenate(parts, 0), whole, "all-gathered shards, %d ranks" % nranks)
    assert len(_emit(g, per_light, 17, 0)) == 0
    g.close()


@pytest.mark.parametrize("medium", [None, 0], ids=["medium-default", "shell-walk-forced"])
@pytest.mark.parametrize("heavy", [None, 0, 4000000000], ids=["default-split", "all-wave", "all-lane"])
# (400, ...): more photons wanted than the wave kernels queue in LDS for the sum (256): the overflow branch sums on the spot
@pytest.mark.parametrize("count,radius", [(100, 1.0), (50, 3.0), (7, 0.5), (400, 3.0), (400, 0.7)])
def test_gather_matches_oracle(oracle_cls, count, radius, heavy, medium):
    """Both gather kernels against the oracle: by default a point goes to the wave-per-point kernel when its
    cube holds more than tuning.gather_heavy (32) photons; tuning.gather_heavy = 0 / huge sends every point one way.
    tuning.gather_medium = 0 forces every queued point through the fine grid's shell walk (interiors counted from the
    cell table) + the lean selection kernel -- the path that at full size only the densest points take; the default
    (4096) reads these small neighbourhoods whole from the coarse grid instead."""
    import raytracer_amd as rt
    sc = scenes.cornell()
    orc = oracle_cls(sc)
    ph = orc.photon_generate(6000, geometry=0, rng=1, seed=42)
    nr, po = orc.software_gbuffer(64, 48)
    pos, nrm, _ = orc.gbuffer_select(nr, po)
    orc.photon_set(ph)
    exp = orc.radiance_estimate(pos, nrm, count, radius)
    g = rt.RtHip(sc, semantics="array", photon_radius=radius)
    if heavy is not None:
        g.set_tuning(gather_heavy=heavy)
        assert g.tuning()["gather_heavy"] == heavy
    if medium is not None:
        g.set_tuning(gather_medium=medium)
        assert g.tuning()["gather_medium"] == medium
    g.photon_set(ph)
    got = g.radiance_estimate(pos, nrm, count, radius)
    np.testing.assert_array_equal(np.isnan(exp), np.isnan(got))   # the empty-neighbourhood NaN (rt_photon_map.cpp:204)
    ok = ~np.isnan(exp)
    assert ok.mean() > 0.5
    np.testing.assert_allclose(got[ok], exp[ok], rtol=2e-5, atol=1e-9)
    # a radius larger than the grid cell the photons were binned with still gathers correctly
    got2 = g.radiance_estimate(pos, nrm, count, radius * 2.5)
    exp2 = orc.radiance_estimate(pos, nrm, count, radius * 2.5)
    ok2 = ~np.isnan(exp2)
    np.testing.assert_allclose(got2[ok2], exp2[ok2], rtol=2e-5, atol=1e-9)
    g.close()


@pytest.mark.parametrize("count,radius", [(100, 1.0), (50, 3.0), (400, 3.0)])
def test_gather_first_pass_two_points_at_once(oracle_cls, count, radius):
    """tuning.gather_pair: the first pass (count and max_dist2, rt_photon_map.cpp:159-169) for two neighbouring queue
    entries in one walk, the distances of a photon to both points in packed FP32.  gather_pair = 0 (every point alone),
    the default width, and a width that pairs whatever is adjacent in the queue: the same bits from all of them (a
    point's count and max_dist2 do not depend on its partner), and the oracle's values.  gather_medium = 0 sends every
    point through the shell walk, where the pairs are formed; an odd number of points leaves the last one alone."""
    import raytracer_amd as rt
    sc = scenes.cornell()
    orc = oracle_cls(sc)
    ph = orc.photon_generate(8000, geometry=0, rng=1, seed=11)
    nr, po = orc.software_gbuffer(64, 48)
    pos, nrm, _ = orc.gbuffer_select(nr, po)
    if pos.shape[0] % 2 == 0:
        pos, nrm = pos[:-1], nrm[:-1]
    orc.photon_set(ph)
    exp = orc.radiance_estimate(pos, nrm, count, radius)
    ok = ~np.isnan(exp)
    assert ok.mean() > 0.5
    outs = {}
    for pair in (0, 100, 25, 100000):
        for medium in (0, 4096, 200):
            g = rt.RtHip(sc, semantics="array", photon_radius=radius, tuning=dict(gather_pair=pair, gather_medium=medium, gather_heavy=0))
            assert g.tuning()["gather_pair"] == pair
            g.photon_set(ph)
            outs[pair, medium] = g.radiance_estimate(pos, nrm, count, radius)
            g.close()
    for key, got in outs.items():
        np.testing.assert_array_equal(np.isnan(exp), np.isnan(got), err_msg=str(key))
        np.testing.assert_allclose(got[ok], exp[ok], rtol=2e-5, atol=1e-9, err_msg=str(key))
        np.testing.assert_array_equal(got[ok], outs[0, 0][ok], err_msg=str(key))


def test_gather_queue_sorted_by_position(oracle_cls):
    """tuning.gather_sort: the wave kernels' queue in Morton order of the points' fine-grid cells.  The points are handed
    over in random order (no two neighbours adjacent); unsorted, sorted, and sorted with every pair formed (which the sort
    makes possible): the same bits, and the oracle's values -- a result is written by point index and a point's count,
    max_dist2 and selection do not depend on who its neighbours in the queue are."""
    import raytracer_amd as rt
    sc = scenes.cornell()
    orc = oracle_cls(sc)
    ph = orc.photon_generate(8000, geometry=0, rng=1, seed=3)
    nr, po = orc.software_gbuffer(64, 48)
    pos, nrm, _ = orc.gbuffer_select(nr, po)
    perm = np.random.RandomState(5).permutation(pos.shape[0])
    pos, nrm = np.ascontiguousarray(pos[perm]), np.ascontiguousarray(nrm[perm])
    orc.photon_set(ph)
    exp = orc.radiance_estimate(pos, nrm, 100, 1.0)
    ok = ~np.isnan(exp)
    outs = {}
    for sort, pair, heavy in ((0, 100, 0), (1, 100, 0), (2, 100, 0), (2, 0, 0), (2, 800, 0), (0, 100, 32), (1, 100, 32)):
        g = rt.RtHip(sc, semantics="array", photon_radius=1.0, tuning=dict(gather_sort=sort, gather_pair=pair, gather_medium=0, gather_heavy=heavy))
        assert g.tuning()["gather_sort"] == sort
        g.photon_set(ph)
        outs[sort, pair, heavy] = g.radiance_estimate(pos, nrm, 100, 1.0)
        g.close()
    for key, got in outs.items():
        np.testing.assert_array_equal(np.isnan(exp), np.isnan(got), err_msg=str(key))
        np.testing.assert_allclose(got[ok], exp[ok], rtol=2e-5, atol=1e-9, err_msg=str(key))
        # (the lane-per-point kernel of the light points sums in another order than the wave kernels: compare like with like)
        np.testing.assert_array_equal(got[ok], outs[0, 100, key[2]][ok], err_msg=str(key))


@pytest.mark.parametrize("medium", [4096, 0], ids=["medium-default", "shell-walk-forced"])
@pytest.mark.parametrize("heavy", [0, 4000000000], ids=["all-wave", "all-lane"])
def test_gather_ties_at_the_kth_distance(oracle_cls, heavy, medium):
    """Every photon stored three times: the k-th distance is shared by photons of which only some may be taken
    (k = 100 is not a multiple of 3).  The copies are identical, so whichever the reference's unstable sort keeps,
    the sum is the same; both kernels must take exactly k."""
    import raytracer_amd as rt
    sc = scenes.cornell()
    orc = oracle_cls(sc)
    ph = np.repeat(orc.photon_generate(3000, geometry=0, rng=1, seed=5), 3, axis=0)
    nr, po = orc.software_gbuffer(48, 36)
    pos, nrm, _ = orc.gbuffer_select(nr, po)
    orc.photon_set(ph)
    exp = orc.radiance_estimate(pos, nrm, 100, 1.5)
    g = rt.RtHip(sc, semantics="array", photon_radius=1.5, tuning=dict(gather_heavy=heavy, gather_medium=medium))
    g.photon_set(ph)
    got = g.radiance_estimate(pos, nrm, 100, 1.5)
    ok = ~np.isnan(exp)
    np.testing.assert_array_equal(np.isnan(exp), np.isnan(got))
    assert ok.mean() > 0.5
    np.testing.assert_allclose(got[ok], exp[ok], rtol=2e-5, atol=1e-9)
    g.close()


def test_gather_without_photons_is_nan(oracle_cls):
    import raytracer_amd as rt
    sc = scenes.cornell()
    g = rt.RtHip(sc, semantics="array")
    g.photon_set(np.zeros((0, 9), np.float32))
    got = g.radiance_estimate(np.zeros((5, 3), np.float32), np.ones((5, 3), np.float32), 100, 1.0)
    assert np.isnan(got).all()
    g.close()


def test_photon_mapped_frame(oracle_cls):
    """shade() with photon_map != NULL (rt_raytracer.cpp:155-157): light = radiance_estimate.
    Bar from north_star: PSNR >= 45 dB under a fixed seed."""
    import raytracer_amd as rt
    sc = scenes.cornell()
    orc = oracle_cls(sc)
    per_light = 20000
    ph = orc.photon_generate(per_light, geometry=0, rng=1, seed=42)
    # the reference's energy scale (5*color/count) makes radiance small; brighten the light so the
    # 8-bit image is not all zeros
    w, h = 80, 60
    nr, po = orc.software_gbuffer(w, h)
    mid = nr[:, 3].astype(np.int32) - 1
    nrm = (nr[:, :3].astype(np.float64) * 2.0 - 1.0).astype(np.float32)
    ph = ph.copy()
    ph[:, 3:6] *= 4000.0
    orc.photon_set(ph)
    exp, _, _ = orc.render_texels(po, nrm, mid, bounces=2, use_photon_map=1, photon_estimate=100, photon_radius=1.0)
    g = rt.RtHip(sc, semantics="array", bounces=2, use_photon_map=1, photon_estimate=100, photon_radius=1.0)
    g.photon_set(ph)
    got = g.render_frame(po, nr, w, h)
    assert exp[:, :3].max() > 30
    mse = np.mean((got[:, :3].astype(np.float64) - exp[:, :3].astype(np.float64)) ** 2)
    psnr = 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)
    assert psnr >= 45.0, psnr
    assert np.abs(got.astype(np.int32) - exp.astype(np.int32)).max() <= 1
    g.close()


def test_c5_pipeline_single_rank(oracle_cls):
    """C5's sequence on one GPU: emit shard -> (all-gather, a no-op at world 1) -> photon_set ->
    photon-mapped frame, through raytracer_amd.photons.build_photon_map.  The photon set must be the
    oracle's counter-RNG set bit for bit, the frame within the photon-image bar."""
    import raytracer_amd as rt
    from raytracer_amd.photons import build_photon_map
    sc = scenes.atrium(2048)
    orc = oracle_cls(sc)
    per_light = 4000
    g = rt.RtHip(sc, semantics="array", bounces=1, use_photon_map=1, photon_estimate=50, photon_radius=2.0, seed=42)
    n = build_photon_map(g, per_light, len(sc.lights), "cuda")
    ph = orc.photon_generate(per_light, geometry=0, rng=1, seed=42)
    assert n == ph.shape[0] > 1000
    w, h = 64, 48
    nr, po = orc.software_gbuffer(w, h)
    mid = nr[:, 3].astype(np.int32) - 1
    nrm = (nr[:, :3].astype(np.float64) * 2.0 - 1.0).astype(np.float32)
    # radiance at the G-buffer points from the GPU-built map == the oracle's from its own photons
    orc.photon_set(ph)
    cov = mid >= 0
    a = orc.radiance_estimate(po[cov], nrm[cov], 50, 2.0)
    b = g.radiance_estimate(po[cov], nrm[cov], 50, 2.0)
    ok = np.isfinite(a).all(1)
    np.testing.assert_array_equal(np.isfinite(b).all(1), ok)
    np.testing.assert_allclose(b[ok], a[ok], rtol=1e-5, atol=1e-9)
    g.close()


def test_lookup_vs_reference_vectors(oracle_cls):
    """PhotonMap::lookup (rt_photon_map.cpp:126-142) against the compiled reference's outputs in
    tests/golden/photons_cornell.npz (`lookup_r1`) and against the oracle at another radius.  The GPU sums the
    energies in grid order, the reference in photon order: rtol 2e-5 (the same bar as radiance_estimate); points
    with no photon in range are exactly zero on both sides."""
    import raytracer_amd as rt
    from test_oracle_golden import load, scene_for
    gold = load("photons_cornell")
    sc = scene_for("cornell", gold)
    g = rt.RtHip(sc, semantics="array", photon_radius=1.0)
    zero = g.photon_lookup(gold["pos"], 1.0)             # no photon map yet: the empty sum
    assert not zero.any()
    g.photon_set(gold["photons"])
    got = g.photon_lookup(gold["pos"], 1.0)
    exp = gold["lookup_r1"]
    assert (exp.sum(1) > 0).mean() > 0.3
    np.testing.assert_array_equal(got == 0, exp == 0)
    np.testing.assert_allclose(got, exp, rtol=2e-5, atol=1e-9)
    orc = oracle_cls(sc)
    orc.photon_set(gold["photons"])
    for radius in (0.35, 2.5):                           # below and above the grid cell the photons were binned with
        np.testing.assert_allclose(g.photon_lookup(gold["pos"], radius), orc.photon_lookup(gold["pos"], radius), rtol=2e-5, atol=1e-9)
    # the device-pointer entry
    torch = _torch()
    d_pos = torch.from_numpy(np.ascontiguousarray(gold["pos"], np.float32)).cuda()
    d_out = torch.zeros_like(d_pos)
    g.photon_lookup_device(d_pos, d_pos.shape[0], 1.0, d_out)
    np.testing.assert_array_equal(d_out.cpu().numpy(), got)
    g.close()
