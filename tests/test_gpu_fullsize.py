"""BASELINE.json's full sizes (C3 atrium-262k at 1080p, C4 atrium-1M, C5 10 M photons) through properties
that do not need the oracle to finish the whole job: BVH == exhaustive search, any-hit == closest-hit
below t_max, the two frame drivers agree, the tile split reassembles the whole frame, the two gather
kernels agree -- plus the oracle itself on a random sample of the full-size frame's texels."""
import numpy as np
import pytest

from conftest import assert_bits_equal
from raytracer_amd import scenes

pytestmark = pytest.mark.gpu

W, H = 1920, 1080


@pytest.fixture(scope="module")
def rt():
    import raytracer_amd
    return raytracer_amd


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.fixture(scope="module")
def c3(rt, torch):
    """atrium-262k, the bench workload's context, its 1080p G-buffer and the frame it renders."""
    sc = scenes.atrium(262144)
    g = rt.RtHip(sc, semantics="array", accel=rt.ACCEL_PLOC, bounces=2, do_shadows=1, shadow_t_max=1.0)
    d_n = torch.zeros(W * H * 4, dtype=torch.float32, device="cuda")
    d_p = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda")
    g.render_gbuffer_device(sc.fov, W, H, d_n, d_p)
    frame = torch.zeros(W * H * 4, dtype=torch.uint8, device="cuda")
    g.render_frame_device(d_p, d_n, W, H, frame)
    rays = g.last_frame_rays()
    yield dict(sc=sc, g=g, d_n=d_n, d_p=d_p, frame=frame, rays=rays)
    g.close()


def _sample_rays(torch, c3, n, seed):
    """Reflection rays of n random covered texels (processRays' ray generation, rt_raytracer.cpp:445-448)."""
    nr = c3["d_n"].view(-1, 4)
    cov = torch.nonzero(nr[:, 3] > 0).flatten()
    gen = torch.Generator(device="cpu").manual_seed(seed)
    pick = cov[torch.randperm(cov.numel(), generator=gen)[:n].cuda()]
    P = c3["d_p"].view(-1, 3)[pick]
    N = (nr[pick, :3].double() * 2.0 - 1.0).float()
    D = P - N * (2.0 * (P * N).sum(1, keepdim=True))
    D = D / D.norm(dim=1, keepdim=True)
    return P.contiguous(), D.contiguous()


def test_c3_bvh_equals_exhaustive_search(rt, torch, c3):
    """65 k reflection rays of the 1080p frame: the PLOC BVH and the brute-force kernel (pinned against the
    oracle at small sizes) return the same primitive, t, u, v bit for bit over 262 k triangles."""
    org, dirs = _sample_rays(torch, c3, 1 << 16, 3)
    rays = torch.cat([org, dirs], 1).cpu().numpy()
    a = c3["g"].intersect(rays)
    brute = rt.RtHip(c3["sc"], semantics="array", accel=1)
    b = brute.intersect(rays)
    brute.close()
    np.testing.assert_array_equal(a["hit"], b["hit"])
    hit = a["hit"] > 0
    assert hit.mean() > 0.8
    np.testing.assert_array_equal(a["hits"]["prim"][hit], b["hits"]["prim"][hit])
    for k in ("t", "u", "v"):
        assert_bits_equal(a["hits"][k][hit], b["hits"][k][hit], k)


def test_c3_any_hit_is_closest_hit_below_tmax(rt, torch, c3):
    """Both queries accept a triangle by the same tests, so occluded(ray) <=> the closest hit exists (t < t_max);
    2 M shadow-style rays (texel -> light 0, un-normalised direction, so t_max = 1e6 is far beyond the light)."""
    n = 1 << 21
    org, _ = _sample_rays(torch, c3, n, 5)
    n = org.shape[0]
    light = torch.tensor(c3["sc"].lights[0].position, dtype=torch.float32, device="cuda")
    dirs = (light[None, :] - org).contiguous()
    g = c3["g"]
    hits = torch.zeros(n * 4, dtype=torch.float32, device="cuda")
    occ = torch.zeros(n, dtype=torch.uint8, device="cuda")
    g.trace_closest_device(org, dirs, n, hits)
    g.trace_any_device(org, dirs, n, occ)
    prim = hits.view(torch.int32).view(-1, 4)[:, 0]
    assert torch.equal(occ > 0, prim != -1)
    assert float((occ > 0).float().mean()) > 0.05


def test_c3_frame_drivers_agree_and_repeat(rt, torch, c3):
    """The sync-free driver (levels sized by bounds) and the host-sized one render the same 1080p image, ray for
    ray; rendering again changes nothing."""
    g = c3["g"]
    again = torch.zeros_like(c3["frame"])
    g.render_frame_device(c3["d_p"], c3["d_n"], W, H, again)
    assert torch.equal(again, c3["frame"]) and g.last_frame_rays() == c3["rays"]
    g.set_tuning(frame_sync=1)
    sync = torch.zeros_like(c3["frame"])
    g.render_frame_device(c3["d_p"], c3["d_n"], W, H, sync)
    g.set_tuning(frame_sync=0)
    assert torch.equal(sync, c3["frame"]) and g.last_frame_rays() == c3["rays"]
    assert c3["rays"] > 10_000_000
    assert int(c3["frame"].view(-1, 4)[:, :3].max()) > 60


@pytest.mark.parametrize("nranks", [2, 8])
def test_c3_tile_split_reassembles_the_frame(rt, torch, c3, nranks):
    """Every virtual rank renders its 64x64 tiles of the 1080p frame straight into its packed chunk; unpacking the
    chunks (what rank 0 receives after the all-to-all) gives the whole frame bit for bit; the ranks' rays add up."""
    from raytracer_amd.tiles import max_owned_tiles
    g = c3["g"]
    chunk = max_owned_tiles(W, H, 64, nranks) * 64 * 64 * 4
    recv = torch.zeros(nranks * chunk, dtype=torch.uint8, device="cuda")
    rays = 0
    for r in range(nranks):
        g.render_tiles_packed_device([c3["d_p"]], [c3["d_n"]], W, H, 64, r, nranks, [recv[r * chunk:(r + 1) * chunk]])
        rays += g.last_frame_rays()
    out = torch.full((W * H * 4,), 9, dtype=torch.uint8, device="cuda")
    g.unpack_chunks_device(recv, chunk, W, H, 64, nranks, out)
    assert torch.equal(out, c3["frame"])
    assert rays == c3["rays"]


def test_c3_oracle_on_a_sample_of_texels(rt, torch, c3, oracle_cls):
    """The oracle's shade() (exhaustive AccelArray search) at 200 random texels of the full-size frame against the
    GPU's pixels there: max-abs 1 per channel (north_star's tolerance; in practice equal)."""
    orc = oracle_cls(c3["sc"])
    nr = c3["d_n"].view(-1, 4).cpu().numpy()
    po = c3["d_p"].view(-1, 3).cpu().numpy()
    cov = np.nonzero(nr[:, 3] > 0)[0]
    pick = np.random.RandomState(11).choice(cov, 200, replace=False)
    mid = nr[pick, 3].astype(np.int32) - 1
    nrm = (nr[pick, :3].astype(np.float64) * 2.0 - 1.0).astype(np.float32)
    exp, _, _ = orc.render_texels(po[pick], nrm, mid, bounces=2, do_shadows=1, shadow_t_max=1.0)
    got = c3["frame"].view(-1, 4).cpu().numpy()[pick]
    diff = np.abs(got.astype(np.int32) - exp.astype(np.int32))
    assert diff.max() <= 1, diff.max()
    assert (diff > 0).mean() < 0.01


def test_c4_bvh_equals_exhaustive_search_1m(rt, torch):
    """C4's geometry (1 M triangles): BVH == brute force on 16 k camera rays, and the 4K frame repeats itself."""
    sc = scenes.atrium(1048576)
    g = rt.RtHip(sc, semantics="array", accel=rt.ACCEL_PLOC, bounces=2, do_shadows=1, shadow_t_max=1.0)
    w, h = 3840, 2160
    d_n = torch.zeros(w * h * 4, dtype=torch.float32, device="cuda")
    d_p = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda")
    g.render_gbuffer_device(sc.fov, w, h, d_n, d_p)
    a = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
    b = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
    g.render_frame_device(d_p, d_n, w, h, a)
    rays = g.last_frame_rays()
    g.render_frame_device(d_p, d_n, w, h, b)
    assert torch.equal(a, b) and g.last_frame_rays() == rays and rays > 40_000_000
    rng = np.random.RandomState(2)
    n = 1 << 14
    o = np.zeros((n, 3), np.float32)
    d = np.stack([rng.uniform(-1, 1, n), rng.uniform(-0.6, 0.6, n), -np.ones(n)], 1).astype(np.float32)
    rays6 = np.concatenate([o, d], 1)
    x = g.intersect(rays6)
    g.close()
    brute = rt.RtHip(sc, semantics="array", accel=1)
    y = brute.intersect(rays6)
    brute.close()
    np.testing.assert_array_equal(x["hit"], y["hit"])
    hit = x["hit"] > 0
    assert hit.mean() > 0.5
    np.testing.assert_array_equal(x["hits"]["prim"][hit], y["hits"]["prim"][hit])
    assert_bits_equal(x["hits"]["t"][hit], y["hits"]["t"][hit], "t")


def test_c5_gather_kernels_agree_at_10m_photons(rt, torch, c3):
    """C5's photon map (10 M emitted): the per-lane and the per-wave gather give the same radiance at 30 k points
    of the frame (2e-5 relative: the order of the float sum differs), NaN exactly where no photon is in range, and
    the sharded emission concatenates to the same photon set."""
    from raytracer_amd.photons import build_photon_map
    from raytracer_amd.tiles import photon_shard
    sc = c3["sc"]
    g = rt.RtHip(sc, semantics="array", accel=rt.ACCEL_PLOC, use_photon_map=1, photon_radius=1.0, seed=42)
    per_light = 2_500_000
    total = per_light * len(sc.lights)
    whole = torch.empty((total, 9), dtype=torch.float32, device="cuda")
    stored = g.photon_emit_device(per_light, 0, total, whole, total)
    assert 4_000_000 < stored < total
    parts = []
    for r in range(4):
        first, count = photon_shard(total, r, 4)
        buf = torch.empty((max(count, 1), 9), dtype=torch.float32, device="cuda")
        k = g.photon_emit_device(per_light, first, count, buf, max(count, 1))
        parts.append(buf[:k])
    assert torch.equal(torch.cat(parts, 0), whole[:stored])
    g.photon_set_device(whole[:stored].contiguous(), stored)
    pos, _ = _sample_rays(torch, c3, 30_000, 9)
    nrm = torch.zeros_like(pos); nrm[:, 1] = 1.0
    outs = {}
    for name, heavy in (("lane", 4000000000), ("wave", 0), ("split", 1024)):
        g.set_tuning(gather_heavy=heavy)
        out = torch.zeros_like(pos)
        g.radiance_estimate_device(pos, nrm, pos.shape[0], 100, 1.0, out)
        outs[name] = out.cpu().numpy()
    g.close()
    nan = np.isnan(outs["lane"])
    assert 0.02 < nan.mean() < 0.98
    for name in ("wave", "split"):
        np.testing.assert_array_equal(np.isnan(outs[name]), nan)
        np.testing.assert_allclose(outs[name][~nan], outs["lane"][~nan], rtol=2e-5, atol=1e-12)
