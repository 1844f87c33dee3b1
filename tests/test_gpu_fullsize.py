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


def test_c3_shared_descent_changes_no_pixel(rt, torch, c3):
    """tuning.shadow_prefix: the descent the 64 shadow rays of an 8x8 tile share is made once per (tile, light) and the
    rays start below it (k_shadow_prefix).  It may only ever add visits, so the frame must be the one every ray
    renders from the root (shadow_prefix = 0) bit for bit, whatever the limits of the shared walk -- also with the
    reference's literal shadows (no t_max on the shadow rays)."""
    g = c3["g"]
    default = g.tuning()
    assert default["shadow_prefix"] > 0 and default["shadow_prefix_steps"] > 0
    try:
        for push, steps in ((0, 24), (1, 1), (3, 200), (12, 5), (30, 1024)):
            g.set_tuning(shadow_prefix=push, shadow_prefix_steps=steps)
            out = torch.zeros_like(c3["frame"])
            g.render_frame_device(c3["d_p"], c3["d_n"], W, H, out)
            assert torch.equal(out, c3["frame"]) and g.last_frame_rays() == c3["rays"], (push, steps)
    finally:
        g.set_tuning(shadow_prefix=default["shadow_prefix"], shadow_prefix_steps=default["shadow_prefix_steps"])
    lit = rt.RtHip(c3["sc"], semantics="array", accel=rt.ACCEL_PLOC, bounces=2, do_shadows=1, shadow_t_max=0.0)
    imgs = []
    for push in (0, 12):
        lit.set_tuning(shadow_prefix=push)
        out = torch.zeros_like(c3["frame"])
        lit.render_frame_device(c3["d_p"], c3["d_n"], W, H, out)
        imgs.append(out)
    lit.close()
    assert torch.equal(imgs[0], imgs[1])
    assert not torch.equal(imgs[0], c3["frame"])      # (literal shadows are darker: the two modes differ)


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


@pytest.fixture(scope="module")
def c4(rt, torch):
    """C4: atrium at ~1 M triangles with refractive objects, its 4K G-buffer and the frame it renders."""
    sc = scenes.atrium(1048576)
    g = rt.RtHip(sc, semantics="array", accel=rt.ACCEL_PLOC, bounces=2, do_shadows=1, shadow_t_max=1.0)
    w, h = 3840, 2160
    d_n = torch.zeros(w * h * 4, dtype=torch.float32, device="cuda")
    d_p = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda")
    g.render_gbuffer_device(sc.fov, w, h, d_n, d_p)
    frame = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
    g.render_frame_device(d_p, d_n, w, h, frame)
    rays = g.last_frame_rays()
    yield dict(sc=sc, g=g, d_n=d_n, d_p=d_p, frame=frame, rays=rays, w=w, h=h)
    g.close()


def test_c4_bvh_equals_exhaustive_search_1m(rt, torch, c4):
    """C4's geometry (1 M triangles): BVH == brute force on 16 k camera rays, and the 4K frame repeats itself."""
    sc, g, w, h = c4["sc"], c4["g"], c4["w"], c4["h"]
    b = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
    g.render_frame_device(c4["d_p"], c4["d_n"], w, h, b)
    assert torch.equal(c4["frame"], b) and g.last_frame_rays() == c4["rays"] and c4["rays"] > 40_000_000
    rng = np.random.RandomState(2)
    n = 1 << 14
    o = np.zeros((n, 3), np.float32)
    d = np.stack([rng.uniform(-1, 1, n), rng.uniform(-0.6, 0.6, n), -np.ones(n)], 1).astype(np.float32)
    rays6 = np.concatenate([o, d], 1)
    x = g.intersect(rays6)
    brute = rt.RtHip(sc, semantics="array", accel=1)
    y = brute.intersect(rays6)
    brute.close()
    np.testing.assert_array_equal(x["hit"], y["hit"])
    hit = x["hit"] > 0
    assert hit.mean() > 0.5
    np.testing.assert_array_equal(x["hits"]["prim"][hit], y["hits"]["prim"][hit])
    assert_bits_equal(x["hits"]["t"][hit], y["hits"]["t"][hit], "t")


def test_c4_oracle_on_a_sample_of_texels_half_of_them_glass(rt, torch, c4, oracle_cls):
    """C4 is the "reflections + refractions" configuration: the oracle's shade() (exhaustive search over the 1 M
    triangles, rt_raytracer.cpp:75-160 incl. the refraction branch :142-152 and quirk Q9) at 400 texels of the 4K
    frame, half of them on the refractive material (alpha 0.6, ior 1.5), against the GPU's pixels there: max-abs 1
    per channel.  Parity of the image is against the restatement (the reference's shade() is dead code: unpinned)."""
    sc = c4["sc"]
    orc = oracle_cls(sc)
    glass = [i for i, m in enumerate(sc.materials) if m.alpha > 0.1]
    assert glass
    nr = c4["d_n"].view(-1, 4).cpu().numpy()
    po = c4["d_p"].view(-1, 3).cpu().numpy()
    mat = nr[:, 3].astype(np.int32) - 1
    on_glass = np.nonzero(np.isin(mat, glass))[0]
    other = np.nonzero((mat >= 0) & ~np.isin(mat, glass))[0]
    assert len(on_glass) > 1000, "the 4K G-buffer shows too few refractive texels (%d)" % len(on_glass)
    rs = np.random.RandomState(21)
    pick = np.concatenate([rs.choice(on_glass, 200, replace=False), rs.choice(other, 200, replace=False)])
    nrm = (nr[pick, :3].astype(np.float64) * 2.0 - 1.0).astype(np.float32)
    exp, _, _ = orc.render_texels(po[pick], nrm, mat[pick], bounces=2, do_shadows=1, shadow_t_max=1.0)
    got = c4["frame"].view(-1, 4).cpu().numpy()[pick]
    diff = np.abs(got.astype(np.int32) - exp.astype(np.int32))
    assert diff.max() <= 1, diff.max()
    # the sample exercises the branch: some of the glass texels are lit (most face away from the lights: black)
    assert len(np.unique(got[:200, :3], axis=0)) >= 5


def test_c4_eight_way_tile_split_at_4k(rt, torch, c4):
    """north_star's C4 partition: the 4K frame over 8 ranks by interleaved 64x64 tiles, each (virtual) rank shading its
    tiles straight into its packed chunk; the unpacked chunks are the whole frame bit for bit, the rays add up."""
    from raytracer_amd.tiles import max_owned_tiles
    g, w, h, nranks = c4["g"], c4["w"], c4["h"], 8
    chunk = max_owned_tiles(w, h, 64, nranks) * 64 * 64 * 4
    recv = torch.zeros(nranks * chunk, dtype=torch.uint8, device="cuda")
    rays = 0
    for r in range(nranks):
        g.render_tiles_packed_device([c4["d_p"]], [c4["d_n"]], w, h, 64, r, nranks, [recv[r * chunk:(r + 1) * chunk]])
        rays += g.last_frame_rays()
    out = torch.full((w * h * 4,), 9, dtype=torch.uint8, device="cuda")
    g.unpack_chunks_device(recv, chunk, w, h, 64, nranks, out)
    assert torch.equal(out, c4["frame"])
    assert rays == c4["rays"]


def test_c5_gather_kernels_agree_at_10m_photons(rt, torch, c3, oracle_cls):
    """C5's photon map (10 M emitted): the per-lane and the per-wave gather give the same radiance at 30 k points
    of the frame (2e-5 relative: the order of the float sum differs), NaN exactly where no photon is in range, and
    the sharded emission concatenates to the same photon set.  And the ORACLE itself (radiance_estimate restated from
    rt_photon_map.cpp:149-208: every photon scanned, sorted, k nearest, Gaussian filter) at 240 of those points, against
    the DEFAULT tuning -- the split, the coarse pass below gather_medium and the fine-grid shell walk above it as they
    run in the bench -- and against the shell walk forced for every point."""
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
    default = g.tuning()
    for name, heavy, medium in (("lane", 4000000000, default["gather_medium"]), ("wave", 0, default["gather_medium"]),
                                ("default", default["gather_heavy"], default["gather_medium"]), ("shell", 0, 0)):
        g.set_tuning(gather_heavy=heavy, gather_medium=medium)
        out = torch.zeros_like(pos)
        g.radiance_estimate_device(pos, nrm, pos.shape[0], 100, 1.0, out)
        outs[name] = out.cpu().numpy()
    g.close()
    nan = np.isnan(outs["lane"])
    assert 0.02 < nan.mean() < 0.98
    for name in ("wave", "default", "shell"):
        np.testing.assert_array_equal(np.isnan(outs[name]), nan)
        np.testing.assert_allclose(outs[name][~nan], outs["lane"][~nan], rtol=2e-5, atol=1e-12)
    # the oracle on 240 of the points: 120 at random, 120 at random among those with photons in range (the heavy ones
    # are among them: a tenth of the frame's points has more than 4096 photons in its cube)
    rs = np.random.RandomState(4)
    lit = np.nonzero(~nan.any(axis=1))[0]
    pick = np.unique(np.concatenate([rs.choice(pos.shape[0], 120, replace=False), rs.choice(lit, 120, replace=False)]))
    orc = oracle_cls(sc)
    orc.photon_set(whole[:stored].cpu().numpy())
    exp = orc.radiance_estimate(pos[pick].cpu().numpy(), nrm[pick].cpu().numpy(), 100, 1.0)
    for name in ("default", "shell"):
        got = outs[name][pick]
        np.testing.assert_array_equal(np.isnan(got), np.isnan(exp))
        ok = ~np.isnan(exp)
        np.testing.assert_allclose(got[ok], exp[ok], rtol=2e-5, atol=1e-9)
