"""Textures: Texture::lookup (data/data.cpp:35-57) in shade()
(rt_raytracer.cpp:93-95) and texture2D(GL_NEAREST, repeat) in the live shading (mesh_fs.cpp:29-30).

Parity unpinned: FreeImage and GL are absent here, so neither lookup can be run from the reference;
the oracle restates them from the source text and the CPU test below pins the restatement to
hand-computed texels.  The GPU tests compare the HIP path with the oracle through the C ABI."""
import copy

import numpy as np
import pytest

from conftest import assert_bits_equal
from raytracer_amd import scenes


def _texture(w, h, seed):
    rng = np.random.RandomState(seed)
    t = rng.randint(0, 256, (h, w, 4)).astype(np.uint8)
    t[..., 3] = 255
    return t


def textured_cornell():
    """The Cornell room with textures on two diffuse walls, the mirror and the glass material, and texture
    coordinates pushed outside [0, 1] (negative, > 1) on every mesh so the wrap rules are exercised."""
    sc = copy.deepcopy(scenes.cornell())
    for k, (mid, size) in enumerate([(0, (16, 8)), (1, (5, 7)), (3, (32, 32)), (4, (3, 2))]):
        sc.materials[mid].texture = _texture(size[0], size[1], 100 + k)
    for k, me in enumerate(sc.meshes):
        me.texcoords = (me.texcoords * np.float32(1.0 + 0.9 * k) - np.float32(0.65 * k)).astype(np.float32)
    return sc


# ---- CPU: the oracle's lookup against hand-computed texels ---------------------------------------
def test_oracle_texture_lookup_cases(oracle_cls):
    """One texel at a time through shade(): a single light straight above the point (lambert = 1),
    kd = 1, bounces = 1 => colour = Texture::lookup(u, v) exactly."""
    tex = _texture(8, 4, 1)
    sc = scenes.Scene(name="one-texel", materials=[scenes.Material(diffuse=(9.0, 9.0, 9.0), texture=tex)], meshes=[],
                      lights=[scenes.Light(position=(0.0, 5.0, -3.0), color=(255.0, 255.0, 255.0), energy=1.0)])
    orc = oracle_cls(sc)
    cases = [  # (u, v) -> (x, y)
        ((0.0, 0.0), (0, 0)), ((0.25, 0.5), (2, 2)), ((0.999, 0.999), (7, 3)),
        ((1.0, 1.0), (7, 3)),          # x == width in the reference (FreeImage fails): clamped here
        ((-0.25, -0.25), (6, 3)),      # u < 0: u += 1
        ((2.5, 1.5), (4, 2)),          # |u| > 1: fmod
        ((-1.75, -3.5), (2, 2)),       # fmod keeps the sign, then + 1
        ((-1.0, 1.0), (0, 3)),         # |u| == 1 is not wrapped: -1 + 1 = 0
    ]
    uv = np.array([c[0] for c in cases], np.float32)
    n = len(cases)
    pos = np.tile(np.array([[0.0, 0.0, -3.0]], np.float32), (n, 1))
    nrm = np.tile(np.array([[0.0, 1.0, 0.0]], np.float32), (n, 1))
    rgba, _, _ = orc.render_texels(pos, nrm, np.zeros(n, np.int32), bounces=1, uv=uv)
    for k, (_, (x, y)) in enumerate(cases):
        assert tuple(rgba[k, :3]) == tuple(tex[y, x, :3]), (cases[k], rgba[k], tex[y, x])
    # untextured call of the same scene object: (0, 0) is the record's default texcoord
    rgba0, _, _ = orc.render_texels(pos, nrm, np.zeros(n, np.int32), bounces=1)
    assert (rgba0[:, :3] == tex[0, 0, :3]).all()


# ---- GPU -------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def rt():
    import raytracer_amd
    return raytracer_amd


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.mark.gpu
def test_gbuffer_uv(rt, oracle_cls):
    torch = _torch()
    sc = textured_cornell()
    orc = oracle_cls(sc)
    w, h = 100, 75
    nr, po, uv = orc.software_gbuffer(w, h, with_uv=True)
    g = rt.RtHip(sc, semantics="array")
    d_n = torch.zeros(w * h * 4, dtype=torch.float32, device="cuda")
    d_p = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda")
    d_uv = torch.full((w * h * 2,), 9.0, dtype=torch.float32, device="cuda")
    g.render_gbuffer_uv_device(sc.fov, w, h, d_n, d_p, d_uv)
    assert_bits_equal(nr, d_n.cpu().numpy().reshape(-1, 4), "normal+alpha")
    assert_bits_equal(po, d_p.cpu().numpy().reshape(-1, 3), "position")
    assert_bits_equal(uv, d_uv.cpu().numpy().reshape(-1, 2), "texcoord")
    assert uv.min() < -0.2 and uv.max() > 1.2
    g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shadows,bounces,stmax", [(0, 3, 0.0), (1, 3, 1.0)], ids=["noshadow-b3", "shadowfix-b3"])
@pytest.mark.parametrize("path", ["async", "sync"])
def test_textured_frame(rt, oracle_cls, shadows, bounces, stmax, path):
    """shade() with textured materials at every level of the tree (level 0 from the G-buffer's texcoord
    plane, deeper levels from the hit's interpolated texcoord), both frame drivers.  Tolerance as the
    untextured frame: max-abs 1 per channel."""
    torch = _torch()
    sc = textured_cornell()
    orc = oracle_cls(sc)
    w, h = 100, 75
    nr, po, uv = orc.software_gbuffer(w, h, with_uv=True)
    mid = nr[:, 3].astype(np.int32) - 1
    nrm = (nr[:, :3].astype(np.float64) * 2.0 - 1.0).astype(np.float32)
    kw = dict(bounces=bounces, do_shadows=shadows, shadow_t_max=stmax)
    exp, _, exp_rays = orc.render_texels(po, nrm, mid, uv=uv, **kw)
    exp0, _, _ = orc.render_texels(po, nrm, mid, **kw)             # level-0 texcoords (0, 0)
    plain, _, _ = oracle_cls(scenes.cornell()).render_texels(po, nrm, mid, **kw)
    assert (exp != plain).any(1).mean() > 0.3 and (exp != exp0).any(1).mean() > 0.1   # the textures matter
    g = rt.RtHip(sc, semantics="array", tuning=dict(frame_sync=1 if path == "sync" else 0), **kw)
    d_uv = torch.from_numpy(uv).cuda()
    g.set_frame_texcoords_device([d_uv])
    got = g.render_frame(po, nr, w, h)
    diff = np.abs(got.astype(np.int32) - exp.astype(np.int32))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3, (diff.max(), (diff > 0).mean())
    assert g.last_frame_rays() == exp_rays
    g.set_frame_texcoords_device([])
    got0 = g.render_frame(po, nr, w, h)
    diff = np.abs(got0.astype(np.int32) - exp0.astype(np.int32))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3
    # removing the textures brings the untextured picture back
    for i, m in enumerate(sc.materials):
        if m.texture is not None:
            g.set_texture(i, None)
    g.update()
    got_plain = g.render_frame(po, nr, w, h)
    diff = np.abs(got_plain.astype(np.int32) - plain.astype(np.int32))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3
    g.close()


@pytest.mark.gpu
def test_textured_tiles_batch(rt, oracle_cls):
    """The split / batched level 0 reads the texcoord plane of its own frame at its own pixels."""
    torch = _torch()
    from raytracer_amd.tiles import owned_tiles
    sc = textured_cornell()
    orc = oracle_cls(sc)
    w, h, tile, nranks = 100, 75, 16, 3
    g = rt.RtHip(sc, semantics="array", bounces=2)
    gb, full = [], []
    for f in range(2):
        nr, po, uv = orc.software_gbuffer(w, h, fov=sc.fov - 0.05 * f, with_uv=True)
        t = [torch.from_numpy(a).cuda() for a in (po, nr, uv)]
        gb.append(t)
        g.set_frame_texcoords_device([t[2]])
        img = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
        g.render_frame_device(t[0], t[1], w, h, img)
        full.append(img.cpu().numpy().reshape(h, w, 4))
    assert (full[0] != full[1]).any()
    g.set_frame_texcoords_device([t[2] for t in gb])
    tiles_x = (w + tile - 1) // tile
    for r in range(nranks):
        outs = [torch.full((w * h * 4,), 7, dtype=torch.uint8, device="cuda") for _ in range(2)]
        g.render_tiles_batch_device([t[0] for t in gb], [t[1] for t in gb], w, h, tile, r, nranks, outs)
        for f in range(2):
            img = outs[f].cpu().numpy().reshape(h, w, 4)
            for tid in owned_tiles(w, h, tile, r, nranks):
                ty, tx = divmod(int(tid), tiles_x)
                sl = (slice(ty * tile, min((ty + 1) * tile, h)), slice(tx * tile, min((tx + 1) * tile, w)))
                assert (img[sl] == full[f][sl]).all(), (r, f, tid)
    g.close()


@pytest.mark.gpu
def test_textured_live_shading(rt, oracle_cls):
    """mesh_fs albedo = texture2D(GL_NEAREST, repeat): shade_results on processRays' records, and the
    live frame (raypass at the hit's texcoord, composite at the G-buffer's texcoord plane)."""
    torch = _torch()
    from test_gpu_parity import _oracle_live_frame
    sc = textured_cornell()
    orc = oracle_cls(sc)
    g = rt.RtHip(sc, accel=2)
    w, h = 96, 64
    nr, po, uv = orc.software_gbuffer(w, h, with_uv=True)
    sel_pos, sel_nrm, _ = orc.gbuffer_select(nr, po)
    out = g.process_rays(sel_pos, sel_nrm)
    rec = out["records"]
    off = np.zeros(len(out["offset_end"]), rt.OFFSET_DTYPE)
    off["material"], off["end"] = out["offset_material"], out["offset_end"]
    mid = np.repeat(off["material"].astype(np.int32), np.diff(np.concatenate([[0], off["end"]])))
    want = orc.shade_records(rec, mid)
    plain = oracle_cls(scenes.cornell()).shade_records(rec, mid)
    assert (np.abs(want - plain).max(1) > 0.02).mean() > 0.2
    rgb, _ = g.shade_results(rec, off)
    np.testing.assert_allclose(rgb, want, rtol=2e-5, atol=1e-6)

    # live frame: the oracle composition of test_gpu_parity with the composite's records carrying uv
    d_p, d_n, d_uv = torch.from_numpy(po).cuda(), torch.from_numpy(nr).cuda(), torch.from_numpy(uv).cuda()
    g.set_frame_texcoords_device([d_uv])
    d_rp = torch.full((w * h, 3), 255, dtype=torch.uint8, device="cuda")
    d_out = torch.zeros((w * h, 4), dtype=torch.uint8, device="cuda")
    rp = np.full((w * h, 3), 255, np.uint8)
    exp, nrays, cov = _oracle_live_frame(orc, sc, nr, po, 1, 0, rp, uv=uv)
    rays = g.live_frame_device(d_p, d_n, w, h, 1, 0, d_rp, d_out)
    assert rays == nrays > 0
    got, got_rp = d_out.cpu().numpy(), d_rp.cpu().numpy()
    d_rp_diff = np.abs(got_rp.astype(int) - rp.astype(int)).max(1)
    d_img = np.abs(got[cov].astype(int) - exp[cov].astype(int)).max(1)
    assert (d_rp_diff > 1).mean() <= 1e-3 and (d_img > 1).mean() <= 1e-3, ((d_rp_diff > 1).mean(), (d_img > 1).mean())
    g.close()


@pytest.mark.gpu
def test_texture_api_errors(rt):
    torch = _torch()
    sc = textured_cornell()
    g = rt.RtHip(sc, semantics="array")
    tex = _texture(4, 4, 9)
    with pytest.raises(rt.RtHipError):
        g.set_texture(len(sc.materials), tex)                     # material out of range
    with pytest.raises(rt.RtHipError):
        g.set_texture(-1, tex)
    with pytest.raises(ValueError):
        g.set_texture(0, tex[..., :3])                            # not RGBA
    uv = torch.zeros(8, dtype=torch.float32, device="cuda")
    with pytest.raises(rt.RtHipError):
        g.set_frame_texcoords_device([uv] * 17)                   # more planes than a batch holds
    g.set_frame_texcoords_device([uv] * 16)
    g.set_frame_texcoords_device([])
    g.close()
