"""rt_hip_update's change detection and BVH refit (north_star (3): "refit"; what the reference does instead is
rebuild its triangle soup on every processRays call, rt_raytracer.cpp:457-503).  The bar: whatever rt_hip_update
decides to do -- nothing, tables only, refit on the kept topology, rebuild -- the results equal those of a context
built from scratch on the same scene, bit for bit (ids, t, u, v, hit records, frames)."""
import numpy as np
import pytest

from conftest import assert_bits_equal
from raytracer_amd import scenes

pytestmark = pytest.mark.gpu
MISS = 0xFFFFFFFF


@pytest.fixture(scope="module")
def rt():
    import raytracer_amd
    return raytracer_amd


def _matrix(rng, scale=1.0):
    ang = rng.uniform(-0.3, 0.3) * scale
    c, s = np.cos(ang), np.sin(ang)
    sc = 1.0 + rng.uniform(-0.2, 0.2) * scale
    m = np.eye(4)
    m[:3, :3] = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]]) * sc
    m[:3, 3] = rng.uniform(-0.5, 0.5, 3) * scale
    return m.astype(np.float32)


def _with_matrices(sc, mats):
    meshes = [scenes.Mesh(m.verts, m.normals, m.texcoords, m.faces, m.material, mats[i], m.position) for i, m in enumerate(sc.meshes)]
    return scenes.Scene(sc.name + "-anim", sc.materials, meshes, sc.lights, sc.fov)


def _samples(oracle_cls, sc, w=96, h=72):
    orc = oracle_cls(sc)
    nr, po = orc.software_gbuffer(w, h)
    pos, nrm, _ = orc.gbuffer_select(nr, po)
    return pos, nrm


def _same_results(a, b, what):
    np.testing.assert_array_equal(a["tidx"], b["tidx"], err_msg=what)
    for k in ("t", "u", "v", "dir"):
        assert_bits_equal(a[k], b[k], "%s: %s" % (what, k))
    assert a["records"].tobytes() == b["records"].tobytes(), "%s: hit records differ" % what
    np.testing.assert_array_equal(a["offset_material"], b["offset_material"])
    np.testing.assert_array_equal(a["offset_end"], b["offset_end"])


@pytest.mark.parametrize("accel", [2, 3, 4, 1], ids=["lbvh", "ploc", "sah", "brute"])
def test_animated_matrices_refit_equals_fresh_build(rt, oracle_cls, accel):
    """Ten frames of moving model matrices: every frame the long-lived context (refit) and a context built from scratch
    return the same processRays output bit for bit; the first frame also equals the oracle."""
    base = scenes.atrium(2048)
    rng = np.random.RandomState(7)
    pos, nrm = _samples(oracle_cls, scenes.with_transforms(base))
    mats = [_matrix(rng) for _ in base.meshes]
    g = rt.RtHip(_with_matrices(base, mats), accel=accel)
    assert g.stats()["last_update"] == rt.UPDATE_REBUILD
    kinds = []
    for frame in range(10):
        if frame:
            for i in range(len(mats)):           # every mesh drifts a little
                mats[i] = (mats[i].astype(np.float64) @ _matrix(rng, 0.1).astype(np.float64)).astype(np.float32)
                g.set_model_mat(i, np.ascontiguousarray(mats[i].T).reshape(16))   # column-major
            g.update()
            kinds.append(g.stats()["last_update"])
        sc_now = _with_matrices(base, mats)
        fresh = rt.RtHip(sc_now, accel=accel)
        a, b = g.process_rays(pos, nrm), fresh.process_rays(pos, nrm)
        fresh.close()
        assert (a["tidx"] != MISS).sum() > 500
        _same_results(a, b, "frame %d" % frame)
        if frame in (0, 9):
            o = oracle_cls(sc_now).soup_closest(pos, a["dir"])
            np.testing.assert_array_equal(o["tidx"], a["tidx"])
            assert_bits_equal(o["t"], a["t"], "t vs exhaustive oracle, frame %d" % frame)
    assert kinds == [rt.UPDATE_REFIT] * 9, kinds
    g.close()


def test_update_does_the_least_that_is_needed(rt, oracle_cls):
    sc = scenes.with_transforms(scenes.atrium(2048))
    pos, nrm = _samples(oracle_cls, sc)
    g = rt.RtHip(sc, accel=3)
    ref = g.process_rays(pos, nrm)
    kind = lambda: g.stats()["last_update"]
    # the reference's processRays re-reads every model matrix each call: the same matrices change nothing
    g.set_model_mats(0, np.stack([m.model_colmajor() for m in sc.meshes]))     # all meshes in one call
    g.update()
    assert kind() == rt.UPDATE_NONE
    with pytest.raises(rt.RtHipError, match="mesh range"):
        g.set_model_mats(len(sc.meshes) - 1, np.zeros((2, 16), np.float32))
    _same_results(g.process_rays(pos, nrm), ref, "unchanged matrices")
    # materials / lights: tables only
    g.load_scene_tables(sc)
    g.update()
    assert kind() == rt.UPDATE_TABLES
    _same_results(g.process_rays(pos, nrm), ref, "tables")
    # one matrix changed: refit
    m0 = sc.meshes[0].model_colmajor().copy()
    m0[12] += 0.25
    g.set_model_mat(0, m0)
    with pytest.raises(rt.RtHipError, match="rt_hip_update"):
        g.process_rays(pos, nrm)       # tracing a stale scene is refused
    g.update()
    assert kind() == rt.UPDATE_REFIT
    moved = g.process_rays(pos, nrm)
    assert moved["records"].tobytes() != ref["records"].tobytes()
    # and back: refit again, the original results return
    g.set_model_mat(0, sc.meshes[0].model_colmajor())
    g.update()
    assert kind() == rt.UPDATE_REFIT
    _same_results(g.process_rays(pos, nrm), ref, "moved back")
    # forced
    g.invalidate()
    g.update()
    assert kind() == rt.UPDATE_REBUILD
    _same_results(g.process_rays(pos, nrm), ref, "rebuilt")
    # the meshes scattered: the kept topology (which pairs triangles of different meshes near the top) no longer fits
    g.set_tuning(refit_max_growth=1e9)      # never give up on the refit ...
    rng = np.random.RandomState(1)
    mats = []
    for i, m in enumerate(sc.meshes):
        mm = m.model.copy()
        mm[:3, 3] += rng.uniform(-40, 40, 3).astype(np.float32)
        mats.append(mm)
        g.set_model_mat(i, np.ascontiguousarray(mm.T).reshape(16))
    g.update()
    st = g.stats()
    assert st["last_update"] == rt.UPDATE_REFIT
    growth = st["bvh_area_cost"] / st["bvh_area_cost_built"]
    assert growth > 1.2, growth             # ... although its area sum says the tree has degraded
    fresh = rt.RtHip(_with_matrices(sc, mats), accel=3)
    exp_scattered = fresh.process_rays(pos, nrm)
    fresh.close()
    _same_results(g.process_rays(pos, nrm), exp_scattered, "scattered, refitted")
    g.set_tuning(refit_max_growth=0.5 * (1.0 + growth))   # now the quality check applies: the next refit is rejected
    nudged = mats[0].copy()
    nudged[0, 3] += 1e-3
    g.set_model_mat(0, np.ascontiguousarray(nudged.T).reshape(16))
    g.update()
    st = g.stats()
    assert st["last_update"] == rt.UPDATE_REBUILD and st["bvh_area_cost"] == st["bvh_area_cost_built"]
    g.set_model_mat(0, np.ascontiguousarray(mats[0].T).reshape(16))
    g.update()
    assert kind() == rt.UPDATE_REFIT
    _same_results(g.process_rays(pos, nrm), exp_scattered, "after the quality rebuild")
    for i, m in enumerate(sc.meshes):
        g.set_model_mat(i, m.model_colmajor())
    g.update()
    # refit switched off: always rebuild
    _same_results(g.process_rays(pos, nrm), ref, "matrices restored")
    g.set_tuning(refit_max_growth=0.0)
    g.set_model_mat(0, m0)
    g.update()
    assert kind() == rt.UPDATE_REBUILD
    _same_results(g.process_rays(pos, nrm), moved, "refit off")
    g.close()


def test_array_semantics_ignore_model_matrices(rt, oracle_cls):
    """AccelArray::addMesh flattens object space (rt_accel_array.cpp:17-32): a model matrix is no change at all."""
    sc = scenes.atrium(2048)
    g = rt.RtHip(sc, semantics="array")
    m = sc.meshes[0].model_colmajor().copy()
    m[12] += 3.0
    g.set_model_mat(0, m)
    g.update()
    assert g.stats()["last_update"] == rt.UPDATE_NONE
    g.close()


def test_refit_frames_equal_fresh_frames(rt, oracle_cls):
    """The shaded frame (shadow + reflection + refraction rays through the refitted tree) and the photon emission."""
    import torch
    base = scenes.atrium(2048)
    rng = np.random.RandomState(3)
    mats = [_matrix(rng, 0.3) for _ in base.meshes]
    g = rt.RtHip(_with_matrices(base, mats), accel=3, bounces=2, do_shadows=1, shadow_t_max=1.0)
    w, h = 96, 72

    def frame(ctx, fov):
        d_n = torch.zeros(w * h * 4, dtype=torch.float32, device="cuda")
        d_p = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda")
        ctx.render_gbuffer_device(fov, w, h, d_n, d_p)
        out = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
        ctx.render_frame_device(d_p, d_n, w, h, out)
        return d_n.cpu().numpy(), d_p.cpu().numpy(), out.cpu().numpy(), ctx.last_frame_rays()

    for step in range(3):
        for i in range(len(mats)):
            mats[i] = (mats[i].astype(np.float64) @ _matrix(rng, 0.15).astype(np.float64)).astype(np.float32)
            g.set_model_mat(i, np.ascontiguousarray(mats[i].T).reshape(16))
        g.update()
        assert g.stats()["last_update"] == rt.UPDATE_REFIT
        sc_now = _with_matrices(base, mats)
        fresh = rt.RtHip(sc_now, accel=3, bounces=2, do_shadows=1, shadow_t_max=1.0)
        a, b = frame(g, base.fov), frame(fresh, base.fov)
        fresh.close()
        for x, y, name in zip(a, b, ("gbuffer normals", "gbuffer positions", "image", "rays")):
            assert np.array_equal(x, y), "step %d: %s differ between refit and fresh build" % (step, name)
        assert a[3] > 10000 and a[2].reshape(-1, 4)[:, :3].max() > 30
    g.close()


def test_too_deep_a_tree_falls_back_to_brute_force(rt, oracle_cls, capfd):
    """A BVH deeper than the traversal stack takes is not used: the scene is served by brute force, with the same results
    (the limit is 95; the test lowers it through the tuning)."""
    sc = scenes.atrium(2048)
    orc = oracle_cls(sc)
    rng = np.random.RandomState(0)
    d = orc.camera_rays(sc.fov, 64, 48)
    rays = np.concatenate([rng.uniform(-1, 1, (d.shape[0], 3)).astype(np.float32), d], 1).astype(np.float32)
    exp = orc.array_intersect(rays)
    g = rt.RtHip(sc, semantics="array", accel=3, tuning=dict(max_bvh_depth=5))
    st = g.stats()
    assert st["bvh_depth"] == 0 and st["num_tris"] == sc.num_tris     # built, measured too deep, not used
    assert "brute force" in capfd.readouterr().err
    got = g.intersect(rays)
    np.testing.assert_array_equal(exp["hit"], got["hit"])
    hit = exp["hit"] > 0
    np.testing.assert_array_equal(exp["index"][hit], got["hits"]["prim"][hit].astype(np.int64))
    assert_bits_equal(exp["t"][hit], got["hits"]["t"][hit], "t")
    g.set_tuning(max_bvh_depth=95)      # takes effect at the next update: rebuild, now with the tree
    g.update()
    assert g.stats()["bvh_depth"] > 5
    got = g.intersect(rays)
    np.testing.assert_array_equal(exp["index"][hit], got["hits"]["prim"][hit].astype(np.int64))
    g.close()


@pytest.mark.parametrize("accel", [1, 3], ids=["brute", "ploc"])
def test_hit_at_exactly_t_max_is_a_miss(rt, oracle_cls, accel):
    """rt_accel_array.cpp:84,116: min_t starts at 1,000,000 and a hit needs t < min_t, so a triangle at exactly
    t == 1e6 is missed (and so is one beyond); one ulp nearer is hit."""
    near = np.nextafter(np.float32(1.0e6), np.float32(0))
    meshes = []
    for z in (1.0e6, 2.0e6):
        v = np.array([[-4, -4, -z], [4, -4, -z], [0, 4, -z]], np.float32)
        n = np.tile(np.array([[0, 0, 1]], np.float32), (3, 1))
        meshes.append(scenes.Mesh(v, n, np.zeros((3, 2), np.float32), np.array([[0, 1, 2]], np.uint32), 0, np.eye(4, dtype=np.float32), (0, 0, 0)))
    base = scenes.cornell()
    # pad with far-away filler so that the BVH builders have something to build
    filler = scenes.atrium(2048)
    for m in filler.meshes:
        meshes.append(scenes.Mesh(m.verts + np.array([5000, 0, 0], np.float32), m.normals, m.texcoords, m.faces, 0, m.model, m.position))
    sc = scenes.Scene("tmax", base.materials[:1], meshes, base.lights, base.fov)
    rays = np.array([[0, 0, 0, 0, 0, -1],                      # t == 1e6 exactly
                     [0, 0, float(1.0e6 - near), 0, 0, -1],    # origin moved back by one ulp of 1e6: t > 1e6
                     [0, 0, -1.0, 0, 0, -1],                   # t = 999,999
                     [0, 0, 0, 0, 0, -2]], np.float32)          # direction scaled: t = 5e5 (t is a parameter, Q13)
    orc = oracle_cls(sc)
    exp = orc.array_intersect(rays)
    assert list(exp["hit"]) == [0, 0, 1, 1], exp["hit"]
    g = rt.RtHip(sc, semantics="array", accel=accel)
    got = g.intersect(rays)
    np.testing.assert_array_equal(exp["hit"], got["hit"])
    hit = exp["hit"] > 0
    np.testing.assert_array_equal(exp["index"][hit], got["hits"]["prim"][hit].astype(np.int64))
    assert_bits_equal(exp["t"][hit], got["hits"]["t"][hit], "t")
    g.close()
