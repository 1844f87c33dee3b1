"""Scene ingest on the GPU path: a scene written as OBJ + MTL and read back with
data/loader.cpp's conventions (raytracer_amd/ingest.py) goes through the HIP path and matches the oracle on the
ingested scene -- processRays bit for bit, the shaded frame to 1/255 -- and the original scene's image."""
import numpy as np
import pytest

from conftest import assert_bits_equal
from raytracer_amd import scenes
from raytracer_amd.ingest import load_obj, save_obj

pytestmark = pytest.mark.gpu
MISS = 0xFFFFFFFF


@pytest.mark.parametrize("scene_fn,accel", [(scenes.cornell, 1), (lambda: scenes.atrium(2048), 3)], ids=["cornell-brute", "atrium2k-ploc"])
def test_ingested_scene_matches_oracle(tmp_path, oracle_cls, scene_fn, accel):
    import raytracer_amd as rt
    src = scene_fn()
    path = str(tmp_path / "scene.obj")
    save_obj(src, path)
    sc = load_obj(path, lights=src.lights, fov=src.fov)
    assert sc.num_tris == src.num_tris and len(sc.materials) >= 1
    orc = oracle_cls(sc)
    w, h = 96, 72
    nr, po = orc.software_gbuffer(w, h)
    pos, nrm, _ = orc.gbuffer_select(nr, po)
    g = rt.RtHip(sc, accel=accel)
    got = g.process_rays(pos, nrm)
    exp = orc.soup_closest(pos, got["dir"])
    assert (exp["tidx"] != MISS).sum() > 1000
    np.testing.assert_array_equal(exp["tidx"], got["tidx"])
    for k in ("t", "u", "v"):
        assert_bits_equal(exp[k], got[k], k)
    ref = orc.process_rays(pos, nrm)
    assert len(ref["records"]) == len(got["records"])
    g.close()

    kw = dict(bounces=2, do_shadows=1, shadow_t_max=1.0)
    g = rt.RtHip(sc, semantics="array", accel=accel, **kw)
    img = g.render_frame(po, nr, w, h)
    mid = nr[:, 3].astype(np.int32) - 1
    n3 = (nr[:, :3].astype(np.float64) * 2.0 - 1.0).astype(np.float32)
    exp_img, _, exp_rays = orc.render_texels(po, n3, mid, **kw)
    assert np.abs(img.astype(np.int32) - exp_img.astype(np.int32)).max() <= 1
    assert g.last_frame_rays() == exp_rays and img[:, :3].max() > 50
    g.close()
