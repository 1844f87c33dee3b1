"""Helpers shared by the parity tests: the tie class north_star allows for primitive-id mismatches."""
import json
import os

import numpy as np


def world_triangles(sc, use_model_mat=True):
    """(T, 3, 3) float64 world-space vertices in global triangle id order (mesh order, then face order:
    rt_raytracer.cpp:465-485)."""
    out = []
    for m in sc.meshes:
        v = np.asarray(m.verts, np.float64)
        if use_model_mat:
            M = np.asarray(m.model, np.float64).reshape(4, 4)
            v = v @ M[:3, :3].T + M[:3, 3]
        out.append(v[np.asarray(m.faces, np.int64)])
    return np.concatenate(out, 0)


def assert_only_edge_ties(sc, ids_a, ids_b, t_a, t_b, what, use_model_mat=True, min_match=0.9999):
    """north_star's bar for the primitive id: >= 99.99 % of the hit rays carry the same id, OR every mismatch is
    an exact tie -- bit-equal t -- between two triangles that share an edge (two vertices) or are coplanar
    duplicates sharing all three (DACRT keeps the first in its traversal order, rt_raytracer.cpp:358, the HIP
    path and rt_accel_array.cpp:116 the lowest id).  Returns the observed match rate."""
    ids_a, ids_b = np.asarray(ids_a, np.int64), np.asarray(ids_b, np.int64)
    diff = np.nonzero(ids_a != ids_b)[0]
    rate = 1.0 - len(diff) / max(len(ids_a), 1)
    if len(diff):
        ta = np.ascontiguousarray(t_a, np.float32).view(np.uint32)[diff]
        tb = np.ascontiguousarray(t_b, np.float32).view(np.uint32)[diff]
        assert (ta == tb).all(), "%s: an id mismatch whose t is not bit-equal" % what
        tri = world_triangles(sc, use_model_mat)
        ext = float(np.ptp(tri.reshape(-1, 3), axis=0).max())
        a, b = tri[ids_a[diff]], tri[ids_b[diff]]              # (k, 3, 3) each
        d = np.abs(a[:, :, None, :] - b[:, None, :, :]).max(-1)  # (k, 3, 3) vertex-to-vertex distances
        shared = (d < 1e-6 * ext).any(2).sum(1)
        assert (shared >= 2).all(), "%s: %d id mismatches are not between triangles sharing an edge" % (what, int((shared < 2).sum()))
    _record(what, rate, len(diff), len(ids_a))
    return rate


def _record(what, rate, mismatches, rays):
    """Observed match rates, kept where the GPU run's artefacts go (gpurun_out/ is merged back)."""
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if not os.path.isdir(d):
        return
    path = os.path.join(d, "parity_id_match.json")
    try:
        data = json.load(open(path)) if os.path.exists(path) else {}
    except Exception:
        data = {}
    data[what] = {"match_rate": rate, "mismatches": int(mismatches), "hit_rays": int(rays)}
    json.dump(data, open(path, "w"), indent=1, sort_keys=True)
