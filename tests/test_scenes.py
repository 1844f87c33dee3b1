"""Procedural stand-in scenes: sizes, winding (backface culling is the reference's only
self-intersection guard), determinism."""
import numpy as np

from raytracer_amd import scenes


def _facing(sc):
    bad = degenerate = 0
    for m in sc.meshes:
        v = m.verts.astype(np.float64)
        f = m.faces
        g = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
        n = m.normals[f[:, 0]] + m.normals[f[:, 1]] + m.normals[f[:, 2]]
        d = (g * n).sum(1)
        bad += int((d < 0).sum())
        degenerate += int((d == 0).sum())
    return bad, degenerate


def test_cornell_is_30_triangles():
    sc = scenes.cornell()
    assert sc.num_tris == 30 and len(sc.lights) == 1
    assert _facing(sc) == (0, 0)


def test_sphere_field():
    sc = scenes.sphere_field()
    assert len(sc.meshes) == 64
    z = np.array([m.position[2] for m in sc.meshes])
    assert z.min() > -31 and z.max() < -5


def test_atrium_targets_and_winding():
    for target in (2048, 8192, 65536):
        sc = scenes.atrium(target)
        assert abs(sc.num_tris - target) <= 0.06 * target, sc.num_tris
        assert len(sc.materials) == 25 and len(sc.lights) == 4
        bad, deg = _facing(sc)
        assert bad == 0
        assert deg <= max(0.02 * sc.num_tris, 160)   # the 8 ornaments' pole triangles only
    a, b = scenes.atrium(2048), scenes.atrium(2048)
    assert all(np.array_equal(x.verts, y.verts) for x, y in zip(a.meshes, b.meshes))
