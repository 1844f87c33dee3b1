"""Scene ingest: a minimal Wavefront OBJ + MTL reader that yields the
reference's data model, so real assets can be dropped in where the procedural stand-ins are now.

The reference loads through Assimp (data/loader.cpp:87-172), which is not available here; this
reader reproduces the conventions loader.cpp applies on top of Assimp:
  * colours x 255 (`_convertColor`, :25-33): Kd -> diffuse, Ks -> specular
  * alpha = 1 - opacity (:103-104): `d` (or 1 - `Tr`)
  * reflectivity = reflective colour's red channel (:100-101): `Kr r g b` (the usual MTL extension;
    plain MTL has no reflective colour)
  * shininess = Ns (:102), ior = Ni (:105)
  * polygons are triangulated (aiProcess_Triangulate, :156) as fans; identical (v, vt, vn) corners
    share one vertex (aiProcess_JoinIdenticalVertices, :157)
  * one Mesh per (object/group, material) run, model matrix = identity, position = 0 (OBJ has no
    node transforms)
Differences, on purpose: a face without normals gets its geometric normal (Assimp leaves mNormals
null there and loader.cpp:62 would dereference it); a mesh without `vt` gets texcoord (0, 0) like
loader.cpp:58-65; textures (`map_Kd`) are recorded by name only (Texture::lookup is 8f row 3); OBJ
carries no lights or camera, so they are arguments.
`save_obj` writes a Scene back (used by the tests for a round trip and handy for exporting the
procedural scenes to other tools).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence

import numpy as np

from .scenes import F32, Light, Material, Mesh, Scene


def _parse_mtl(path: str) -> Dict[str, dict]:
    mats: Dict[str, dict] = {}
    cur = None
    with open(path) as f:
        for raw in f:
            t = raw.split("#", 1)[0].split()
            if not t:
                continue
            key, val = t[0], t[1:]
            if key == "newmtl":
                cur = mats.setdefault(" ".join(val), dict(Kd=(0.6, 0.6, 0.6), Ks=(0.0, 0.0, 0.0), Kr=(0.0, 0.0, 0.0), Ns=0.0,
                                                          d=1.0, Ni=1.0, map_Kd=None))
            elif cur is None:
                continue
            elif key in ("Kd", "Ks", "Kr"):
                v = [float(x) for x in val[:3]]
                cur[key] = tuple(v + [v[-1]] * (3 - len(v)))
            elif key in ("Ns", "Ni", "d"):
                cur[key] = float(val[0])
            elif key == "Tr":
                cur["d"] = 1.0 - float(val[0])
            elif key == "map_Kd":
                cur["map_Kd"] = val[-1]
    return mats


def _material(m: dict) -> Material:
    return Material(diffuse=tuple(255.0 * c for c in m["Kd"]), specular=tuple(255.0 * c for c in m["Ks"]),
                    shininess=float(m["Ns"]), reflectivity=float(m["Kr"][0]), alpha=1.0 - float(m["d"]), ior=float(m["Ni"]))


def load_obj(path: str, lights: Optional[Sequence[Light]] = None, fov: float = 0.86, name: Optional[str] = None) -> Scene:
    pos: List[List[float]] = []
    nrm: List[List[float]] = []
    uv: List[List[float]] = []
    mtl: Dict[str, dict] = {}
    mat_order: List[str] = []
    runs = []                      # (group name, material name, list of polygons; polygon = list of (v, vt, vn))
    group, mat = "default", None
    cur_polys = None

    def start_run():
        nonlocal cur_polys
        cur_polys = []
        runs.append((group, mat, cur_polys))

    base = os.path.dirname(os.path.abspath(path))
    with open(path) as f:
        for raw in f:
            t = raw.split("#", 1)[0].split()
            if not t:
                continue
            key, val = t[0], t[1:]
            if key == "v":
                pos.append([float(x) for x in val[:3]])
            elif key == "vn":
                nrm.append([float(x) for x in val[:3]])
            elif key == "vt":
                uv.append([float(val[0]), float(val[1]) if len(val) > 1 else 0.0])
            elif key == "mtllib":
                lib = _parse_mtl(os.path.join(base, " ".join(val)))
                mtl.update(lib)
                for k in lib:                                   # material ids follow the library's order
                    if k not in mat_order:
                        mat_order.append(k)
            elif key in ("o", "g"):
                group = " ".join(val) or "default"
                cur_polys = None
            elif key == "usemtl":
                mat = " ".join(val)
                if mat not in mat_order:
                    mat_order.append(mat)
                cur_polys = None
            elif key == "f":
                if cur_polys is None:
                    start_run()
                corners = []
                for c in val:
                    p = (c.split("/") + ["", ""])[:3]

                    def idx(s, n):
                        if not s:
                            return -1
                        i = int(s)
                        return i - 1 if i > 0 else n + i
                    corners.append((idx(p[0], len(pos)), idx(p[1], len(uv)), idx(p[2], len(nrm))))
                if len(corners) >= 3:
                    cur_polys.append(corners)

    if None in [m for _, m, _ in runs] and "__default__" not in mat_order:
        mat_order.append("__default__")
    materials = [_material(mtl.get(m, dict(Kd=(0.6, 0.6, 0.6), Ks=(0, 0, 0), Kr=(0, 0, 0), Ns=0.0, d=1.0, Ni=1.0)))
                 for m in mat_order]
    P = np.asarray(pos, dtype=np.float64).reshape(-1, 3)
    N = np.asarray(nrm, dtype=np.float64).reshape(-1, 3)
    T = np.asarray(uv, dtype=np.float64).reshape(-1, 2)
    meshes: List[Mesh] = []
    for _, m, polys in runs:
        if not polys:
            continue
        verts, normals, uvs, faces = [], [], [], []
        seen: Dict[tuple, int] = {}
        for poly in polys:
            face_n = None
            if any(c[2] < 0 for c in poly):                     # no normal given: the polygon's geometric normal
                a, b, c = (P[poly[k][0]] for k in (0, 1, 2))
                face_n = np.cross(b - a, c - a)
                ln = np.linalg.norm(face_n)
                face_n = face_n / ln if ln > 0 else np.array([0.0, 0.0, 1.0])
            ids = []
            for k, (vi, ti, ni) in enumerate(poly):
                key = (vi, ti, ni) if ni >= 0 else (vi, ti, ("flat", len(faces), k))
                j = seen.get(key)
                if j is None:
                    j = seen[key] = len(verts)
                    verts.append(P[vi])
                    normals.append(N[ni] if ni >= 0 else face_n)
                    uvs.append(T[ti] if ti >= 0 else (0.0, 0.0))
                ids.append(j)
            for k in range(1, len(ids) - 1):                    # fan
                faces.append((ids[0], ids[k], ids[k + 1]))
        meshes.append(Mesh(verts=np.asarray(verts, dtype=F32).reshape(-1, 3), normals=np.asarray(normals, dtype=F32).reshape(-1, 3),
                           texcoords=np.asarray(uvs, dtype=F32).reshape(-1, 2), faces=np.asarray(faces, dtype=np.uint32).reshape(-1, 3),
                           material=mat_order.index(m if m is not None else "__default__")))
    if not meshes:
        raise ValueError("%s holds no faces" % path)
    if lights is None:
        allv = np.concatenate([m.verts for m in meshes], 0)
        lo, hi = allv.min(0), allv.max(0)
        lights = [Light(position=(float((lo[0] + hi[0]) / 2), float(hi[1] - 0.05 * (hi[1] - lo[1])), float((lo[2] + hi[2]) / 2)))]
    return Scene(name=name or os.path.splitext(os.path.basename(path))[0], materials=materials, meshes=meshes,
                 lights=list(lights), fov=fov)


def save_obj(scene: Scene, path: str) -> None:
    """Write `scene` as OBJ + MTL (world space: each mesh's model matrix is applied)."""
    mtl_path = os.path.splitext(path)[0] + ".mtl"
    with open(mtl_path, "w") as f:
        for i, m in enumerate(scene.materials):
            f.write("newmtl m%d\nKd %r %r %r\nKs %r %r %r\nKr %r %r %r\nNs %r\nd %r\nNi %r\n\n" % (
                i, *[float(F32(c) / F32(255)) for c in m.diffuse], *[float(F32(c) / F32(255)) for c in m.specular],
                float(m.reflectivity), float(m.reflectivity), float(m.reflectivity), float(m.shininess),
                float(1.0 - m.alpha), float(m.ior)))
    with open(path, "w") as f:
        f.write("mtllib %s\n" % os.path.basename(mtl_path))
        base = 1
        for k, mesh in enumerate(scene.meshes):
            M = np.asarray(mesh.model, dtype=np.float64)
            v = mesh.verts.astype(np.float64) @ M[:3, :3].T + M[:3, 3]
            nm = mesh.normals.astype(np.float64) @ np.linalg.inv(M[:3, :3])
            f.write("o mesh%d\nusemtl m%d\n" % (k, mesh.material))
            for p in v:
                f.write("v %r %r %r\n" % (float(F32(p[0])), float(F32(p[1])), float(F32(p[2]))))
            for t in mesh.texcoords:
                f.write("vt %r %r\n" % (float(t[0]), float(t[1])))
            for n in nm:
                f.write("vn %r %r %r\n" % (float(F32(n[0])), float(F32(n[1])), float(F32(n[2]))))
            for a, b, c in mesh.faces.astype(np.int64) + base:
                f.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % (a, a, a, b, b, b, c, c, c))
            base += mesh.verts.shape[0]
