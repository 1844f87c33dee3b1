"""Procedural scenes for the ray-tracing hot path (numpy only, deterministic).

The reference ships no assets and real Sponza / San Miguel cannot be fetched offline, so every
configuration in BASELINE.json is served by a seeded procedural stand-in:

  C1  sphere_field()      64 unit spheres (one per Mesh::position, rt_accel_spheres.cpp:27-31)
  C2  cornell()           30 triangles: 5 walls + 2 boxes x 5 faces
  C3  atrium(262144)      colonnade / arches / drapes, ~262k triangles, 25 materials, 4 lights
  C4  atrium(1048576)     same generator at ~1M triangles + refractive objects

Conventions follow the reference's data model (data/data.h:18-69): colours are 0..255 floats,
`alpha` is 1 - opacity, the eye sits at the origin looking down -Z with no view matrix
(rasterizer/shaders/prepass_vs.cpp:12-13), and every surface a ray is meant to see is wound
counter-clockwise as seen from the ray's side, because backface culling is the reference's only
self-intersection guard (rt_raytracer.cpp:336).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

F32 = np.float32


@dataclass
class Material:
    diffuse: tuple
    specular: tuple = (255.0, 255.0, 255.0)
    shininess: float = 32.0
    reflectivity: float = 0.0
    alpha: float = 0.0
    ior: float = 1.0
    texture: object = None     # (h, w, 4) u8 RGBA, row 0 = bottom scanline (Texture, data/data.h:7-16), or None


@dataclass
class Mesh:
    verts: np.ndarray          # (nv, 3) f32
    normals: np.ndarray        # (nv, 3) f32
    texcoords: np.ndarray      # (nv, 2) f32
    faces: np.ndarray          # (nf, 3) u32
    material: int
    model: np.ndarray = field(default_factory=lambda: np.eye(4, dtype=F32))  # row-major 4x4
    position: tuple = (0.0, 0.0, 0.0)

    def model_colmajor(self) -> np.ndarray:
        return np.ascontiguousarray(self.model.T.reshape(-1), dtype=F32)


@dataclass
class Light:
    position: tuple
    color: tuple = (255.0, 255.0, 255.0)
    energy: float = 1.0


@dataclass
class Scene:
    name: str
    materials: List[Material]
    meshes: List[Mesh]
    lights: List[Light]
    fov: float = 0.86          # data/scene.cpp:10 (radians, stored as tan(fov) by Camera::setFOV)

    @property
    def num_tris(self) -> int:
        return int(sum(m.faces.shape[0] for m in self.meshes))


# --------------------------------------------------------------------------------------------
# mesh helpers
# --------------------------------------------------------------------------------------------
def _mesh(verts, normals, uvs, faces, material, model=None, position=(0.0, 0.0, 0.0)) -> Mesh:
    return Mesh(
        verts=np.ascontiguousarray(verts, dtype=F32).reshape(-1, 3),
        normals=np.ascontiguousarray(normals, dtype=F32).reshape(-1, 3),
        texcoords=np.ascontiguousarray(uvs, dtype=F32).reshape(-1, 2),
        faces=np.ascontiguousarray(faces, dtype=np.uint32).reshape(-1, 3),
        material=material,
        model=np.eye(4, dtype=F32) if model is None else np.asarray(model, dtype=F32),
        position=tuple(float(x) for x in position),
    )


def quad(p0, p1, p2, p3, material, nu=1, nv=1) -> Mesh:
    """Tessellated quad p0->p1 (u) and p0->p3 (v); front face = counter-clockwise p0,p1,p2,p3."""
    p0, p1, p2, p3 = (np.asarray(p, dtype=np.float64) for p in (p0, p1, p2, p3))
    u = np.linspace(0.0, 1.0, nu + 1)
    v = np.linspace(0.0, 1.0, nv + 1)
    uu, vv = np.meshgrid(u, v, indexing="xy")        # (nv+1, nu+1)
    pts = ((1 - uu)[..., None] * (1 - vv)[..., None] * p0 + uu[..., None] * (1 - vv)[..., None] * p1
           + uu[..., None] * vv[..., None] * p2 + (1 - uu)[..., None] * vv[..., None] * p3)
    n = np.cross(p1 - p0, p3 - p0)
    n = n / np.linalg.norm(n)
    normals = np.broadcast_to(n, pts.shape)
    uvs = np.stack([uu, vv], axis=-1)
    return _mesh(pts, normals, uvs, _grid_faces(nu, nv), material)


def _grid_faces(nu, nv, flip=False) -> np.ndarray:
    """Two triangles per cell of an (nv+1) x (nu+1) vertex grid, CCW for +normal = du x dv."""
    j, i = np.meshgrid(np.arange(nv), np.arange(nu), indexing="ij")
    a = (j * (nu + 1) + i).reshape(-1)
    b = a + 1
    c = a + (nu + 1) + 1
    d = a + (nu + 1)
    if flip:
        f = np.stack([np.stack([a, c, b], -1), np.stack([a, d, c], -1)], axis=1)
    else:
        f = np.stack([np.stack([a, b, c], -1), np.stack([a, c, d], -1)], axis=1)
    return f.reshape(-1, 3).astype(np.uint32)


def param_surface(fn, nu, nv, material, flip=False) -> Mesh:
    """Surface fn(u, v) -> (pos, normal) on [0,1]^2, tessellated nu x nv cells."""
    u = np.linspace(0.0, 1.0, nu + 1)
    v = np.linspace(0.0, 1.0, nv + 1)
    uu, vv = np.meshgrid(u, v, indexing="xy")
    pos, nrm = fn(uu, vv)
    uvs = np.stack([uu, vv], axis=-1)
    return _mesh(pos, nrm, uvs, _grid_faces(nu, nv, flip), material)


def box(lo, hi, material, faces="xXyYzZ", n=1) -> List[Mesh]:
    """Axis-aligned box with outward-facing quads; `faces` selects which of the 6 to emit."""
    x0, y0, z0 = lo
    x1, y1, z1 = hi
    out = []
    if "x" in faces:
        out.append(quad((x0, y0, z0), (x0, y0, z1), (x0, y1, z1), (x0, y1, z0), material, n, n))
    if "X" in faces:
        out.append(quad((x1, y0, z1), (x1, y0, z0), (x1, y1, z0), (x1, y1, z1), material, n, n))
    if "y" in faces:
        out.append(quad((x0, y0, z0), (x1, y0, z0), (x1, y0, z1), (x0, y0, z1), material, n, n))
    if "Y" in faces:
        out.append(quad((x0, y1, z1), (x1, y1, z1), (x1, y1, z0), (x0, y1, z0), material, n, n))
    if "z" in faces:
        out.append(quad((x1, y0, z0), (x0, y0, z0), (x0, y1, z0), (x1, y1, z0), material, n, n))
    if "Z" in faces:
        out.append(quad((x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1), material, n, n))
    return out


def merge(meshes: List[Mesh], material: Optional[int] = None) -> Mesh:
    """Concatenate meshes (same material) into one."""
    vs, ns, ts, fs = [], [], [], []
    base = 0
    for m in meshes:
        vs.append(m.verts)
        ns.append(m.normals)
        ts.append(m.texcoords)
        fs.append(m.faces + np.uint32(base))
        base += m.verts.shape[0]
    return _mesh(np.concatenate(vs), np.concatenate(ns), np.concatenate(ts), np.concatenate(fs),
                 meshes[0].material if material is None else material)


# --------------------------------------------------------------------------------------------
# C1: sphere field
# --------------------------------------------------------------------------------------------
def sphere_field(n_side: int = 8, seed: int = 1) -> Scene:
    """n_side^2 unit spheres on a jittered grid, z in [-30, -6]; one Mesh per sphere whose only
    meaningful field is `position` (rt_accel_spheres.cpp:27-31).  Each mesh also carries one
    small triangle so triangle back ends accept the scene."""
    rng = np.random.RandomState(seed)
    meshes = []
    for j in range(n_side):
        for i in range(n_side):
            x = (i - (n_side - 1) / 2.0) * 3.0 + rng.uniform(-0.6, 0.6)
            y = rng.uniform(-2.5, 2.5)
            z = -6.0 - (j / max(n_side - 1, 1)) * 24.0 + rng.uniform(-0.6, 0.6)
            tri_v = np.array([[x - 0.1, y, z], [x + 0.1, y, z], [x, y + 0.1, z]], dtype=F32)
            tri_n = np.tile(np.array([[0, 0, 1]], dtype=F32), (3, 1))
            tri_t = np.zeros((3, 2), dtype=F32)
            meshes.append(_mesh(tri_v, tri_n, tri_t, np.array([[0, 1, 2]]), 0, position=(x, y, z)))
    mats = [Material((200.0, 200.0, 200.0), reflectivity=0.5)]
    lights = [Light((0.0, 20.0, 0.0), (255.0, 255.0, 255.0), 1.0)]
    return Scene("spheres-%d" % (n_side * n_side), mats, meshes, lights)


# --------------------------------------------------------------------------------------------
# C2: Cornell box, 30 triangles
# --------------------------------------------------------------------------------------------
def cornell() -> Scene:
    """5 walls (10 tris) + 2 boxes x 5 visible faces (20 tris) = 30 triangles; open towards the
    eye at the origin.  Room: x,y in [-5,5], z in [-16,-4]."""
    WHITE, RED, GREEN, MIRROR, GLASS, FLOOR = range(6)
    mats = [
        Material((230.0, 230.0, 230.0)),
        Material((220.0, 40.0, 40.0)),
        Material((40.0, 200.0, 40.0)),
        Material((240.0, 240.0, 250.0), reflectivity=0.8, shininess=96.0),
        Material((200.0, 220.0, 255.0), alpha=0.5, ior=1.5, shininess=64.0),
        Material((200.0, 200.0, 180.0), reflectivity=0.3),
    ]
    x0, x1, y0, y1, z0, z1 = -5.0, 5.0, -5.0, 5.0, -16.0, -4.0
    meshes = [
        quad((x0, y0, z1), (x1, y0, z1), (x1, y0, z0), (x0, y0, z0), FLOOR),    # floor, +Y
        quad((x0, y1, z0), (x1, y1, z0), (x1, y1, z1), (x0, y1, z1), WHITE),    # ceiling, -Y
        quad((x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0), WHITE),    # back, +Z
        quad((x0, y0, z1), (x0, y0, z0), (x0, y1, z0), (x0, y1, z1), RED),      # left, +X
        quad((x1, y0, z0), (x1, y0, z1), (x1, y1, z1), (x1, y1, z0), GREEN),    # right, -X
    ]
    meshes += box((-3.5, -5.0, -13.0), (-0.5, 1.0, -10.0), MIRROR, faces="xXYzZ")   # tall box
    meshes += box((0.8, -5.0, -9.5), (3.6, -2.2, -6.8), GLASS, faces="xXYzZ")       # short box
    lights = [Light((0.0, 4.5, -9.0), (255.0, 255.0, 255.0), 1.0)]
    scene = Scene("cornell-30", mats, meshes, lights)
    assert scene.num_tris == 30
    return scene


# --------------------------------------------------------------------------------------------
# C3 / C4: procedural atrium
# --------------------------------------------------------------------------------------------
def _cylinder(cx, cz, y0, y1, r, nu, nv, material) -> Mesh:
    def fn(u, v):
        a = 2 * np.pi * u
        pos = np.stack([cx + r * np.cos(a), y0 + (y1 - y0) * v, cz - r * np.sin(a)], -1)
        nrm = np.stack([np.cos(a), np.zeros_like(a), -np.sin(a)], -1)
        return pos, nrm
    return param_surface(fn, nu, nv, material)


def _uv_sphere(c, r, nu, nv, material) -> Mesh:
    def fn(u, v):
        a = 2 * np.pi * u
        b = np.pi * (v - 0.5)
        d = np.stack([np.cos(b) * np.cos(a), np.sin(b), -np.cos(b) * np.sin(a)], -1)
        return np.asarray(c) + r * d, d
    return param_surface(fn, nu, nv, material)


def _arch(z, x0, x1, y_base, rise, depth, nu, nv, material) -> List[Mesh]:
    """A vaulted rib spanning x0..x1 at depth z: underside (faces down) and front face strip."""
    def under(u, v):
        x = x0 + (x1 - x0) * u
        y = y_base + rise * np.sin(np.pi * u)
        zz = z + depth * (v - 0.5)
        pos = np.stack([x, y, zz], -1)
        dydx = rise * np.pi * np.cos(np.pi * u) / (x1 - x0)
        n = np.stack([dydx, -np.ones_like(u), np.zeros_like(u)], -1)
        n /= np.linalg.norm(n, axis=-1, keepdims=True)
        return pos, n

    def front(u, v):
        x = x0 + (x1 - x0) * u
        ylo = y_base + rise * np.sin(np.pi * u)
        yhi = y_base + rise + 1.0
        pos = np.stack([x, ylo + (yhi - ylo) * v, np.full_like(u, z + depth * 0.5)], -1)
        n = np.broadcast_to(np.array([0.0, 0.0, 1.0]), pos.shape)
        return pos, n

    return [param_surface(under, nu, nv, material), param_surface(front, nu, max(nv // 2, 1), material)]


def _drape(x, z0, z1, y_top, y_bot, amp, waves, nu, nv, material, facing=+1) -> Mesh:
    """A waving cloth sheet in the y-z plane at x, visible from the +x (facing=+1) or -x side."""
    z0, z1 = min(z0, z1), max(z0, z1)

    def fn(u, v):
        zz = z0 + (z1 - z0) * u
        yy = y_top + (y_bot - y_top) * v
        ph = 2 * np.pi * waves * u
        xx = x + amp * np.sin(ph) * (0.3 + 0.7 * v)
        pos = np.stack([xx, yy, zz], -1)
        dxdz = amp * np.cos(ph) * 2 * np.pi * waves / (z1 - z0) * (0.3 + 0.7 * v)
        dxdy = amp * np.sin(ph) * 0.7 / (y_bot - y_top)
        n = np.stack([np.ones_like(u), -dxdy, -dxdz], -1) * facing
        n /= np.linalg.norm(n, axis=-1, keepdims=True)
        return pos, n
    # du = +z, dv = -y  ->  du x dv = (+z) x (-y) = +x ; flip for the -x side
    return param_surface(fn, nu, nv, material, flip=(facing < 0))


def atrium(target_tris: int = 262144, seed: int = 3, refractive: bool = True) -> Scene:
    """Colonnaded hall in front of the eye: x in [-12,12], y in [-3,9], z in [-62,-2].
    Tessellation density is solved from `target_tris` (within about 1 %)."""
    rng = np.random.RandomState(seed)
    palette = [
        (214, 204, 184), (196, 180, 150), (160, 150, 140), (120, 96, 80), (180, 60, 50),
        (60, 110, 170), (70, 150, 90), (200, 170, 60), (150, 80, 160), (230, 230, 230),
        (90, 90, 100), (190, 120, 70), (110, 160, 160), (170, 190, 120), (210, 140, 150),
        (140, 140, 200), (100, 70, 60), (220, 200, 120), (80, 130, 110), (160, 100, 100),
    ]
    mats = [Material(tuple(float(c) for c in p), shininess=float(16 + 8 * (i % 5))) for i, p in enumerate(palette)]
    FLOOR = len(mats); mats.append(Material((200.0, 195.0, 185.0), reflectivity=0.35, shininess=80.0))
    MARBLE = len(mats); mats.append(Material((225.0, 225.0, 235.0), reflectivity=0.25, shininess=64.0))
    CHROME = len(mats); mats.append(Material((245.0, 245.0, 250.0), reflectivity=0.85, shininess=128.0))
    GLASS = len(mats); mats.append(Material((210.0, 230.0, 255.0), reflectivity=0.1, alpha=0.6 if refractive else 0.0, ior=1.5, shininess=96.0))
    WATER = len(mats); mats.append(Material((120.0, 170.0, 200.0), reflectivity=0.5, shininess=100.0))
    assert len(mats) == 25

    X0, X1, Y0, Y1, Z0, Z1 = -12.0, 12.0, -3.0, 9.0, -62.0, -2.0
    n_cols_per_row = 10
    col_z = np.linspace(Z1 - 4.0, Z0 + 4.0, n_cols_per_row)

    def build(k: float) -> List[Mesh]:
        """k scales linear tessellation density."""
        def q(n):  # at least 1 segment
            return max(int(round(n * k)), 1)
        ms: List[Mesh] = []
        # shell: floor (reflective), ceiling, side walls, back wall
        ms.append(quad((X0, Y0, Z1), (X1, Y0, Z1), (X1, Y0, Z0), (X0, Y0, Z0), FLOOR, q(48), q(120)))
        ms.append(quad((X0, Y1, Z0), (X1, Y1, Z0), (X1, Y1, Z1), (X0, Y1, Z1), 0, q(24), q(60)))
        ms.append(quad((X0, Y0, Z1), (X0, Y0, Z0), (X0, Y1, Z0), (X0, Y1, Z1), 1, q(60), q(12)))
        ms.append(quad((X1, Y0, Z0), (X1, Y0, Z1), (X1, Y1, Z1), (X1, Y1, Z0), 1, q(60), q(12)))
        ms.append(quad((X0, Y0, Z0), (X1, Y0, Z0), (X1, Y1, Z0), (X0, Y1, Z0), 2, q(24), q(12)))
        # two rows of columns with plinths and capitals
        for row, cx in enumerate((-6.0, 6.0)):
            for ci, cz in enumerate(col_z):
                mat = MARBLE if (ci + row) % 3 == 0 else 3 + ((ci * 2 + row) % 17)
                ms.append(_cylinder(cx, cz, Y0 + 0.6, Y1 - 3.0, 0.55, q(40), q(48), mat))
                ms += box((cx - 0.9, Y0, cz - 0.9), (cx + 0.9, Y0 + 0.6, cz + 0.9), 2, faces="xXYzZ", n=q(3))
                ms += box((cx - 0.8, Y1 - 3.0, cz - 0.8), (cx + 0.8, Y1 - 2.5, cz + 0.8), 2, faces="xXyzZ", n=q(3))
        # arches across the nave and along the aisles
        for ci, cz in enumerate(col_z):
            ms += _arch(cz, -6.0, 6.0, Y1 - 2.5, 1.8, 1.2, q(64), q(8), 4 + (ci % 5))
            ms += _arch(cz, X0, -6.0, Y1 - 2.5, 1.0, 1.2, q(32), q(8), 9)
            ms += _arch(cz, 6.0, X1, Y1 - 2.5, 1.0, 1.2, q(32), q(8), 9)
        # drapes between alternate columns (each side of the nave, both faces)
        for ci in range(0, n_cols_per_row - 1, 2):
            z_a, z_b = col_z[ci] - 0.8, col_z[ci + 1] + 0.8
            for cx in (-6.0, 6.0):
                mat = 5 + ((ci + (cx > 0)) % 10)
                amp = 0.25 + 0.1 * rng.rand()
                ms.append(_drape(cx, z_a, z_b, Y1 - 3.2, Y0 + 1.5, amp, 5, q(96), q(64), mat, +1))
                ms.append(_drape(cx, z_a, z_b, Y1 - 3.2, Y0 + 1.5, amp, 5, q(96), q(64), mat, -1))
        # reflective / refractive ornaments floating in the nave
        for si in range(8):
            c = (rng.uniform(-3.5, 3.5), rng.uniform(-1.5, 3.0), -8.0 - 6.0 * si + rng.uniform(-1, 1))
            r = rng.uniform(0.7, 1.3)
            ms.append(_uv_sphere(c, r, q(64), q(32), CHROME if si % 2 == 0 else GLASS))
        # a shallow rippled pool
        def pool(u, v):
            x = -3.0 + 6.0 * u
            z = -30.0 - 10.0 * v
            y = Y0 + 0.05 + 0.03 * np.sin(9 * x) * np.cos(7 * z)
            pos = np.stack([x, y, z], -1)
            n = np.stack([-0.27 * np.cos(9 * x) * np.cos(7 * z), np.ones_like(x), 0.21 * np.sin(9 * x) * np.sin(7 * z)], -1)
            n /= np.linalg.norm(n, axis=-1, keepdims=True)
            return pos, n
        ms.append(param_surface(pool, q(96), q(160), WATER))
        return ms

    # solve k: triangle count grows ~ k^2
    k = 1.0
    for _ in range(12):
        n = sum(m.faces.shape[0] for m in build(k))
        if abs(n - target_tris) <= 0.005 * target_tris:
            break
        k *= (target_tris / n) ** 0.5
    meshes = build(k)
    lights = [
        Light((0.0, 7.5, -8.0), (255.0, 244.0, 230.0), 1.0),
        Light((-4.0, 7.0, -24.0), (230.0, 240.0, 255.0), 1.0),
        Light((4.0, 7.0, -40.0), (255.0, 250.0, 240.0), 1.0),
        Light((0.0, 7.5, -56.0), (255.0, 255.0, 255.0), 1.0),
    ]
    return Scene("atrium-%d" % sum(m.faces.shape[0] for m in meshes), mats, meshes, lights)


def small_atrium(target_tris: int = 2048, seed: int = 3) -> Scene:
    """Same generator at test size (oracle-friendly)."""
    return atrium(target_tris, seed)


def with_transforms(scene: Scene, seed: int = 11) -> Scene:
    """Copy of `scene` whose meshes carry non-trivial model matrices (rotation+scale+translation)
    while describing the same world-space geometry is NOT attempted: this is a different scene,
    used to exercise the model_mat path of the soup rebuild (rt_raytracer.cpp:466-475)."""
    rng = np.random.RandomState(seed)
    out = []
    for m in scene.meshes:
        ang = rng.uniform(-0.3, 0.3)
        c, s = np.cos(ang), np.sin(ang)
        sc = rng.uniform(0.8, 1.2)
        mat = np.eye(4, dtype=np.float64)
        mat[:3, :3] = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]]) * sc
        mat[:3, 3] = rng.uniform(-0.5, 0.5, size=3)
        out.append(Mesh(m.verts, m.normals, m.texcoords, m.faces, m.material, mat.astype(F32), m.position))
    return Scene(scene.name + "-xf", scene.materials, out, scene.lights, scene.fov)
