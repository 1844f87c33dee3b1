"""ctypes bindings for the two CPU checkers (TEST INFRASTRUCTURE ONLY).

  Oracle  -> oracle/librt_oracle.so       the CPU restatement (rt_oracle.cpp); travels everywhere
  Ref     -> oracle/_ref/libref_oracle.so the compiled reference; exists only where it was built
                                          from /root/reference (this container), then travels as
                                          a built .so

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "librt_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libref_oracle.so")
DROPIN_SO = os.path.join(HERE, "_ref", "libref_dropin.so")  # reference objects + AccelHip shim (GPU only)

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int)
c_u32_p = C.POINTER(C.c_uint32)
c_u8_p = C.POINTER(C.c_uint8)

RAY_DTYPE = np.dtype([("origin", "<f4", 3), ("direction", "<f4", 3), ("position", "<f4", 3),
                      ("normal", "<f4", 3), ("texcoord", "<f4", 2), ("pad", "<f4", 2)])
assert RAY_DTYPE.itemsize == 64


def build(ref: bool = True) -> None:
    """Compile the restatement (always) and the compiled reference (when /root/reference exists)."""
    subprocess.check_call(["make", "-s", "-C", HERE, "librt_oracle.so"])
    if ref and os.path.isdir("/root/reference/raytracer"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])
        if os.path.exists(os.path.join(HERE, "..", "raytracer_amd", "librt_hip.so")):
            subprocess.check_call(["make", "-s", "-C", HERE, "dropin"])


def have_ref() -> bool:
    return os.path.exists(REF_SO)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t=c_float_p):
    return a.ctypes.data_as(t)


class ShadeConfig(C.Structure):
    _fields_ = [("bounces", C.c_int), ("do_shadows", C.c_int), ("use_photon_map", C.c_int),
                ("photon_estimate", C.c_int), ("photon_radius", C.c_float), ("geometry", C.c_int),
                ("shadow_t_max", C.c_float)]


class Oracle:
    """The CPU restatement, one orc_scene per instance."""

    def __init__(self, scene=None):
        if not os.path.exists(ORACLE_SO):
            build(ref=False)
        self.lib = C.CDLL(ORACLE_SO)
        L = self.lib
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_add_material.argtypes = [C.c_void_p, c_float_p, c_float_p, C.c_float, C.c_float, C.c_float, C.c_float]
        L.orc_add_mesh.argtypes = [C.c_void_p, c_float_p, c_float_p, c_float_p, C.c_int, c_u32_p, C.c_int,
                                   c_float_p, C.c_int, c_float_p]
        L.orc_add_light.argtypes = [C.c_void_p, c_float_p, c_float_p, C.c_float]
        L.orc_num_tris.argtypes = [C.c_void_p]
        L.orc_reflect_raygen.argtypes = [C.c_int, c_float_p, c_float_p, c_float_p, c_float_p]
        L.orc_ray_aabb.argtypes = [c_float_p] * 4
        L.orc_get_soup.argtypes = [C.c_void_p, c_float_p, c_float_p]
        L.orc_process_rays.argtypes = [C.c_void_p, C.c_int, c_float_p, c_float_p, c_u32_p, c_float_p, c_float_p,
                                       c_float_p, c_float_p, C.c_void_p, c_int_p, c_int_p, c_int_p]
        L.orc_soup_closest.argtypes = [C.c_void_p, C.c_int, c_float_p, c_float_p, C.c_float, C.c_float, C.c_float,
                                       c_u32_p, c_float_p, c_float_p, c_float_p]
        L.orc_array_intersect.argtypes = [C.c_void_p, C.c_int, c_float_p, c_u8_p, C.c_void_p, c_int_p, c_int_p, c_float_p]
        L.orc_array_occlude.argtypes = [C.c_void_p, C.c_int, c_float_p, c_u8_p]
        L.orc_spheres_intersect.argtypes = [C.c_void_p, C.c_int, c_float_p, c_u8_p, C.c_void_p, c_float_p, c_int_p]
        L.orc_spheres_occlude.argtypes = [C.c_void_p, C.c_int, c_float_p, c_u8_p]
        L.orc_camera_rays.argtypes = [C.c_float, C.c_int, C.c_int, c_float_p]
        L.orc_gbuffer_encode.argtypes = [C.c_int, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p]
        L.orc_gbuffer_select.argtypes = [C.c_int, c_float_p, c_float_p, C.c_int, C.c_int, c_float_p, c_float_p, c_int_p]
        L.orc_render_texels.argtypes = [C.c_void_p, C.POINTER(ShadeConfig), C.c_int, c_float_p, c_float_p, c_int_p,
                                        c_u8_p, c_float_p, C.POINTER(C.c_int64)]
        L.orc_render_texels_uv.argtypes = [C.c_void_p, C.POINTER(ShadeConfig), C.c_int, c_float_p, c_float_p, c_float_p,
                                           c_int_p, c_u8_p, c_float_p, C.POINTER(C.c_int64)]
        L.orc_set_texture.argtypes = [C.c_void_p, C.c_int, c_u8_p, C.c_int, C.c_int]
        L.orc_shade_records.argtypes = [C.c_void_p, C.c_int, C.c_void_p, c_int_p, c_float_p]
        L.orc_photon_generate.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint64]
        L.orc_photon_generate_range.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_int64, C.c_int64]
        L.orc_photon_count.argtypes = [C.c_void_p]
        L.orc_photon_get.argtypes = [C.c_void_p, c_float_p]
        L.orc_photon_set.argtypes = [C.c_void_p, C.c_int, c_float_p]
        L.orc_radiance_estimate.argtypes = [C.c_void_p, C.c_int, c_float_p, c_float_p, C.c_uint, C.c_float, c_float_p]
        L.orc_photon_lookup.argtypes = [C.c_void_p, C.c_int, c_float_p, C.c_float, c_float_p]
        self.h = C.c_void_p(L.orc_scene_create())
        self.scene = None
        if scene is not None:
            self.load_scene(scene)

    def __del__(self):
        try:
            self.lib.orc_scene_destroy(self.h)
        except Exception:
            pass

    def load_scene(self, scene):
        self.scene = scene
        L = self.lib
        for m in scene.materials:
            L.orc_add_material(self.h, _p(_f(m.diffuse)), _p(_f(m.specular)), m.shininess, m.reflectivity, m.alpha, m.ior)
        for i, m in enumerate(scene.materials):
            tex = getattr(m, "texture", None)
            if tex is not None:                                      # (h, w, 4) u8, row 0 = bottom scanline
                tex = np.ascontiguousarray(tex, dtype=np.uint8)
                L.orc_set_texture(self.h, i, _p(tex, c_u8_p), tex.shape[1], tex.shape[0])
        for me in scene.meshes:
            L.orc_add_mesh(self.h, _p(me.verts), _p(me.normals), _p(me.texcoords), me.verts.shape[0],
                           _p(me.faces, c_u32_p), me.faces.shape[0], _p(me.model_colmajor()), me.material,
                           _p(_f(me.position)))
        for li in scene.lights:
            L.orc_add_light(self.h, _p(_f(li.position)), _p(_f(li.color)), li.energy)

    # ---- geometry -------------------------------------------------------------------------
    def num_tris(self):
        return self.lib.orc_num_tris(self.h)

    def soup(self):
        n = self.num_tris()
        v = np.zeros((n, 9), np.float32)
        b = np.zeros(6, np.float32)
        self.lib.orc_get_soup(self.h, _p(v), _p(b))
        return v, b

    def reflect_raygen(self, pos, nrm):
        pos, nrm = _f(pos), _f(nrm)
        n = pos.shape[0]
        o = np.zeros((n, 3), np.float32)
        d = np.zeros((n, 3), np.float32)
        self.lib.orc_reflect_raygen(n, _p(pos), _p(nrm), _p(o), _p(d))
        return o, d

    def ray_aabb(self, o, d, mn, mx):
        return bool(self.lib.orc_ray_aabb(_p(_f(o)), _p(_f(d)), _p(_f(mn)), _p(_f(mx))))

    def process_rays(self, pos, nrm):
        pos, nrm = _f(pos), _f(nrm)
        n = pos.shape[0]
        out = dict(tidx=np.zeros(n, np.uint32), t=np.zeros(n, np.float32), u=np.zeros(n, np.float32),
                   v=np.zeros(n, np.float32), dir=np.zeros((n, 3), np.float32))
        rec = np.zeros(max(n, 1), RAY_DTYPE)
        nm_cap = max(len(self.scene.materials), 1) + 1
        om = np.zeros(nm_cap, np.int32)
        oe = np.zeros(nm_cap, np.int32)
        nmat = C.c_int(0)
        nres = self.lib.orc_process_rays(self.h, n, _p(pos), _p(nrm), _p(out["tidx"], c_u32_p), _p(out["t"]),
                                         _p(out["u"]), _p(out["v"]), _p(out["dir"]), rec.ctypes.data_as(C.c_void_p),
                                         _p(om, c_int_p), _p(oe, c_int_p), C.byref(nmat))
        out["records"] = rec[:nres]
        out["offset_material"] = om[:nmat.value].copy()
        out["offset_end"] = oe[:nmat.value].copy()
        return out

    def soup_closest(self, org, dir, eps_det=float(np.finfo(np.float32).eps), eps_t=float(np.finfo(np.float32).eps),
                     t_max=float("inf")):
        org, dir = _f(org), _f(dir)
        n = org.shape[0]
        tidx = np.zeros(n, np.uint32)
        t = np.zeros(n, np.float32)
        u = np.zeros(n, np.float32)
        v = np.zeros(n, np.float32)
        self.lib.orc_soup_closest(self.h, n, _p(org), _p(dir), eps_det, eps_t, t_max, _p(tidx, c_u32_p), _p(t), _p(u), _p(v))
        return dict(tidx=tidx, t=t, u=u, v=v)

    def array_intersect(self, rays6):
        rays6 = _f(rays6)
        n = rays6.shape[0]
        hit = np.zeros(n, np.uint8)
        rec = np.zeros(n, RAY_DTYPE)
        mat = np.zeros(n, np.int32)
        idx = np.zeros(n, np.int32)
        t = np.zeros(n, np.float32)
        self.lib.orc_array_intersect(self.h, n, _p(rays6), _p(hit, c_u8_p), rec.ctypes.data_as(C.c_void_p),
                                     _p(mat, c_int_p), _p(idx, c_int_p), _p(t))
        return dict(hit=hit, records=rec, material=mat, index=idx, t=t)

    def array_occlude(self, rays6):
        rays6 = _f(rays6)
        n = rays6.shape[0]
        occ = np.zeros(n, np.uint8)
        self.lib.orc_array_occlude(self.h, n, _p(rays6), _p(occ, c_u8_p))
        return occ

    def spheres_intersect(self, rays6):
        rays6 = _f(rays6)
        n = rays6.shape[0]
        hit = np.zeros(n, np.uint8)
        rec = np.zeros(n, RAY_DTYPE)
        dif = np.zeros((n, 3), np.float32)
        idx = np.zeros(n, np.int32)
        self.lib.orc_spheres_intersect(self.h, n, _p(rays6), _p(hit, c_u8_p), rec.ctypes.data_as(C.c_void_p), _p(dif),
                                       _p(idx, c_int_p))
        return dict(hit=hit, records=rec, diffuse=dif, index=idx)

    def spheres_occlude(self, rays6):
        rays6 = _f(rays6)
        n = rays6.shape[0]
        occ = np.zeros(n, np.uint8)
        self.lib.orc_spheres_occlude(self.h, n, _p(rays6), _p(occ, c_u8_p))
        return occ

    def camera_rays(self, fov, w, h):
        d = np.zeros((w * h, 3), np.float32)
        self.lib.orc_camera_rays(fov, w, h, _p(d))
        return d

    # ---- G-buffer -------------------------------------------------------------------------
    def gbuffer_encode(self, normal, position, alpha):
        normal, position, alpha = _f(normal), _f(position), _f(alpha)
        n = normal.shape[0]
        nr = np.zeros((n, 4), np.float32)
        po = np.zeros((n, 3), np.float32)
        self.lib.orc_gbuffer_encode(n, _p(normal), _p(position), _p(alpha), _p(nr), _p(po))
        return nr, po

    def gbuffer_select(self, nrm_rgba, pos_rgb, interlace=1, phase=0):
        nrm_rgba, pos_rgb = _f(nrm_rgba), _f(pos_rgb)
        n = nrm_rgba.shape[0]
        op = np.zeros((n, 3), np.float32)
        on = np.zeros((n, 3), np.float32)
        oi = np.zeros(n, np.int32)
        c = self.lib.orc_gbuffer_select(n, _p(nrm_rgba), _p(pos_rgb), interlace, phase, _p(op), _p(on), _p(oi, c_int_p))
        return op[:c].copy(), on[:c].copy(), oi[:c].copy()

    def software_gbuffer(self, w, h, fov=None, with_uv=False):
        """Primary visibility by camera rays (Camera::getScreenRays geometry) against the object-space
        triangles with AccelArray semantics, then the reference's G-buffer encodings.  alpha =
        material id + 1 (0 = background).  Returns (nrm_rgba, pos_rgb)."""
        fov = self.scene.fov if fov is None else fov
        d = self.camera_rays(fov, w, h)
        rays6 = np.concatenate([np.zeros_like(d), d], axis=1)
        r = self.array_intersect(rays6)
        alpha = np.where(r["hit"] > 0, r["material"] + 1, 0).astype(np.float32)
        nrm = np.where(r["hit"][:, None] > 0, r["records"]["normal"], np.array([0, 0, 1], np.float32))
        pos = np.where(r["hit"][:, None] > 0, r["records"]["position"], 0).astype(np.float32)
        if with_uv:                                                  # the rasterizer's interpolated outCoord
            uv = np.where(r["hit"][:, None] > 0, r["records"]["texcoord"], 0).astype(np.float32)
            return self.gbuffer_encode(nrm, pos, alpha) + (np.ascontiguousarray(uv),)
        return self.gbuffer_encode(nrm, pos, alpha)

    # ---- shading --------------------------------------------------------------------------
    def render_texels(self, pos, nrm, material_id, bounces=2, do_shadows=0, use_photon_map=0, photon_estimate=100,
                      photon_radius=1.0, geometry=0, shadow_t_max=0.0, uv=None):
        pos, nrm = _f(pos), _f(nrm)
        uv = None if uv is None else _f(uv)
        mid = np.ascontiguousarray(material_id, dtype=np.int32)
        n = pos.shape[0]
        cfg = ShadeConfig(bounces, do_shadows, use_photon_map, photon_estimate, photon_radius, geometry, shadow_t_max)
        rgba = np.zeros((n, 4), np.uint8)
        rgbf = np.zeros((n, 3), np.float32)
        rays = C.c_int64(0)
        self.lib.orc_render_texels_uv(self.h, C.byref(cfg), n, _p(pos), _p(nrm), None if uv is None else _p(uv),
                                      _p(mid, c_int_p), _p(rgba, c_u8_p), _p(rgbf), C.byref(rays))
        return rgba, rgbf, rays.value

    def shade_records(self, records, material_id):
        records = np.ascontiguousarray(records)
        mid = np.ascontiguousarray(material_id, dtype=np.int32)
        n = records.shape[0]
        rgb = np.zeros((n, 3), np.float32)
        self.lib.orc_shade_records(self.h, n, records.ctypes.data_as(C.c_void_p), _p(mid, c_int_p), _p(rgb))
        return rgb

    # ---- photons --------------------------------------------------------------------------
    def photon_generate(self, count, geometry=0, rng=0, seed=0):
        n = self.lib.orc_photon_generate(self.h, count, geometry, rng, C.c_uint64(seed))
        out = np.zeros((n, 9), np.float32)
        self.lib.orc_photon_get(self.h, _p(out))
        return out

    def photon_generate_range(self, count, first, n, geometry=0, seed=0):
        """Counter-RNG photons of the global indices [first, first+n) only (one rank's shard)."""
        k = self.lib.orc_photon_generate_range(self.h, count, geometry, C.c_uint64(seed), C.c_int64(first), C.c_int64(n))
        out = np.zeros((k, 9), np.float32)
        self.lib.orc_photon_get(self.h, _p(out))
        return out

    def photon_set(self, photons9):
        photons9 = _f(photons9)
        self.lib.orc_photon_set(self.h, photons9.shape[0], _p(photons9))

    def radiance_estimate(self, pos, nrm, count=100, radius=1.0):
        pos, nrm = _f(pos), _f(nrm)
        out = np.zeros_like(pos)
        self.lib.orc_radiance_estimate(self.h, pos.shape[0], _p(pos), _p(nrm), count, radius, _p(out))
        return out

    def photon_lookup(self, pos, radius=1.0):
        pos = _f(pos)
        out = np.zeros_like(pos)
        self.lib.orc_photon_lookup(self.h, pos.shape[0], _p(pos), radius, _p(out))
        return out


# ---- the GL side of the live frame with a prepass smaller than the frame (numpy restatement, float32 like a shader) ----
def raster_point_texel(pos, fov, frame_w, frame_h, pw, ph):
    """Where drawRayData's GL_POINTS land (ras_rasterizer.cpp:619-669, viewport = prepass size): gl_Position = proj_mat *
    vec4(Ray.origin, 1) (mesh_vs.cpp:17) with projectionFromCamera's matrix (ras_rasterizer.cpp:140-153: n = 1, f = 100,
    r = frame aspect * scale, t = scale), GL's clip test, viewport transform, and the fragment of a size-1 point = the texel
    containing the window position.  scale = tan(fov) as this build's G-buffer camera has it (the reference applies tan() to an
    already-tan'd value, ras_rasterizer.cpp:143 / camera.cpp:41-44,60-63; it is self-consistent, and so is this).
    Returns the texel index per point, -1 = clipped.  Parity unpinned (no GL here)."""
    f32 = np.float32
    P = np.asarray(pos, np.float32).reshape(-1, 3)
    n_, f_ = f32(1.0), f32(100.0)
    scale = f32(np.tan(np.float64(f32(fov))) * np.float64(n_))
    r = f32(frame_w) / f32(frame_h) * scale
    t = scale
    m00, m11 = n_ / r, n_ / t
    m22, m32 = -(f_ + n_) / (f_ - n_), f32(-2) * f_ * n_ / (f_ - n_)
    cx, cy, cz, cw = m00 * P[:, 0], m11 * P[:, 1], m22 * P[:, 2] + m32, -P[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        inside = (cx >= -cw) & (cx <= cw) & (cy >= -cw) & (cy <= cw) & (cz >= -cw) & (cz <= cw)
        xw = (cx / cw) * (f32(0.5) * f32(pw)) + f32(0.5) * f32(pw)
        yw = (cy / cw) * (f32(0.5) * f32(ph)) + f32(0.5) * f32(ph)
    xw, yw = np.where(inside, xw, f32(-1)), np.where(inside, yw, f32(-1))
    ix, iy = np.floor(xw).astype(np.int64), np.floor(yw).astype(np.int64)
    ok = inside & (ix >= 0) & (ix < pw) & (iy >= 0) & (iy < ph)
    return np.where(ok, iy * pw + ix, -1).astype(np.int64)


def raypass_linear(raypass_u8, pw, ph, w, h):
    """texture2D(raypassBuffer, gl_FragCoord.st / frame_size) (mesh_fs.cpp:46) for every pixel of a w x h frame: the pw x ph
    RGB8 raypass target (UNORM8 -> c / 255) with GL_LINEAR filters (ras_rasterizer.cpp:600-601) and the default GL_REPEAT
    wrap.  Returns (w * h, 3) float32.  Parity unpinned (filter precision is the GL implementation's)."""
    f32 = np.float32
    T = np.asarray(raypass_u8, np.uint8).reshape(ph, pw, 3).astype(np.float32) / f32(255.0)
    xs, ys = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
    u, v = (xs + f32(0.5)) / f32(w), (ys + f32(0.5)) / f32(h)
    x, y = u * f32(pw) - f32(0.5), v * f32(ph) - f32(0.5)
    fx, fy = np.floor(x), np.floor(y)
    a, b = (x - fx)[..., None], (y - fy)[..., None]
    x0, y0 = np.mod(fx.astype(np.int64), pw), np.mod(fy.astype(np.int64), ph)
    x1, y1 = (x0 + 1) % pw, (y0 + 1) % ph
    one = f32(1.0)
    lo = T[y0, x0] * (one - a) + T[y0, x1] * a
    hi = T[y1, x0] * (one - a) + T[y1, x1] * a
    return (lo * (one - b) + hi * b).reshape(-1, 3).astype(np.float32)


class Ref:
    """The compiled reference behind oracle/ref_harness.cpp (process-global state)."""

    def __init__(self, scene=None, so_path=None):
        so_path = so_path or REF_SO
        if not os.path.exists(so_path):
            raise RuntimeError("compiled reference not built (%s)" % so_path)
        self.lib = C.CDLL(so_path)
        L = self.lib
        L.ref_add_material.argtypes = [c_float_p, c_float_p, C.c_float, C.c_float, C.c_float, C.c_float]
        L.ref_add_mesh.argtypes = [c_float_p, c_float_p, c_float_p, C.c_int, c_u32_p, C.c_int, c_float_p, C.c_int, c_float_p]
        L.ref_add_light.argtypes = [c_float_p, c_float_p, C.c_float]
        L.ref_process_rays.argtypes = [C.c_int, c_float_p, c_float_p, c_u32_p, c_float_p, c_float_p, c_float_p, c_float_p,
                                       C.c_void_p, c_int_p, c_int_p, c_int_p]
        L.ref_array_intersect.argtypes = [C.c_int, c_float_p, c_u8_p, C.c_void_p, c_int_p]
        L.ref_array_occlude.argtypes = [C.c_int, c_float_p, c_u8_p]
        L.ref_spheres_intersect.argtypes = [C.c_int, c_float_p, c_u8_p, C.c_void_p, c_float_p]
        L.ref_spheres_occlude.argtypes = [C.c_int, c_float_p, c_u8_p]
        L.ref_photon_generate.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_longlong)]
        L.ref_photon_get.argtypes = [c_float_p]
        L.ref_photon_set.argtypes = [C.c_int, c_float_p]
        L.ref_radiance_estimate.argtypes = [C.c_int, c_float_p, c_float_p, C.c_uint, C.c_float, c_float_p]
        L.ref_photon_lookup.argtypes = [C.c_int, c_float_p, C.c_float, c_float_p]
        L.ref_camera_rays.argtypes = [C.c_float, C.c_int, C.c_int, c_float_p]
        self.scene = None
        if scene is not None:
            self.load_scene(scene)

    def sizes(self):
        L = self.lib
        return dict(ray=L.ref_sizeof_ray(), material=L.ref_sizeof_material(), light=L.ref_sizeof_light(),
                    tri=L.ref_sizeof_tri())

    def load_scene(self, scene):
        self.scene = scene
        L = self.lib
        L.ref_reset()
        for m in scene.materials:
            L.ref_add_material(_p(_f(m.diffuse)), _p(_f(m.specular)), m.shininess, m.reflectivity, m.alpha, m.ior)
        for me in scene.meshes:
            L.ref_add_mesh(_p(me.verts), _p(me.normals), _p(me.texcoords), me.verts.shape[0], _p(me.faces, c_u32_p),
                           me.faces.shape[0], _p(me.model_colmajor()), me.material, _p(_f(me.position)))
        for li in scene.lights:
            L.ref_add_light(_p(_f(li.position)), _p(_f(li.color)), li.energy)

    def process_rays(self, pos, nrm):
        pos, nrm = _f(pos), _f(nrm)
        n = pos.shape[0]
        out = dict(tidx=np.zeros(n, np.uint32), t=np.zeros(n, np.float32), u=np.zeros(n, np.float32),
                   v=np.zeros(n, np.float32), dir=np.zeros((n, 3), np.float32))
        rec = np.zeros(max(n, 1), RAY_DTYPE)
        nm_cap = max(len(self.scene.materials), 1) + 1
        om = np.zeros(nm_cap, np.int32)
        oe = np.zeros(nm_cap, np.int32)
        nmat = C.c_int(0)
        nres = self.lib.ref_process_rays(n, _p(pos), _p(nrm), _p(out["tidx"], c_u32_p), _p(out["t"]), _p(out["u"]),
                                         _p(out["v"]), _p(out["dir"]), rec.ctypes.data_as(C.c_void_p), _p(om, c_int_p),
                                         _p(oe, c_int_p), C.byref(nmat))
        out["records"] = rec[:nres]
        out["offset_material"] = om[:nmat.value].copy()
        out["offset_end"] = oe[:nmat.value].copy()
        return out

    def array_intersect(self, rays6):
        rays6 = _f(rays6)
        n = rays6.shape[0]
        hit = np.zeros(n, np.uint8)
        rec = np.zeros(n, RAY_DTYPE)
        mat = np.zeros(n, np.int32)
        self.lib.ref_array_intersect(n, _p(rays6), _p(hit, c_u8_p), rec.ctypes.data_as(C.c_void_p), _p(mat, c_int_p))
        return dict(hit=hit, records=rec, material=mat)

    def array_occlude(self, rays6):
        rays6 = _f(rays6)
        occ = np.zeros(rays6.shape[0], np.uint8)
        self.lib.ref_array_occlude(rays6.shape[0], _p(rays6), _p(occ, c_u8_p))
        return occ

    def spheres_intersect(self, rays6):
        rays6 = _f(rays6)
        n = rays6.shape[0]
        hit = np.zeros(n, np.uint8)
        rec = np.zeros(n, RAY_DTYPE)
        dif = np.zeros((n, 3), np.float32)
        self.lib.ref_spheres_intersect(n, _p(rays6), _p(hit, c_u8_p), rec.ctypes.data_as(C.c_void_p), _p(dif))
        return dict(hit=hit, records=rec, diffuse=dif)

    def spheres_occlude(self, rays6):
        rays6 = _f(rays6)
        occ = np.zeros(rays6.shape[0], np.uint8)
        self.lib.ref_spheres_occlude(rays6.shape[0], _p(rays6), _p(occ, c_u8_p))
        return occ

    def camera_rays(self, fov, w, h):
        d = np.zeros((w * h, 3), np.float32)
        self.lib.ref_camera_rays(fov, w, h, _p(d))
        return d

    def photon_generate(self, count, geometry=0):
        seed = C.c_longlong(0)
        n = self.lib.ref_photon_generate(count, geometry, C.byref(seed))
        if n < 0:
            raise RuntimeError("reference photon generation kept straddling a second boundary")
        out = np.zeros((n, 9), np.float32)
        self.lib.ref_photon_get(_p(out))
        return out, seed.value

    def photon_set(self, photons9):
        photons9 = _f(photons9)
        self.lib.ref_photon_set(photons9.shape[0], _p(photons9))

    def radiance_estimate(self, pos, nrm, count=100, radius=1.0):
        pos, nrm = _f(pos), _f(nrm)
        out = np.zeros_like(pos)
        self.lib.ref_radiance_estimate(pos.shape[0], _p(pos), _p(nrm), count, radius, _p(out))
        return out

    def photon_lookup(self, pos, radius=1.0):
        pos = _f(pos)
        out = np.zeros_like(pos)
        self.lib.ref_photon_lookup(pos.shape[0], _p(pos), radius, _p(out))
        return out
