"""raytracer_amd -- MI355X-native ray-tracing back end for the G-buffer ray tracer.

The product is `librt_hip.so` (hand-written HIP for gfx950 behind the C ABI of
`include/rt_hip.h`).  This package is the thin Python host binding used by the tests and by
bench.py; it holds no compute and no CPU fallback: if the shared library is missing or no GPU is
visible, calls fail loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# RT_HIP_LIBRARY: another build of the same library (kernel-variant experiments, scripts/)
LIB_PATH = os.environ.get("RT_HIP_LIBRARY") or os.path.join(HERE, "librt_hip.so")

c_float_p = C.POINTER(C.c_float)
c_u8_p = C.POINTER(C.c_uint8)
c_u32_p = C.POINTER(C.c_uint32)
c_int_p = C.POINTER(C.c_int)

RAY_DTYPE = np.dtype([("origin", "<f4", 3), ("direction", "<f4", 3), ("position", "<f4", 3),
                      ("normal", "<f4", 3), ("texcoord", "<f4", 2), ("pad", "<f4", 2)])
HIT_DTYPE = np.dtype([("prim", "<u4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4")])
OFFSET_DTYPE = np.dtype([("material", "<u4"), ("end", "<i4")])
MATERIAL_DTYPE = np.dtype([("diffuse_color", "<f4", 3), ("specular_color", "<f4", 3), ("shininess", "<f4"),
                           ("reflectivity", "<f4"), ("alpha", "<f4"), ("ior", "<f4"), ("texture", "<u8")])
LIGHT_DTYPE = np.dtype([("position", "<f4", 3), ("color", "<f4", 3), ("energy", "<f4")])
assert RAY_DTYPE.itemsize == 64 and MATERIAL_DTYPE.itemsize == 48 and LIGHT_DTYPE.itemsize == 28

MISS = 0xFFFFFFFF
TRIANGLES, SPHERES = 0, 1
ACCEL_AUTO, ACCEL_BRUTE, ACCEL_LBVH, ACCEL_PLOC, ACCEL_SAH = 0, 1, 2, 3, 4
NOALLOC_MAX_TEXELS = 1 << 24   # RT_HIP_NOALLOC_MAX_TEXELS (include/rt_hip.h)


class Config(C.Structure):
    _fields_ = [("device", C.c_int), ("geometry", C.c_int), ("accel", C.c_int), ("use_model_mat", C.c_int),
                ("eps_det", C.c_float), ("eps_t", C.c_float), ("t_max", C.c_float), ("bounces", C.c_int),
                ("do_shadows", C.c_int), ("use_photon_map", C.c_int), ("photon_count", C.c_int),
                ("photon_estimate", C.c_int), ("photon_radius", C.c_float), ("photon_max_dist", C.c_float),
                ("photon_absorb", C.c_float), ("seed", C.c_uint64), ("profile", C.c_int), ("shadow_t_max", C.c_float), ("census", C.c_int), ("reserved", C.c_int)]


class Stats(C.Structure):
    _fields_ = [("closest_ms", C.c_double), ("closest_launches", C.c_int64), ("rays_closest", C.c_int64),
                ("any_ms", C.c_double), ("any_launches", C.c_int64), ("rays_any", C.c_int64),
                ("build_ms", C.c_double), ("num_tris", C.c_int64), ("num_nodes", C.c_int64),
                ("bvh_depth", C.c_int32), ("last_update", C.c_int32), ("frame_ms", C.c_double), ("frames", C.c_int64),
                ("census_nodes_closest", C.c_uint64), ("census_tris_closest", C.c_uint64),
                ("census_rays_closest", C.c_uint64), ("census_nodes_any", C.c_uint64),
                ("census_tris_any", C.c_uint64), ("census_rays_any", C.c_uint64),
                ("gather_ms", C.c_double), ("gather_launches", C.c_int64), ("gather_points", C.c_int64),
                ("census_gather_visits", C.c_uint64), ("bvh_area_cost", C.c_double), ("bvh_area_cost_built", C.c_double)]


class Tuning(C.Structure):
    """rt_hip_tuning (include/rt_hip.h): knobs outside the reference's semantics."""
    _fields_ = [("lds_stack_depth", C.c_int), ("xcd_chunk", C.c_int), ("trace_oversub", C.c_int), ("auto_accel", C.c_int),
                ("ploc_radius", C.c_int), ("frame_sync", C.c_int), ("gather_heavy", C.c_uint32), ("extra_lds_bytes", C.c_int),
                ("max_bvh_depth", C.c_int), ("refit_max_growth", C.c_float), ("gather_fine", C.c_int), ("gather_xcd_chunk", C.c_int), ("gather_fine_cells", C.c_int), ("gather_medium", C.c_uint32), ("shadow_prefix", C.c_int), ("shadow_prefix_steps", C.c_int), ("shadow_prefix_min_rays", C.c_int), ("frame_streams", C.c_int), ("gather_pair", C.c_int), ("gather_sort", C.c_int)]


UPDATE_NONE, UPDATE_TABLES, UPDATE_REFIT, UPDATE_REBUILD = 0, 1, 2, 3

# every symbol include/rt_hip.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "rt_hip_default_config", "rt_hip_array_config", "rt_hip_create", "rt_hip_destroy", "rt_hip_last_error",
    "rt_hip_set_config", "rt_hip_set_materials", "rt_hip_set_lights", "rt_hip_add_mesh", "rt_hip_add_sphere",
    "rt_hip_set_model_mat", "rt_hip_update", "rt_hip_intersect", "rt_hip_occlude", "rt_hip_sphere_colors",
    "rt_hip_process_gbuffer", "rt_hip_last_hits", "rt_hip_process_gbuffer_device", "rt_hip_trace_closest_device",
    "rt_hip_trace_any_device", "rt_hip_gbuffer_select_device", "rt_hip_render_gbuffer_device", "rt_hip_render_frame",
    "rt_hip_render_frame_device", "rt_hip_render_tiles_device", "rt_hip_last_frame_rays", "rt_hip_photon_emit_device",
    "rt_hip_photon_set_device", "rt_hip_photon_set", "rt_hip_radiance_estimate_device", "rt_hip_radiance_estimate",
    "rt_hip_shade_results", "rt_hip_shade_results_device", "rt_hip_pack_tiles_device", "rt_hip_unpack_tiles_device",
    "rt_hip_render_tiles_batch_device", "rt_hip_live_frame_device", "rt_hip_live_frame_scaled_device", "rt_hip_render_tiles_packed_device",
    "rt_hip_unpack_chunks_device", "rt_hip_set_texture", "rt_hip_set_frame_texcoords_device",
    "rt_hip_render_gbuffer_uv_device",
    "rt_hip_get_stats", "rt_hip_reset_stats", "rt_hip_sync", "rt_hip_stream", "rt_hip_get_bvh",
    "rt_hip_trace_census_device", "rt_hip_gbuffer_select_noalloc_device",
    "rt_hip_frame_enqueue_device", "rt_hip_frame_enqueue", "rt_hip_frame_enqueue_g16", "rt_hip_frame_finish", "rt_hip_comm_unique_id", "rt_hip_comm_init", "rt_hip_comm_destroy", "rt_hip_render_frame_multi_device", "rt_hip_frame_multi_enqueue_device", "rt_hip_frames_multi_enqueue_device", "rt_hip_tiles_packed_enqueue_device", "rt_hip_tiles_packed_batch_enqueue_device",
    "rt_hip_photon_generate_multi", "rt_hip_set_model_mats", "rt_hip_photon_generate", "rt_hip_photon_get", "rt_hip_get_tuning", "rt_hip_set_tuning", "rt_hip_invalidate", "rt_hip_photon_lookup_device", "rt_hip_photon_lookup",
]


def build(verbose: bool = False) -> str:
    """Compile librt_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(HERE, "csrc"), "-j8"]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None


def load_library() -> C.CDLL:
    """dlopen librt_hip.so.  Never falls back to anything else."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("librt_hip.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                           "raytracer_amd has no CPU fallback")
    # One HIP runtime per process: PyTorch's wheel bundles its own libamdhip64.  If librt_hip.so
    # were loaded first it would bind /opt/rocm's copy and torch.cuda would then find no device
    # (measured on the GPU box).  Importing torch first makes both share torch's runtime; callers
    # that never use torch tensors are unaffected.
    import sys
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(LIB_PATH)
    P = C.c_void_p
    L.rt_hip_last_error.restype = C.c_char_p
    L.rt_hip_default_config.argtypes = [C.POINTER(Config)]
    L.rt_hip_array_config.argtypes = [C.POINTER(Config)]
    L.rt_hip_create.argtypes = [C.POINTER(Config), C.POINTER(P)]
    L.rt_hip_destroy.argtypes = [P]
    L.rt_hip_set_config.argtypes = [P, C.POINTER(Config)]
    L.rt_hip_set_materials.argtypes = [P, P, C.c_int]
    L.rt_hip_set_lights.argtypes = [P, P, C.c_int]
    L.rt_hip_add_mesh.argtypes = [P, c_float_p, c_float_p, c_float_p, C.c_int, c_u32_p, C.c_int, c_float_p, C.c_int, c_float_p]
    L.rt_hip_add_sphere.argtypes = [P, C.c_float, C.c_float, C.c_float, C.c_float]
    L.rt_hip_set_model_mat.argtypes = [P, C.c_int, c_float_p]
    L.rt_hip_set_model_mats.argtypes = [P, C.c_int, C.c_int, c_float_p]
    L.rt_hip_update.argtypes = [P]
    L.rt_hip_intersect.argtypes = [P, P, C.c_int, c_u8_p, c_u32_p, P]
    L.rt_hip_occlude.argtypes = [P, P, C.c_int, c_u8_p]
    L.rt_hip_sphere_colors.argtypes = [P, c_float_p, C.c_int]
    L.rt_hip_process_gbuffer.argtypes = [P, c_float_p, c_float_p, C.c_int, C.POINTER(P), C.POINTER(P), c_int_p, c_int_p]
    L.rt_hip_last_hits.argtypes = [P, P, c_float_p, C.c_int]
    L.rt_hip_process_gbuffer_device.argtypes = [P, P, P, C.c_int, P, P, c_int_p, c_int_p]
    L.rt_hip_trace_closest_device.argtypes = [P, P, P, C.c_int, P]
    L.rt_hip_trace_any_device.argtypes = [P, P, P, C.c_int, P]
    L.rt_hip_trace_census_device.argtypes = [P, P, P, C.c_int, C.c_int, P, P]
    L.rt_hip_gbuffer_select_device.argtypes = [P, P, P, C.c_int, C.c_int, C.c_int, P, P, P, c_int_p]
    L.rt_hip_gbuffer_select_noalloc_device.argtypes = [P, P, P, C.c_int, C.c_int, C.c_int, P, P, P, P, c_int_p]
    L.rt_hip_render_gbuffer_device.argtypes = [P, C.c_float, C.c_int, C.c_int, P, P]
    L.rt_hip_render_gbuffer_uv_device.argtypes = [P, C.c_float, C.c_int, C.c_int, P, P, P]
    L.rt_hip_set_texture.argtypes = [P, C.c_int, P, C.c_int, C.c_int]
    L.rt_hip_set_frame_texcoords_device.argtypes = [P, P, C.c_int]
    L.rt_hip_render_frame.argtypes = [P, c_float_p, c_float_p, C.c_int, C.c_int, c_u8_p]
    L.rt_hip_render_frame_device.argtypes = [P, P, P, C.c_int, C.c_int, P]
    L.rt_hip_render_tiles_device.argtypes = [P, P, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P]
    L.rt_hip_frame_enqueue_device.argtypes = [P, P, P, C.c_int, C.c_int, P]
    L.rt_hip_frame_enqueue.argtypes = [P, P, P, C.c_int, C.c_int, P]
    L.rt_hip_frame_enqueue_g16.argtypes = [P, P, P, C.c_int, C.c_int, P]
    L.rt_hip_frame_finish.argtypes = [P, C.POINTER(C.c_int64)]
    L.rt_hip_last_frame_rays.argtypes = [P]
    L.rt_hip_last_frame_rays.restype = C.c_int64
    L.rt_hip_photon_emit_device.argtypes = [P, C.c_int, C.c_int64, C.c_int64, P, C.c_int64, C.POINTER(C.c_int64)]
    L.rt_hip_photon_generate.argtypes = [P, C.c_int, C.POINTER(C.c_int64)]
    L.rt_hip_photon_get.argtypes = [P, c_float_p, C.c_int64, C.POINTER(C.c_int64)]
    L.rt_hip_photon_set_device.argtypes = [P, P, C.c_int64]
    L.rt_hip_photon_set.argtypes = [P, c_float_p, C.c_int64]
    L.rt_hip_radiance_estimate_device.argtypes = [P, P, P, C.c_int, C.c_int, C.c_float, P]
    L.rt_hip_radiance_estimate.argtypes = [P, c_float_p, c_float_p, C.c_int, C.c_int, C.c_float, c_float_p]
    L.rt_hip_render_tiles_batch_device.argtypes = [P, P, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P]
    L.rt_hip_render_tiles_packed_device.argtypes = [P, P, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P]
    L.rt_hip_unpack_chunks_device.argtypes = [P, P, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, P]
    L.rt_hip_live_frame_device.argtypes = [P, P, P, C.c_int, C.c_int, C.c_int, C.c_int, P, P, C.POINTER(C.c_int)]
    L.rt_hip_live_frame_scaled_device.argtypes = [P, P, P, C.c_int, C.c_int, P, P, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, P, P, C.POINTER(C.c_int)]
    L.rt_hip_pack_tiles_device.argtypes = [P, P, C.c_int, C.c_int, C.c_int, P, C.c_int, P]
    L.rt_hip_unpack_tiles_device.argtypes = [P, P, C.c_int, C.c_int, C.c_int, P, C.c_int, P]
    L.rt_hip_shade_results.argtypes = [P, P, P, C.c_int, C.c_int, c_float_p, P]
    L.rt_hip_shade_results_device.argtypes = [P, P, P, C.c_int, C.c_int, P, P]
    L.rt_hip_comm_unique_id.argtypes = [P]
    L.rt_hip_comm_init.argtypes = [P, P, C.c_int, C.c_int]
    L.rt_hip_comm_destroy.argtypes = [P]
    L.rt_hip_render_frame_multi_device.argtypes = [P, P, P, C.c_int, C.c_int, C.c_int, P]
    L.rt_hip_frame_multi_enqueue_device.argtypes = [P, P, P, C.c_int, C.c_int, C.c_int, P]
    L.rt_hip_frames_multi_enqueue_device.argtypes = [P, P, P, C.c_int, C.c_int, C.c_int, C.c_int, P]
    L.rt_hip_tiles_packed_batch_enqueue_device.argtypes = [P, P, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P]
    L.rt_hip_tiles_packed_enqueue_device.argtypes = [P, P, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P]
    L.rt_hip_photon_generate_multi.argtypes = [P, C.c_int, C.POINTER(C.c_int64)]
    L.rt_hip_get_tuning.argtypes = [P, C.POINTER(Tuning)]
    L.rt_hip_set_tuning.argtypes = [P, C.POINTER(Tuning)]
    L.rt_hip_invalidate.argtypes = [P]
    L.rt_hip_photon_lookup_device.argtypes = [P, P, C.c_int, C.c_float, P]
    L.rt_hip_photon_lookup.argtypes = [P, c_float_p, C.c_int, C.c_float, c_float_p]
    L.rt_hip_get_stats.argtypes = [P, C.POINTER(Stats)]
    L.rt_hip_reset_stats.argtypes = [P]
    L.rt_hip_sync.argtypes = [P]
    L.rt_hip_stream.argtypes = [P]
    L.rt_hip_stream.restype = P
    L.rt_hip_get_bvh.argtypes = [P, c_float_p, C.c_int64, c_u32_p, C.c_int64]
    _lib = L
    return L


class RtHipError(RuntimeError):
    pass


COMM_ID_BYTES = 128   # RT_HIP_COMM_ID_BYTES


def comm_unique_id() -> bytes:
    """rt_hip_comm_unique_id: call on rank 0, hand the bytes to every rank's RtHip.comm_init."""
    L = load_library()
    buf = (C.c_char * COMM_ID_BYTES)()
    if L.rt_hip_comm_unique_id(buf) < 0:
        raise RtHipError(L.rt_hip_last_error().decode())
    return bytes(buf)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t=c_float_p):
    return a.ctypes.data_as(t)


def _dptr(t):
    """Device pointer of a torch tensor (or a raw int address)."""
    if isinstance(t, int):
        return C.c_void_p(t)
    return C.c_void_p(t.data_ptr())


def _fence_torch():
    """The library runs on a stream of its own.  A tensor handed to it may still have work pending on
    torch's current stream (torch.zeros / torch.full / copy_ are asynchronous fills): drain that stream
    first, or the fill can land AFTER the library's kernels have written the buffer."""
    import sys
    t = sys.modules.get("torch")
    if t is not None and t.cuda.is_available() and t.cuda.is_initialized():
        t.cuda.current_stream().synchronize()


def _fenced(method):
    import functools

    @functools.wraps(method)
    def wrapper(self, *args, **kwargs):
        _fence_torch()
        out = method(self, *args, **kwargs)
        self.sync()   # and the other way round: the outputs are complete before torch (or a collective) reads them
        return out
    return wrapper


class RtHip:
    """Host mirror of the reference's tracer objects over the C ABI.

    `semantics="raytracer"` behaves like RayTracer (world space, FLT_EPSILON, t_max = inf);
    `semantics="array"` like AccelArray (object space, 1e-5 as double, t_max = 1e6);
    `geometry=SPHERES` like AccelSpheres.
    """

    def __init__(self, scene=None, semantics="raytracer", geometry=TRIANGLES, accel=ACCEL_AUTO, device=0, profile=0,
                 tuning=None, **cfg_overrides):
        self.lib = load_library()
        self.cfg = Config()
        if semantics == "array":
            self._ck(self.lib.rt_hip_array_config(C.byref(self.cfg)))
        else:
            self._ck(self.lib.rt_hip_default_config(C.byref(self.cfg)))
        self.cfg.geometry = geometry
        self.cfg.accel = accel
        self.cfg.device = device
        self.cfg.profile = profile
        for k, v in cfg_overrides.items():
            setattr(self.cfg, k, v)
        self.h = C.c_void_p()
        self._ck(self.lib.rt_hip_create(C.byref(self.cfg), C.byref(self.h)))
        if tuning:
            self.set_tuning(**tuning)
        self.scene = None
        if scene is not None:
            self.load_scene(scene)

    def _ck(self, status):
        if status is not None and status < 0:
            raise RtHipError("rt_hip status %d: %s" % (status, self.lib.rt_hip_last_error().decode()))
        return status

    def close(self):
        if self.h:
            self.lib.rt_hip_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- scene ----------------------------------------------------------------------------
    def load_scene(self, scene, update=True):
        self.scene = scene
        self.load_scene_tables(scene)
        for me in scene.meshes:
            self._ck(self.lib.rt_hip_add_mesh(self.h, _p(me.verts), _p(me.normals), _p(me.texcoords), me.verts.shape[0],
                                              _p(me.faces, c_u32_p), me.faces.shape[0], _p(me.model_colmajor()),
                                              me.material, _p(_f(me.position))))
        if update:
            self.update()

    def load_scene_tables(self, scene):
        """Scene::materials (with their textures) and Scene::lights; takes effect at the next update()."""
        mats = np.zeros(len(scene.materials), MATERIAL_DTYPE)
        for i, m in enumerate(scene.materials):
            mats[i] = (m.diffuse, m.specular, m.shininess, m.reflectivity, m.alpha, m.ior, 0)
        self._ck(self.lib.rt_hip_set_materials(self.h, mats.ctypes.data_as(C.c_void_p), len(mats)))
        for i, m in enumerate(scene.materials):
            if getattr(m, "texture", None) is not None:
                self.set_texture(i, m.texture)
        lights = np.zeros(len(scene.lights), LIGHT_DTYPE)
        for i, li in enumerate(scene.lights):
            lights[i] = (li.position, li.color, li.energy)
        self._ck(self.lib.rt_hip_set_lights(self.h, lights.ctypes.data_as(C.c_void_p), len(lights)))

    def update(self):
        self._ck(self.lib.rt_hip_update(self.h))

    def set_texture(self, material_id, rgba):
        """Material::texture: (h, w, 4) u8 RGBA with row 0 = the bottom scanline, or None to remove.  Takes effect
        at the next update()."""
        if rgba is None:
            self._ck(self.lib.rt_hip_set_texture(self.h, material_id, None, 0, 0))
            return
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        if rgba.ndim != 3 or rgba.shape[2] != 4:
            raise ValueError("texture must be (h, w, 4) u8")
        self._ck(self.lib.rt_hip_set_texture(self.h, material_id, rgba.ctypes.data_as(C.c_void_p), rgba.shape[1], rgba.shape[0]))

    def set_model_mat(self, mesh_id, model_colmajor16):
        """Mesh::model_mat of one mesh (column-major 4x4); takes effect at the next update()."""
        m = np.ascontiguousarray(model_colmajor16, dtype=np.float32).reshape(16)
        self._ck(self.lib.rt_hip_set_model_mat(self.h, int(mesh_id), _p(m)))

    def tuning(self):
        t = Tuning()
        self._ck(self.lib.rt_hip_get_tuning(self.h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in Tuning._fields_ if k != "reserved"}

    def set_tuning(self, **kw):
        """rt_hip_set_tuning: launch-shape / algorithm knobs (results never depend on them)."""
        t = Tuning()
        self._ck(self.lib.rt_hip_get_tuning(self.h, C.byref(t)))
        for k, v in kw.items():
            if not hasattr(t, k):
                raise AttributeError("rt_hip_tuning has no field %r" % k)
            setattr(t, k, v)
        self._ck(self.lib.rt_hip_set_tuning(self.h, C.byref(t)))

    def invalidate(self):
        """The next update() rebuilds from scratch."""
        self._ck(self.lib.rt_hip_invalidate(self.h))

    def force_rebuild(self):
        self.invalidate()

    def set_model_mats(self, first_mesh, mats_colmajor):
        """Mesh::model_mat of consecutive meshes, (n, 16) column-major; takes effect at the next update()."""
        m = np.ascontiguousarray(mats_colmajor, dtype=np.float32).reshape(-1, 16)
        self._ck(self.lib.rt_hip_set_model_mats(self.h, int(first_mesh), m.shape[0], _p(m)))

    def set_config(self, **kw):
        for k, v in kw.items():
            setattr(self.cfg, k, v)
        self._ck(self.lib.rt_hip_set_config(self.h, C.byref(self.cfg)))

    # ---- IAccel ---------------------------------------------------------------------------
    def intersect(self, rays6):
        rays6 = _f(rays6)
        n = rays6.shape[0]
        rec = np.zeros(n, RAY_DTYPE)
        rec["origin"] = rays6[:, :3]
        rec["direction"] = rays6[:, 3:]
        hit = np.zeros(n, np.uint8)
        mat = np.zeros(n, np.uint32)
        hits = np.zeros(n, HIT_DTYPE)
        self._ck(self.lib.rt_hip_intersect(self.h, rec.ctypes.data_as(C.c_void_p), n, _p(hit, c_u8_p), _p(mat, c_u32_p),
                                           hits.ctypes.data_as(C.c_void_p)))
        return dict(hit=hit, records=rec, material=mat.astype(np.int64).astype(np.int32), hits=hits)

    def occlude(self, rays6):
        rays6 = _f(rays6)
        n = rays6.shape[0]
        rec = np.zeros(n, RAY_DTYPE)
        rec["origin"] = rays6[:, :3]
        rec["direction"] = rays6[:, 3:]
        occ = np.zeros(n, np.uint8)
        self._ck(self.lib.rt_hip_occlude(self.h, rec.ctypes.data_as(C.c_void_p), n, _p(occ, c_u8_p)))
        return occ

    def sphere_colors(self, n):
        out = np.zeros((n, 3), np.float32)
        got = self._ck(self.lib.rt_hip_sphere_colors(self.h, _p(out), n))
        return out[:got]

    # ---- processRays ----------------------------------------------------------------------
    def process_rays(self, pos, nrm):
        pos, nrm = _f(pos), _f(nrm)
        n = pos.shape[0]
        res_p, off_p = C.c_void_p(), C.c_void_p()
        nres, nmat = C.c_int(0), C.c_int(0)
        self._ck(self.lib.rt_hip_process_gbuffer(self.h, _p(pos), _p(nrm), n, C.byref(res_p), C.byref(off_p), C.byref(nres),
                                                 C.byref(nmat)))
        rec = np.zeros(nres.value, RAY_DTYPE)
        if nres.value:
            C.memmove(rec.ctypes.data, res_p.value, nres.value * 64)
        off = np.zeros(nmat.value, OFFSET_DTYPE)
        if nmat.value:
            C.memmove(off.ctypes.data, off_p.value, nmat.value * 8)
        hits = np.zeros(n, HIT_DTYPE)
        dirs = np.zeros((n, 3), np.float32)
        if n:
            self._ck(self.lib.rt_hip_last_hits(self.h, hits.ctypes.data_as(C.c_void_p), _p(dirs), n))
        return dict(tidx=hits["prim"].copy(), t=hits["t"].copy(), u=hits["u"].copy(), v=hits["v"].copy(), dir=dirs,
                    records=rec, offset_material=off["material"].astype(np.int32), offset_end=off["end"].copy())

    # ---- device-pointer entries (torch tensors or raw addresses) -----------------------------
    def trace_closest_device(self, d_org, d_dir, n, d_hits):
        self._ck(self.lib.rt_hip_trace_closest_device(self.h, _dptr(d_org), _dptr(d_dir), n, _dptr(d_hits)))

    def trace_any_device(self, d_org, d_dir, n, d_occ):
        self._ck(self.lib.rt_hip_trace_any_device(self.h, _dptr(d_org), _dptr(d_dir), n, _dptr(d_occ)))

    def trace_census_device(self, d_org, d_dir, n, any_hit, d_nv, d_nt):
        self._ck(self.lib.rt_hip_trace_census_device(self.h, _dptr(d_org), _dptr(d_dir), n, int(any_hit), _dptr(d_nv), _dptr(d_nt)))

    def process_gbuffer_device(self, d_pos, d_nrm, n, d_results, d_offsets):
        nres, nmat = C.c_int(0), C.c_int(0)
        self._ck(self.lib.rt_hip_process_gbuffer_device(self.h, _dptr(d_pos), _dptr(d_nrm), n, _dptr(d_results),
                                                        _dptr(d_offsets), C.byref(nres), C.byref(nmat)))
        return nres.value, nmat.value

    def gbuffer_select_device(self, d_nrm_rgba, d_pos_rgb, n, interlace, phase, d_pos_out, d_nrm_out, d_index_out):
        cnt = C.c_int(0)
        self._ck(self.lib.rt_hip_gbuffer_select_device(self.h, _dptr(d_nrm_rgba), _dptr(d_pos_rgb), n, interlace, phase,
                                                       _dptr(d_pos_out), _dptr(d_nrm_out), _dptr(d_index_out), C.byref(cnt)))
        return cnt.value

    def gbuffer_select_noalloc_device(self, d_nrm_rgba, d_pos_rgb, n, interlace, phase, d_pos_out, d_nrm_out, d_index_out, d_count=None):
        """The allocation-free variant (fixed per-GPU scratch, n <= 2^24).  With d_count (a device int) the count stays on
        the device and nothing is copied or waited for (returns None); otherwise the count is returned."""
        if d_count is not None:
            self._ck(self.lib.rt_hip_gbuffer_select_noalloc_device(self.h, _dptr(d_nrm_rgba), _dptr(d_pos_rgb), n, interlace, phase,
                                                                   _dptr(d_pos_out), _dptr(d_nrm_out), _dptr(d_index_out), _dptr(d_count), None))
            return None
        cnt = C.c_int(0)
        self._ck(self.lib.rt_hip_gbuffer_select_noalloc_device(self.h, _dptr(d_nrm_rgba), _dptr(d_pos_rgb), n, interlace, phase,
                                                               _dptr(d_pos_out), _dptr(d_nrm_out), _dptr(d_index_out), None, C.byref(cnt)))
        return cnt.value

    def render_gbuffer_device(self, fov, w, h, d_nrm_rgba, d_pos_rgb):
        self._ck(self.lib.rt_hip_render_gbuffer_device(self.h, fov, w, h, _dptr(d_nrm_rgba), _dptr(d_pos_rgb)))

    def render_gbuffer_uv_device(self, fov, w, h, d_nrm_rgba, d_pos_rgb, d_uv):
        self._ck(self.lib.rt_hip_render_gbuffer_uv_device(self.h, fov, w, h, _dptr(d_nrm_rgba), _dptr(d_pos_rgb), _dptr(d_uv)))

    def set_frame_texcoords_device(self, d_uv_list):
        """Level-0 texture coordinates (w*h x 2 f32 per frame) for the frame calls that follow; [] clears.  The
        tensors must stay alive until replaced."""
        n = len(d_uv_list)
        arr = (C.c_void_p * max(n, 1))(*[_dptr(t) for t in d_uv_list])
        self._frame_uv_keepalive = list(d_uv_list)
        self._ck(self.lib.rt_hip_set_frame_texcoords_device(self.h, arr if n else None, n))

    def render_frame_device(self, d_pos_rgb, d_nrm_rgba, w, h, d_rgba):
        self._ck(self.lib.rt_hip_render_frame_device(self.h, _dptr(d_pos_rgb), _dptr(d_nrm_rgba), w, h, _dptr(d_rgba)))

    def frame_enqueue_host(self, h_pos_rgb, h_nrm_rgba, w, h, h_rgba):
        """rt_hip_frame_enqueue: the pipeline fed from (pinned) host tensors / arrays; they must stay alive and untouched
        until frame_finish() has returned for this frame."""
        ptr = lambda a: C.c_void_p(a.data_ptr() if hasattr(a, "data_ptr") else a.ctypes.data)
        self._ck(self.lib.rt_hip_frame_enqueue(self.h, ptr(h_pos_rgb), ptr(h_nrm_rgba), w, h, ptr(h_rgba)))

    def frame_enqueue_host_g16(self, h_nrm_rgba16, h_pos_rgb16f, w, h, h_rgba):
        """rt_hip_frame_enqueue_g16: the same pipeline fed with the G-buffer in its native 16-bit formats (uint16 x 4 UNORM
        normals with alpha = material id + 1, IEEE half x 3 positions as uint16 / float16 arrays)."""
        ptr = lambda a: C.c_void_p(a.data_ptr() if hasattr(a, "data_ptr") else a.ctypes.data)
        self._ck(self.lib.rt_hip_frame_enqueue_g16(self.h, ptr(h_nrm_rgba16), ptr(h_pos_rgb16f), w, h, ptr(h_rgba)))

    def frame_enqueue_device(self, d_pos_rgb, d_nrm_rgba, w, h, d_rgba):
        """Queue a whole frame without waiting for it (at most max(2, tuning.frame_streams) unfinished); frame_finish() completes the oldest."""
        self._ck(self.lib.rt_hip_frame_enqueue_device(self.h, _dptr(d_pos_rgb), _dptr(d_nrm_rgba), w, h, _dptr(d_rgba)))

    def frame_finish(self):
        rays = C.c_int64(0)
        self._ck(self.lib.rt_hip_frame_finish(self.h, C.byref(rays)))
        return rays.value

    def render_tiles_device(self, d_pos_rgb, d_nrm_rgba, w, h, tile, rank, nranks, d_rgba):
        self._ck(self.lib.rt_hip_render_tiles_device(self.h, _dptr(d_pos_rgb), _dptr(d_nrm_rgba), w, h, tile, rank, nranks,
                                                     _dptr(d_rgba)))

    def render_tiles_batch_device(self, d_pos_list, d_nrm_list, w, h, tile, rank, nranks, d_rgba_list):
        n = len(d_pos_list)
        arr = C.c_void_p * n
        self._ck(self.lib.rt_hip_render_tiles_batch_device(self.h, arr(*[_dptr(t) for t in d_pos_list]),
                                                           arr(*[_dptr(t) for t in d_nrm_list]), n, w, h, tile, rank, nranks,
                                                           arr(*[_dptr(t) for t in d_rgba_list])))

    def live_frame_device(self, d_pos_rgb, d_nrm_rgba, w, h, interlace, phase, d_raypass_rgb8, d_rgba):
        rays = C.c_int(0)
        self._ck(self.lib.rt_hip_live_frame_device(self.h, _dptr(d_pos_rgb), _dptr(d_nrm_rgba), w, h, interlace, phase,
                                                   _dptr(d_raypass_rgb8), _dptr(d_rgba), C.byref(rays)))
        return rays.value

    def live_frame_scaled_device(self, d_pre_pos, d_pre_nrm, pw, ph, d_pos_rgb, d_nrm_rgba, w, h, fov, interlace, phase, d_raypass_rgb8, d_rgba):
        """rt_hip_live_frame_scaled_device: prepass (pw x ph) and frame (w x h) of different sizes (ras_rasterizer.cpp:39,167-168)."""
        rays = C.c_int(0)
        self._ck(self.lib.rt_hip_live_frame_scaled_device(self.h, _dptr(d_pre_pos), _dptr(d_pre_nrm), pw, ph, _dptr(d_pos_rgb), _dptr(d_nrm_rgba),
                                                          w, h, float(fov), interlace, phase, _dptr(d_raypass_rgb8), _dptr(d_rgba), C.byref(rays)))
        return rays.value

    def render_tiles_packed_device(self, d_pos_list, d_nrm_list, w, h, tile, rank, nranks, d_packed_list):
        n = len(d_pos_list)
        arr = C.c_void_p * n
        self._ck(self.lib.rt_hip_render_tiles_packed_device(self.h, arr(*[_dptr(t) for t in d_pos_list]),
                                                            arr(*[_dptr(t) for t in d_nrm_list]), n, w, h, tile, rank, nranks,
                                                            arr(*[_dptr(t) for t in d_packed_list])))

    def unpack_chunks_device(self, d_chunks, chunk_bytes, w, h, tile, nranks, d_frame):
        self._ck(self.lib.rt_hip_unpack_chunks_device(self.h, _dptr(d_chunks), chunk_bytes, w, h, tile, nranks, _dptr(d_frame)))

    def pack_tiles_device(self, d_frame, w, h, tile, d_tile_list, ntiles, d_packed):
        self._ck(self.lib.rt_hip_pack_tiles_device(self.h, _dptr(d_frame), w, h, tile, _dptr(d_tile_list), ntiles, _dptr(d_packed)))

    def unpack_tiles_device(self, d_packed, w, h, tile, d_tile_list, ntiles, d_frame):
        self._ck(self.lib.rt_hip_unpack_tiles_device(self.h, _dptr(d_packed), w, h, tile, _dptr(d_tile_list), ntiles, _dptr(d_frame)))

    def render_frame(self, pos_rgb, nrm_rgba, w, h):
        pos_rgb, nrm_rgba = _f(pos_rgb), _f(nrm_rgba)
        out = np.zeros((h * w, 4), np.uint8)
        self._ck(self.lib.rt_hip_render_frame(self.h, _p(pos_rgb), _p(nrm_rgba), w, h, _p(out, c_u8_p)))
        return out

    def last_frame_rays(self):
        return int(self.lib.rt_hip_last_frame_rays(self.h))

    # ---- multi-GPU behind the C ABI (RCCL called by the library) ----------------------------------
    def comm_init(self, rank, nranks, unique_id=None, broadcast=None):
        """rt_hip_comm_init.  unique_id: the 128 bytes rank 0 got from comm_unique_id(); or pass `broadcast`, a callable
        that takes rank 0's bytes (None on other ranks) and returns them on every rank (e.g. over torch.distributed)."""
        if unique_id is None:
            mine = comm_unique_id() if rank == 0 else None
            unique_id = broadcast(mine) if broadcast else mine
        buf = (C.c_char * COMM_ID_BYTES).from_buffer_copy(bytes(unique_id))
        self._ck(self.lib.rt_hip_comm_init(self.h, buf, rank, nranks))

    def comm_destroy(self):
        self._ck(self.lib.rt_hip_comm_destroy(self.h))

    def render_frame_multi_device(self, d_pos_rgb, d_nrm_rgba, w, h, tile, d_rgba):
        self._ck(self.lib.rt_hip_render_frame_multi_device(self.h, _dptr(d_pos_rgb), _dptr(d_nrm_rgba), w, h, tile, _dptr(d_rgba)))

    def frame_multi_enqueue_device(self, d_pos_rgb, d_nrm_rgba, w, h, tile, d_rgba):
        """rt_hip_frame_multi_enqueue_device: the split frame queued without waiting; frame_finish() completes the oldest."""
        self._ck(self.lib.rt_hip_frame_multi_enqueue_device(self.h, _dptr(d_pos_rgb), _dptr(d_nrm_rgba), w, h, tile, _dptr(d_rgba)))

    def frames_multi_enqueue_device(self, d_pos_list, d_nrm_list, w, h, tile, d_rgba_list):
        """rt_hip_frames_multi_enqueue_device: several split frames as ONE pipeline entry (one pass of launches, one all-gather)."""
        n = len(d_pos_list)
        arr = C.c_void_p * n
        self._ck(self.lib.rt_hip_frames_multi_enqueue_device(self.h, arr(*[_dptr(t) for t in d_pos_list]), arr(*[_dptr(t) for t in d_nrm_list]),
                                                             n, w, h, tile, arr(*[_dptr(t) for t in d_rgba_list])))

    def tiles_packed_enqueue_device(self, d_pos_rgb, d_nrm_rgba, w, h, tile, rank, nranks, d_packed):
        """rt_hip_tiles_packed_enqueue_device: one frame's packed share queued without waiting; frame_finish() completes the oldest."""
        self._ck(self.lib.rt_hip_tiles_packed_enqueue_device(self.h, _dptr(d_pos_rgb), _dptr(d_nrm_rgba), w, h, tile, rank, nranks, _dptr(d_packed)))

    def tiles_packed_batch_enqueue_device(self, d_pos_list, d_nrm_list, w, h, tile, rank, nranks, d_packed_list):
        """rt_hip_tiles_packed_batch_enqueue_device: several frames' packed shares as one pipeline entry."""
        n = len(d_pos_list)
        arr = C.c_void_p * n
        self._ck(self.lib.rt_hip_tiles_packed_batch_enqueue_device(self.h, arr(*[_dptr(t) for t in d_pos_list]), arr(*[_dptr(t) for t in d_nrm_list]),
                                                                   n, w, h, tile, rank, nranks, arr(*[_dptr(t) for t in d_packed_list])))

    def photon_generate_multi(self, per_light):
        stored = C.c_int64(0)
        self._ck(self.lib.rt_hip_photon_generate_multi(self.h, per_light, C.byref(stored)))
        return stored.value

    # ---- photons --------------------------------------------------------------------------
    def photon_emit_device(self, per_light, first, count, d_out, capacity):
        stored = C.c_int64(0)
        self._ck(self.lib.rt_hip_photon_emit_device(self.h, per_light, first, count, _dptr(d_out), capacity, C.byref(stored)))
        return stored.value

    def photon_generate(self, per_light):
        """PhotonMap::generate on one GPU: emit, keep, build the grid; returns the number of photons stored."""
        stored = C.c_int64(0)
        self._ck(self.lib.rt_hip_photon_generate(self.h, per_light, C.byref(stored)))
        return stored.value

    def photon_get(self):
        """The installed photons in photon-index order, (n, 9) f32."""
        n = C.c_int64(0)
        self._ck(self.lib.rt_hip_photon_get(self.h, None, 0, C.byref(n)))
        out = np.zeros((n.value, 9), np.float32)
        if n.value:
            self._ck(self.lib.rt_hip_photon_get(self.h, _p(out), n.value, C.byref(n)))
        return out

    def photon_set_device(self, d_photons, n):
        self._ck(self.lib.rt_hip_photon_set_device(self.h, _dptr(d_photons), n))

    def photon_set(self, photons9):
        photons9 = _f(photons9)
        self._ck(self.lib.rt_hip_photon_set(self.h, _p(photons9), photons9.shape[0]))

    def radiance_estimate(self, pos, nrm, count=100, radius=1.0):
        pos, nrm = _f(pos), _f(nrm)
        out = np.zeros_like(pos)
        self._ck(self.lib.rt_hip_radiance_estimate(self.h, _p(pos), _p(nrm), pos.shape[0], count, radius, _p(out)))
        return out

    def photon_lookup(self, pos, radius=1.0):
        """PhotonMap::lookup at each point: summed energy of the photons within `radius`."""
        pos = _f(pos)
        out = np.zeros_like(pos)
        self._ck(self.lib.rt_hip_photon_lookup(self.h, _p(pos), pos.shape[0], radius, _p(out)))
        return out

    def photon_lookup_device(self, d_pos, npts, radius, d_out):
        self._ck(self.lib.rt_hip_photon_lookup_device(self.h, _dptr(d_pos), npts, radius, _dptr(d_out)))

    def radiance_estimate_device(self, d_pos, d_nrm, npts, count, radius, d_out):
        self._ck(self.lib.rt_hip_radiance_estimate_device(self.h, _dptr(d_pos), _dptr(d_nrm), npts, count, radius, _dptr(d_out)))

    # ---- live shading of hit records (mesh_fs.cpp:26-49 as drawRayData applies it) ----------
    def shade_results(self, records, offsets, want_rgb8=True):
        records = np.ascontiguousarray(records, RAY_DTYPE)
        offsets = np.ascontiguousarray(offsets, OFFSET_DTYPE)
        n = records.shape[0]
        rgb = np.zeros((n, 3), np.float32)
        rgb8 = np.zeros((n, 3), np.uint8) if want_rgb8 else None
        self._ck(self.lib.rt_hip_shade_results(self.h, records.ctypes.data_as(C.c_void_p), offsets.ctypes.data_as(C.c_void_p),
                                               n, offsets.shape[0], _p(rgb),
                                               rgb8.ctypes.data_as(C.c_void_p) if want_rgb8 else None))
        return (rgb, rgb8) if want_rgb8 else rgb

    def shade_results_device(self, d_records, offsets, n, d_rgb, d_rgb8=None):
        offsets = np.ascontiguousarray(offsets, OFFSET_DTYPE)
        self._ck(self.lib.rt_hip_shade_results_device(self.h, _dptr(d_records), offsets.ctypes.data_as(C.c_void_p), n,
                                                      offsets.shape[0], _dptr(d_rgb), _dptr(d_rgb8) if d_rgb8 is not None else None))

    # ---- introspection --------------------------------------------------------------------
    def stats(self):
        s = Stats()
        self._ck(self.lib.rt_hip_get_stats(self.h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in Stats._fields_}

    def reset_stats(self):
        self._ck(self.lib.rt_hip_reset_stats(self.h))

    def sync(self):
        self._ck(self.lib.rt_hip_sync(self.h))

    def stream(self):
        return self.lib.rt_hip_stream(self.h)

    def get_bvh(self):
        st = self.stats()
        nn, nt = int(st["num_nodes"]), int(st["num_tris"])
        nodes = np.zeros((nn, 16), np.float32)
        prims = np.zeros(nt, np.uint32)
        self._ck(self.lib.rt_hip_get_bvh(self.h, _p(nodes), nn, _p(prims, c_u32_p), nt))
        return nodes, prims


# every entry that takes device pointers first waits for torch's pending work on those buffers
for _name in [n for n in dir(RtHip) if n.endswith("_device")]:
    setattr(RtHip, _name, _fenced(getattr(RtHip, _name)))
