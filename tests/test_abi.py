"""The C-ABI library: every function include/rt_hip.h declares is exported, the POD layouts match
the reference's measured sizes, and -- with no GPU -- the library refuses to create a context
instead of falling back to anything."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import raytracer_amd as rt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "rt_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_hip_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    assert os.path.exists(rt.LIB_PATH), "build librt_hip.so first (__graft_entry__.build())"
    out = subprocess.check_output(["nm", "-D", "--defined-only", rt.LIB_PATH]).decode()
    exported = set(l.split()[-1] for l in out.splitlines() if " T " in l)
    decl = declared_functions()
    assert len(decl) >= 30
    missing = [f for f in decl if f not in exported]
    assert not missing, "declared in include/rt_hip.h but not exported: %s" % missing
    assert sorted(rt.ABI_SYMBOLS) == decl, "python binding list out of sync with the header"


def test_struct_layouts():
    assert rt.RAY_DTYPE.itemsize == 64          # data/data.h:34-42
    assert [rt.RAY_DTYPE.fields[k][1] for k in ("origin", "direction", "position", "normal", "texcoord", "pad")] == [0, 12, 24, 36, 48, 56]
    assert rt.MATERIAL_DTYPE.itemsize == 48     # data/data.h:18-26
    assert [rt.MATERIAL_DTYPE.fields[k][1] for k in ("diffuse_color", "specular_color", "shininess", "reflectivity", "alpha", "ior", "texture")] == [0, 12, 24, 28, 32, 36, 40]
    assert rt.LIGHT_DTYPE.itemsize == 28        # data/data.h:44-48
    assert rt.HIT_DTYPE.itemsize == 16 and rt.OFFSET_DTYPE.itemsize == 8


def test_library_loads_and_configs():
    L = rt.load_library()
    cfg = rt.Config()
    assert L.rt_hip_default_config(C.byref(cfg)) == 0
    assert cfg.bounces == 2 and cfg.do_shadows == 0 and cfg.photon_count == 10000 and cfg.photon_estimate == 100
    assert cfg.eps_det == np.finfo(np.float32).eps and np.isinf(cfg.t_max) and cfg.use_model_mat == 1
    assert L.rt_hip_array_config(C.byref(cfg)) == 0
    # the float threshold must reproduce `(double)x < 0.00001` exactly
    f = np.float32(cfg.eps_det)
    below = np.nextafter(f, np.float32(0))
    assert float(below) < 0.00001 <= float(f)
    assert cfg.t_max == 1000000.0 and cfg.use_model_mat == 0
    assert L.rt_hip_default_config(None) < 0 and b"null" in L.rt_hip_last_error()


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    L = rt.load_library()
    cfg = rt.Config()
    L.rt_hip_default_config(C.byref(cfg))
    h = C.c_void_p()
    assert L.rt_hip_create(C.byref(cfg), C.byref(h)) < 0
    assert not h.value
    with pytest.raises(rt.RtHipError):
        rt.RtHip()
