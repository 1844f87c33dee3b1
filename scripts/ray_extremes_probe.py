#!/usr/bin/env python3
"""BVH vs brute force on extreme rays: zero direction, tiny / huge magnitudes, origins far away, NaN / inf components."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
from raytracer_amd import scenes
sc = scenes.atrium(2048)
rng = np.random.RandomState(1)
n = 4096
o = rng.uniform(-5, 5, (n, 3)).astype(np.float32); o[:, 2] -= 30
d = rng.normal(size=(n, 3)).astype(np.float32)
groups = {}
def add(name, oo, dd):
    groups[name] = np.concatenate([oo, dd], 1).astype(np.float32)
add("zero-dir", o, np.zeros_like(d))
add("tiny-dir", o, d * np.float32(1e-30))
add("denormal-dir", o, d * np.float32(1e-41))
add("huge-dir", o, d * np.float32(1e30))
add("far-origin", o * np.float32(1e4), -o)
dn = d.copy(); dn[::3, 0] = np.nan
add("nan-dir", o, dn)
di = d.copy(); di[::3, 1] = np.inf
add("inf-dir", o, di)
on = o.copy(); on[::3, 2] = np.nan
add("nan-origin", on, d)
ref = rt.RtHip(sc, semantics="array", accel=1)
g = rt.RtHip(sc, semantics="array", accel=3)
for name, rays in groups.items():
    a = ref.intersect(rays); b = g.intersect(rays)
    same_hit = np.array_equal(a["hit"], b["hit"])
    hit = (a["hit"] > 0) & (b["hit"] > 0)
    same_prim = np.array_equal(a["hits"]["prim"][hit], b["hits"]["prim"][hit])
    same_t = np.array_equal(a["hits"]["t"][hit].view(np.uint32), b["hits"]["t"][hit].view(np.uint32))
    print(name, "hits brute", int(a["hit"].sum()), "bvh", int(b["hit"].sum()), "hit==", same_hit, "prim==", same_prim, "t==", same_t)
# detail for the far-origin group
rays = groups["far-origin"]
a = ref.intersect(rays); b = g.intersect(rays)
hit = (a["hit"] > 0) & (b["hit"] > 0)
pa, pb = a["hits"]["prim"][hit].astype(np.int64), b["hits"]["prim"][hit].astype(np.int64)
bad = pa != pb
print("far-origin prim mismatches", int(bad.sum()), "of", int(hit.sum()), "bvh id higher", int((pb[bad] > pa[bad]).sum()), "lower", int((pb[bad] < pa[bad]).sum()))
ta, tb = a["hits"]["t"][hit], b["hits"]["t"][hit]
print("t equal where prim differs", bool(np.array_equal(ta[bad], tb[bad])), "sample", pa[bad][:5], pb[bad][:5], ta[bad][:5])
for acc in (2, 4):
    h = rt.RtHip(sc, semantics="array", accel=acc)
    c = h.intersect(rays)
    pc = c["hits"]["prim"][hit].astype(np.int64)
    print("accel", acc, "mismatches vs brute", int((pc != pa).sum()), "vs ploc", int((pc != pb).sum()))
