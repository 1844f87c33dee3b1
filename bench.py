#!/usr/bin/env python3
"""bench.py -- secondary-ray throughput of the HIP hot path on the BASELINE.json workload.

    python bench.py --gpus N --steps K --warmup W          (N>1 is launched by torch.distributed.run)

Workload (config.workload): "atrium-262k-1080p" = BASELINE.json configs[2], the configuration the
metric ("secondary Mrays/s + ms/frame @1080p, 1/2/4/8 MI355X") is quoted on: a ~262k-triangle
procedural atrium (real Sponza cannot be fetched offline), GPU LBVH, 1080p G-buffer resident in
HBM, Gen-1 shading with shadow + reflection + refraction rays, 2 bounces.

A step = one frame's worth of texels per GPU.  At N GPUs the job renders N frames (N fields of
view) per step and every frame is split over the ranks by interleaved 64x64 tiles, so each rank
traces one frame's worth of texels (weak scaling: per-GPU work is fixed) -- its tiles of all N
frames as one batch, so the launches stay frame-sized.  The scene and the LBVH
are replicated; the only collective is the gather of the finished tiles, inside the timed region: every
rank packs its tiles of frame f, one RCCL all-to-all sends them to rank f, which assembles frame f
(raytracer_amd/frames.py), so only owned pixels travel.

value = secondary rays issued by all ranks / wall time (barrier + synchronize on both sides, max
over ranks).  roofline = the dominant kernel (the shadow any-hit traversal k_shadow) against HBM:
algorithmic bytes = rays x (32 B x nodes/ray + 48 B x tris/ray + 48 B ray in/out), with nodes/ray
and tris/ray counted on the device by a census pass over the identical frame.  cpu_baseline = the
compiled reference's own RayTracer::processRays (oracle/_ref, kind "reference") or the oracle's
restatement (kind "port") on a bounded sample of the same G-buffer, single thread as the
reference is.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
NODE_FLOP, TRI_FLOP = 34.0, 45.0   # per node visit: 12 FMA (24) + 8 min/max + 2 compares; per triangle test: Moeller-Trumbore up to t
NODE_BYTES = 32.0  # the traversal reads 32-byte quantised nodes (rt_dev.h: QNode); SURVEY 8d prices the float node at 64
VALU_PEAK_TFLOPS = 78.6  # per-lane v_fma_f32 rate (157 TFLOP/s needs packed/dual-issue FP32 throughout)


def pmc_traffic(workload, shadow_mode, accel):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 counter passes of this
    same command (profiles/r1_pmc_traffic.json, made by scripts/profile_round.sh); None for any other
    configuration -- counters cannot be read from inside the process being measured."""
    path = os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")
    if not (workload == "atrium-262k-1080p" and shadow_mode == "corrected" and accel == "ploc" and os.path.exists(path)):
        return None
    with open(path) as f:
        return float(json.load(f)["traffic_bytes_per_launch"])


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="atrium-262k-1080p",
                    choices=["atrium-262k-1080p", "atrium-1m-1080p", "atrium-1m-4k", "cornell-1080p", "atrium-8k-small"])
    ap.add_argument("--shadow-mode", default="corrected", choices=["corrected", "literal"],
                    help="corrected: a hit shadows only before the light (shadow_t_max=1); literal: the reference's any-hit")
    ap.add_argument("--accel", default="ploc", choices=["lbvh", "ploc", "sah"], help="BVH builder (lbvh, ploc: on the GPU; sah: one host thread, static scenes)")
    ap.add_argument("--photons", type=int, default=0,
                    help="C5: photons per light (0 = off).  The sharded photon pass (emit -> RCCL all-gather -> grid) runs once, "
                         "untimed by `value` and reported as config.photon_pass_ms; frames then shade with radiance_estimate "
                         "instead of shadow rays")
    ap.add_argument("--inflight", type=int, default=1,
                    help="N = 1 only: frames in flight = contexts (each its own stream) fed from as many host threads; the "
                         "default 1 keeps the kernels of a frame alone on the GPU, which is what the roofline object measures")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="G-buffer texels fed to the CPU baseline (0 = auto)")
    return ap.parse_args()


WORKLOADS = {
    "atrium-262k-1080p": dict(tris=262144, w=1920, h=1080),
    "atrium-1m-1080p": dict(tris=1048576, w=1920, h=1080),
    "atrium-1m-4k": dict(tris=1048576, w=3840, h=2160),
    "cornell-1080p": dict(tris=0, w=1920, h=1080),
    "atrium-8k-small": dict(tris=8192, w=480, h=270),
}


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))

    import torch
    import torch.distributed as dist
    import raytracer_amd as rt
    from raytracer_amd import scenes

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # BENCH_DIST_BACKEND=gloo rehearses the N > 1 code path on a one-GPU box (all ranks share device 0 and the tile
    # exchange is staged through the host); the driver's multi-GPU runs use the default, nccl = RCCL over xGMI.
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    red_dev = dev if backend == "nccl" else torch.device("cpu")   # where the scalar reductions live

    wl = WORKLOADS[args.workload]
    W, H = wl["w"], wl["h"]
    ACCEL = {"lbvh": rt.ACCEL_LBVH, "ploc": rt.ACCEL_PLOC, "sah": rt.ACCEL_SAH}
    scene = scenes.cornell() if wl["tris"] == 0 else scenes.atrium(wl["tris"])
    shadow_t_max = 1.0 if args.shadow_mode == "corrected" else 0.0
    g = rt.RtHip(scene, semantics="array", accel=ACCEL[args.accel], device=dev_index, profile=1, bounces=2, do_shadows=1,
                 shadow_t_max=shadow_t_max, use_photon_map=1 if args.photons > 0 else 0, photon_estimate=100,
                 photon_radius=1.0, seed=42)
    build_stats = g.stats()
    photon_info = None
    if args.photons > 0:
        from raytracer_amd.photons import build_photon_map
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        nph = build_photon_map(g, args.photons, len(scene.lights), dev)
        g.sync()
        photon_info = {"photons_per_light": args.photons, "photons_stored": nph,
                       "photon_pass_ms": round((time.perf_counter() - t0) * 1e3, 3)}

    # ---- inputs resident in HBM: one G-buffer per frame of the step -------------------------
    nframes = world
    npix = W * H
    gb_n, gb_p = [], []
    for f in range(nframes):
        d_n = torch.zeros(npix * 4, dtype=torch.float32, device=dev)
        d_p = torch.zeros(npix * 3, dtype=torch.float32, device=dev)
        g.render_gbuffer_device(scene.fov - 0.03 * f, W, H, d_n, d_p)
        gb_n.append(d_n)
        gb_p.append(d_p)
    g.sync()
    if os.environ.get("BENCH_DEBUG"):
        cov = [int((n.view(-1, 4)[:, 3] > 0).sum().item()) for n in gb_n]
        print("rank %d: build %s, covered texels per frame %s" % (rank, {k: build_stats[k] for k in ("num_tris", "num_nodes", "bvh_depth", "build_ms")}, cov),
              file=sys.stderr, flush=True)
    tiles = [torch.zeros(npix * 4, dtype=torch.uint8, device=dev) for _ in range(nframes)]   # this rank's tiles of each frame
    frame = torch.zeros(npix * 4, dtype=torch.uint8, device=dev)    # N > 1: frame `rank`, assembled from all ranks' tiles
    xchg = None
    if world > 1:
        from raytracer_amd.frames import TileExchange
        xchg = TileExchange(g, W, H, 64, rank, world, dev)

    render_batch = rt.RtHip.render_tiles_batch_device.__wrapped__

    def step():
        # this rank's tiles of all the step's frames in one wavefront (a batch of one at N = 1)
        if xchg and os.environ.get("BENCH_XCHG") == "pack":   # debugging aid: full-size tile images, then pack kernels
            g.render_tiles_batch_device(gb_p, gb_n, W, H, 64, rank, world, tiles)
            rays = g.last_frame_rays()
            for f in range(nframes):
                xchg.pack(f, tiles[f])
            xchg.exchange(frame)
        elif xchg and os.environ.get("BENCH_XCHG") == "sync":   # collective between the steps instead of under the next one
            xchg.render(gb_p, gb_n)
            rays = g.last_frame_rays()
            xchg.exchange(frame)
        elif xchg:
            xchg.render(gb_p, gb_n)       # ... written straight into the packed chunks: chunk f goes to rank f
            rays = g.last_frame_rays()
            xchg.finish(frame)            # the PREVIOUS step's all-to-all ran under this step's kernels: unpack -> its frame `rank` is complete
            xchg.begin()                  # this step's all-to-all (RCCL, own stream); finished during the next step, or by drain()
        else:
            # the binding's un-fenced entry: the G-buffers were complete before the loop and the call itself
            # returns only after its last kernel, so the per-call torch-stream drain has nothing to wait for
            render_batch(g, gb_p, gb_n, W, H, 64, rank, world, tiles)
            rays = g.last_frame_rays()
        return rays

    def drain():
        if xchg:
            xchg.finish(frame)            # the last step's exchange belongs to the timed region

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # --inflight M: M - 1 further contexts (scene and BVH replicated, own stream, own output), one host thread each;
    # step k goes to context k % M.  The frames overlap on the GPU: the tail of one frame's kernels and the host's
    # per-frame work disappear under the other frame's kernels.
    inflight = max(1, args.inflight) if world == 1 and args.photons == 0 else 1
    extra = []
    for _ in range(inflight - 1):
        ge = rt.RtHip(scene, semantics="array", accel=ACCEL[args.accel], device=dev_index,
                      profile=0, bounces=2, do_shadows=1, shadow_t_max=shadow_t_max, seed=42)
        extra.append((ge, [torch.zeros(npix * 4, dtype=torch.uint8, device=dev)]))

    def run_steps(count):
        if not extra:
            return sum(step() for _ in range(count))
        import threading
        rays = [0] * inflight

        def worker(t):
            ctx, out = (g, tiles) if t == 0 else extra[t - 1]
            for _ in range(t, count, inflight):
                render_batch(ctx, gb_p, gb_n, W, H, 64, rank, world, out)
                rays[t] += ctx.last_frame_rays()
        th = [threading.Thread(target=worker, args=(t,)) for t in range(inflight)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        return sum(rays)

    run_steps(args.warmup)
    drain()
    fence()
    g.reset_stats()
    t0 = time.perf_counter()
    rays_rank = run_steps(args.steps)
    drain()
    fence()
    elapsed = time.perf_counter() - t0
    st = g.stats()
    if os.environ.get("BENCH_DEBUG"):
        print("rank %d: %d rays in %d steps, %.3f s; any %.3f ms / %d launches" % (rank, rays_rank, args.steps, elapsed, st["any_ms"], st["any_launches"]),
              file=sys.stderr, flush=True)

    el = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    ry = torch.tensor([float(rays_rank)], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(ry, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    total_rays = float(ry.item())

    # ---- roofline of the dominant kernel (rank 0's launches) --------------------------------
    any_ms, any_launches, rays_any = st["any_ms"], st["any_launches"], st["rays_any"]
    clo_ms, clo_launches, rays_clo = st["closest_ms"], st["closest_launches"], st["rays_closest"]
    # census pass: same frame(s), counting node visits / triangle tests per ray (untimed)
    g.set_config(census=1, profile=0)
    g.reset_stats()
    g.render_tiles_batch_device(gb_p, gb_n, W, H, 64, rank, world, tiles)
    cs = g.stats()
    g.set_config(census=0, profile=1)
    nodes_any = cs["census_nodes_any"] / max(cs["census_rays_any"], 1)
    tris_any = cs["census_tris_any"] / max(cs["census_rays_any"], 1)
    nodes_clo = cs["census_nodes_closest"] / max(cs["census_rays_closest"], 1)
    tris_clo = cs["census_tris_closest"] / max(cs["census_rays_closest"], 1)
    bytes_per_any = NODE_BYTES * nodes_any + 48.0 * tris_any + 48.0
    bytes_per_clo = NODE_BYTES * nodes_clo + 48.0 * tris_clo + 48.0
    fused = clo_launches == 0 and rays_clo > 0   # the frame driver folds the closest-hit queues into the shadow launches
    if fused:
        dominant = "k_shadow (shadow any-hit rays + the level's closest-hit queue in the same launch)"
        launches = max(any_launches, 1)
        rays_per_launch = (rays_any + rays_clo) / launches
        avg_ms = any_ms / launches
        bytes_per_launch = (rays_any * bytes_per_any + rays_clo * bytes_per_clo) / launches
        bpr = bytes_per_launch / max(rays_per_launch, 1)
        nodes_dom = (rays_any * nodes_any + rays_clo * nodes_clo) / max(rays_any + rays_clo, 1)
        tris_dom = (rays_any * tris_any + rays_clo * tris_clo) / max(rays_any + rays_clo, 1)
    elif any_ms >= clo_ms:
        dominant = "k_shadow (any-hit)"
        rays_per_launch = rays_any / max(any_launches, 1)
        avg_ms = any_ms / max(any_launches, 1)
        bpr, nodes_dom, tris_dom = bytes_per_any, nodes_any, tris_any
    else:
        dominant = "k_bvh (closest-hit)"
        rays_per_launch = rays_clo / max(clo_launches, 1)
        avg_ms = clo_ms / max(clo_launches, 1)
        bpr, nodes_dom, tris_dom = bytes_per_clo, nodes_clo, tris_clo
    achieved = (rays_per_launch * bpr) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": pmc_traffic(args.workload, args.shadow_mode, args.accel) if (world == 1 and args.photons == 0) else None,
                "traffic_note": "HBM-side bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, profiles/r1_pmc_traffic.json); the "
                                "algorithmic bytes are served by L1/L2/Infinity Cache, so achieved > peak is expected and HBM "
                                "is not the binding limit: see `valu`",
                "valu": {"unit": "TFLOP/s", "flop_per_ray": round(NODE_FLOP * nodes_dom + TRI_FLOP * tris_dom, 1),
                         "achieved": round(rays_per_launch * (NODE_FLOP * nodes_dom + TRI_FLOP * tris_dom) / (avg_ms * 1e-3) / 1e12, 2) if avg_ms > 0 else 0.0,
                         "peak": VALU_PEAK_TFLOPS,
                         "frac": round(rays_per_launch * (NODE_FLOP * nodes_dom + TRI_FLOP * tris_dom) / (avg_ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS, 4) if avg_ms > 0 else 0.0},
                "kernel": dominant, "avg_launch_ms": round(avg_ms, 4), "rays_per_launch": round(rays_per_launch, 1),
                "nodes_per_ray": round(nodes_dom, 2), "tris_per_ray": round(tris_dom, 2),
                "bytes_per_ray": round(bpr, 1),
                "kernel_mrays_per_s": round(rays_per_launch / (avg_ms * 1e-3) / 1e6, 1) if avg_ms > 0 else 0.0,
                "any_hit": {"rays_per_frame": round(rays_any / max(st["frames"], 1), 1), "nodes_per_ray": round(nodes_any, 2),
                            "tris_per_ray": round(tris_any, 2)},
                "closest": {"rays_per_frame": round(rays_clo / max(st["frames"], 1), 1), "nodes_per_ray": round(nodes_clo, 2),
                            "tris_per_ray": round(tris_clo, 2), "standalone_launches": clo_launches,
                            "avg_launch_ms": round(clo_ms / max(clo_launches, 1), 4)}}

    # ---- CPU baseline on rank 0 (N = 1 only) ------------------------------------------------
    # ---- N > 1 (untimed): the frame assembled from all ranks' tiles == the same frame shaded whole on this rank ----
    tiles_verified = None
    if xchg:
        whole = torch.zeros(npix * 4, dtype=torch.uint8, device=dev)
        g.render_frame_device(gb_p[rank], gb_n[rank], W, H, whole)
        ok = torch.tensor([1.0 if torch.equal(whole, frame) else 0.0], dtype=torch.float64, device=red_dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        tiles_verified = bool(ok.item() == 1.0)

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = run_cpu_baseline(scene, gb_n[0], gb_p[0], W, H, args.cpu_sample)

    if rank == 0:
        out = {
            "metric": "secondary Mrays/s", "value": round(total_rays / elapsed / 1e6, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload + ("+photons" if args.photons > 0 else ""), "scene": scene.name, "triangles": scene.num_tris, "width": W, "height": H,
                       "frames_per_step": nframes, "tile": 64, "bounces": 2, "lights": len(scene.lights),
                       "shadows": args.shadow_mode, "accel": args.accel, "parallelism": "tiles x%d" % world,
                       **({"frames_in_flight": inflight} if inflight > 1 else {}),
                       "rays_per_frame": round(total_rays / args.steps / nframes, 1),
                       "ms_per_frame": round(elapsed / args.steps * 1e3, 3),
                       "bvh_build_ms": round(build_stats["build_ms"], 3), "bvh_depth": build_stats["bvh_depth"],
                       **({"exchange": "all-to-all of packed tiles, overlapped with the next step" if os.environ.get("BENCH_XCHG") is None else os.environ["BENCH_XCHG"],
                           "assembled_frames_equal_whole_frames": tiles_verified} if xchg else {}),
                       **({"photon_map": photon_info} if photon_info else {})},
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(out))
    g.close()
    if world > 1:
        dist.destroy_process_group()


def run_cpu_baseline(scene, d_n, d_p, W, H, sample):
    """RayTracer::processRays of the compiled reference (or the oracle port) on a bounded sample."""
    from oracle import pyoracle
    nr = d_n.cpu().numpy().reshape(-1, 4)
    po = d_p.cpu().numpy().reshape(-1, 3)
    sel = np.nonzero(nr[:, 3] > 0)[0]
    kind = "reference" if pyoracle.have_ref() else "port"
    cpu = pyoracle.Ref(scene) if kind == "reference" else pyoracle.Oracle(scene)
    nrm = (nr[:, :3].astype(np.float64) * 2.0 - 1.0).astype(np.float32)
    # calibrate on 1/64 of the texels, then size the sample for about 15 s
    probe = sel[::64]
    t0 = time.perf_counter()
    cpu.process_rays(po[probe], nrm[probe])
    dt = time.perf_counter() - t0
    if sample <= 0:
        rate = len(probe) / max(dt, 1e-6)
        sample = int(min(len(sel), max(len(probe), rate * 15.0)))
    stride = max(len(sel) // sample, 1)
    pick = sel[::stride]
    t0 = time.perf_counter()
    r = cpu.process_rays(po[pick], nrm[pick])
    dt = time.perf_counter() - t0
    out = {"value": round(len(pick) / dt / 1e6, 4), "unit": "Mrays/s", "cores": 1, "kind": kind,
           "sample": "RayTracer::processRays (DACRT, reflection rays, closest hit) on %d of the %d covered 1080p "
                     "G-buffer texels (every %dth), %.1f s, %d hits" % (len(pick), len(sel), stride, dt, len(r["records"]))}
    # The reference is single-threaded and non-re-entrant (file-scope DACRT arrays, rt_raytracer.cpp:27-48).  The
    # all-cores column of SURVEY 8d is the oracle port, one scene object and one contiguous slice of the sample per
    # thread (ctypes releases the GIL for the call); no omp.h on this image, so plain threads.
    try:
        from concurrent.futures import ThreadPoolExecutor
        cores = max(1, min(16, os.cpu_count() or 1))     # the GPU box gives one job 16 CPUs
        workers = [pyoracle.Oracle(scene) for _ in range(cores)]
        n_mt = int(min(len(sel), len(pick) * max(cores // 2, 1)))
        pick_mt = sel[::max(len(sel) // n_mt, 1)]
        slices = np.array_split(pick_mt, cores)
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            hits = list(ex.map(lambda a: len(a[0].process_rays(po[a[1]], nrm[a[1]])["records"]), zip(workers, slices)))
        dt_mt = time.perf_counter() - t0
        out["all_cores"] = {"value": round(len(pick_mt) / dt_mt / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
                            "sample": "the oracle's processRays restatement, %d texels in %d contiguous slices, one thread "
                                      "each, %.1f s, %d hits" % (len(pick_mt), cores, dt_mt, sum(hits))}
    except Exception as e:                                # the single-core column stands on its own
        out["all_cores"] = {"error": str(e)}
    return out


if __name__ == "__main__":
    main()
