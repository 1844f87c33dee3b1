#!/usr/bin/env python3
"""bench.py -- secondary-ray throughput of the HIP hot path on the BASELINE.json workload.

    python bench.py --gpus N --steps K --warmup W          (N>1 is launched by torch.distributed.run)

Workload (config.workload): "atrium-262k-1080p" = BASELINE.json configs[2], the configuration the
metric ("secondary Mrays/s + ms/frame @1080p, 1/2/4/8 MI355X") is quoted on: a ~262k-triangle
procedural atrium (real Sponza cannot be fetched offline), GPU-built BVH, 1080p G-buffer resident in
HBM, Gen-1 shading with shadow + reflection + refraction rays, 2 bounces.

A step = `config.frames_per_step` frames (300 at N = 1), so that the timed region lasts about five seconds and the
driver's own GPU-busy sampling can see it.  Two ways to use N GPUs:
  --scaling strong (default) every sub-step renders ONE frame split over the N ranks by interleaved 64x64 tiles, through the
        C ABI alone (--route cabi: rt_hip_comm_init with the id broadcast over torch.distributed, then
        rt_hip_render_frame_multi_device: tiles shaded into the packed chunk, ncclAllGather, unpack -- the whole image
        on every rank before the next frame starts), so ms_per_frame is the latency of a frame.  Total work is fixed:
        north_star's "the frame is partitioned across the 8 GPUs".  --route torch: the same through frames.py.
  --scaling weak            every sub-step renders N frames (N fields of view); each frame is split over the ranks, so a
        rank traces one frame's worth of texels (its tiles of all N frames in ONE wavefront); one all-to-all per
        sub-step sends rank r's tiles of frame f to rank f, overlapped with the next sub-step's kernels.
config.ms_per_frame is always wall time / frames completed by the whole job (1 / throughput);
config.frame_latency_ms is the time from a frame's first kernel to its assembled image.

value = secondary rays issued by all ranks / wall time (barrier + synchronize on both sides, max over
ranks), inputs resident in HBM.  roofline = the kernel with the most GPU time among ALL timed kernels
of the frame (the fused shadow / closest-hit traversal k_shadow; with --photons the photon gather),
priced against the resource the counters show it is bound by -- the vector ALU -- with frac <= 1:
algorithmic FLOP (34 per BVH node visit, 45 per triangle test, counted on the device by a census pass
over the identical frame with every ray traced from the root by itself) / launch time against the 157.3 TFLOP/s FP32 vector peak; the HBM side
(counter bytes per launch, offline rocprofv3 passes of this same command under profiles/) is reported
beside it.  cpu_baseline = the compiled reference's own RayTracer::processRays (oracle/_ref, kind
"reference") or the oracle's restatement (kind "port") on a bounded sample of the same G-buffer,
single thread as the reference is.
"""
from __future__ import annotations

import argparse
import ctypes
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s
L2_PEAK_GBS = 34500.0            # MI355X_MICROARCH.md, L2 (per XCD): ~34.5 TB/s over the 8 XCDs
L1_PEAK_GBS = 39321.6            # 64 B/clk/CU x 256 CUs x 2.4 GHz
VALU_PEAK_TFLOPS = 157.3         # MI355X_MICROARCH.md: FP32 vector peak = 2 FLOP x 32 lanes/clk/SIMD x 1024 SIMDs x 2.4 GHz: a wave64 v_fma_f32
                                 # issues over 2 cycles when more than one wave feeds the SIMD (4 for a wave alone); plain per-lane FMA code can
                                 # reach it -- packed FP32 takes 4 cycles per instruction and adds nothing (scripts/probes/pk_rate.hip, measured)
NODE_FLOP, TRI_FLOP = 34.0, 45.0  # per node visit: 12 FMA (24) + 8 min/max + 2 compares for both child boxes; per triangle test: Moeller-Trumbore up to t
NODE_BYTES = 32.0                # the traversal reads 32-byte quantised nodes (rt_dev.h: QNode)
NODE_BYTES_SURVEY = 64.0         # SURVEY.md 8d prices a node visit at the 64-byte float node
GATHER_VISIT_FLOP = 10.0         # per photon visited: 3 sub + 5 (dot) + compare + count/max
GATHER_VISIT_BYTES = 16.0        # position record


def kernel_source_hash(photons=False):
    """Identity of the kernels a counter profile was taken with: sha256 over the sources of the traversal and frame
    kernels (and of the gather kernels for photon-mapped workloads) and their build flags."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "raytracer_amd", "csrc")
    for name in ("Makefile", "k_frame.hip", "k_trace.hip", "rt_dev.h", "rt_trace_dev.h") + (("k_photon.hip",) if photons else ()):
        with open(os.path.join(d, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def offline_counters(workload, shadow_mode, accel, photons):
    """The rocprofv3 --pmc passes of this same command (scripts/profile_round.sh writes the summary; counters
    cannot be read from inside the process being measured).  Returned only when the profile was taken on the
    same workload, builder and kernel sources; otherwise (None, why)."""
    for name in (("r3_pmc_traffic_c5.json", "r2_pmc_traffic_c5.json") if photons else ("r3_pmc_traffic.json", "r2_pmc_traffic.json", "r1_pmc_traffic.json")):
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        with open(path) as f:
            c = json.load(f)
        want = dict(workload=workload + ("+photons" if photons else ""), shadows=shadow_mode, accel=accel)
        have = dict(workload=c.get("workload"), shadows=c.get("shadows", "corrected"), accel=c.get("accel", "ploc"))
        if photons:   # the counter profile of a photon-mapped frame belongs to one photon count
            want["photons_per_light"] = int(photons); have["photons_per_light"] = c.get("photons_per_light")
        if want != have:
            return None, "profiles/%s is for %s, this run is %s" % (name, have, want)
        if c.get("kernel_source_hash") != kernel_source_hash(photons):
            return None, "profiles/%s was taken with other kernel sources (%s, now %s)" % (name, c.get("kernel_source_hash"), kernel_source_hash(photons))
        c["file"] = "profiles/" + name
        return c, "offline: rocprofv3 --pmc passes of this command, same workload / builder / kernel sources (%s)" % c["file"]
    return None, "no counter profile committed"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames-per-step", type=int, default=0,
                    help="sub-steps per step (0 = auto: 300, or 2 with --photons): sized so that the timed region of the default run lasts "
                         "more than 5 s (a driver-side GPU-busy sample 5 s apart lands in it)")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="strong (default): every sub-step is ONE frame, split over the N ranks (north_star's partition); weak: N frames per sub-step")
    ap.add_argument("--route", default="cabi", choices=["cabi", "torch"],
                    help="N > 1, strong: cabi = rt_hip_comm_init + rt_hip_render_frame_multi_device (RCCL called by the library, the route a C++ "
                         "host takes; the unique id is broadcast over torch.distributed); torch = tiles.py / frames.py over torch.distributed")
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
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="frames the library's pipeline keeps in flight on streams of their own (tuning.frame_streams, 1..4). 0 = auto: 1 at N = 1 "
                         "(launches that do not overlap, which is what the roofline object prices), 3 for a frame split over N > 1 GPUs "
                         "through the C ABI (rt_hip_frame_multi_enqueue_device)")
    ap.add_argument("--frames-per-entry", type=int, default=0,
                    help="N > 1 through the C ABI: frames per pipeline entry (rt_hip_frames_multi_enqueue_device: one pass of launches and one "
                         "all-gather for all of them, 1..8). 0 = auto: N / 2, at most 4 -- every launch then as large as half a frame's")
    ap.add_argument("--tuning", default="", help="rt_hip_tuning overrides, name=value,name=value (variant measurements; results never depend on them)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the untimed extra figures (host-to-host and with-rebuild ms/frame, literal shadows, the 1M-triangle target config)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="G-buffer texels fed to the CPU baseline (0 = auto)")
    return ap.parse_args()


_SCENES = {}


def atrium_cached(scenes, tris):
    """The 1 M-triangle stand-in takes seconds to generate on the host: made once per run."""
    if tris not in _SCENES:
        _SCENES[tris] = scenes.atrium(tris)
    return _SCENES[tris]


WORKLOADS = {
    "atrium-262k-1080p": dict(tris=262144, w=1920, h=1080),
    "atrium-1m-1080p": dict(tris=1048576, w=1920, h=1080),
    "atrium-1m-4k": dict(tris=1048576, w=3840, h=2160),
    "cornell-1080p": dict(tris=0, w=1920, h=1080),
    "atrium-8k-small": dict(tris=8192, w=480, h=270),
}


def timed_frames(g, gb_p, gb_n, W, H, out, frames, warm=3):
    """ms per frame of `frames` device-resident whole frames on context g (after `warm` untimed ones)."""
    import torch
    render = type(g).render_tiles_batch_device.__wrapped__
    rays = 0
    for _ in range(warm):
        render(g, [gb_p], [gb_n], W, H, 64, 0, 1, [out])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        render(g, [gb_p], [gb_n], W, H, 64, 0, 1, [out])
        rays += g.last_frame_rays()
    g.sync()
    dt = time.perf_counter() - t0
    return dt / frames * 1e3, rays / dt / 1e6, rays // frames


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
    tuning = {}
    for item in filter(None, args.tuning.split(",")):
        k, v = item.split("=")
        tuning[k] = type(g.tuning()[k])(float(v))
    if tuning:
        g.set_tuning(**tuning)
        g.invalidate()
        g.update()
    build_stats = g.stats()
    photon_info = None
    # BENCH_REHEARSE_CABI=1: the N > 1 route (communicator, rt_hip_frame_multi_enqueue_device, verification) with ONE rank, on a one-GPU box
    rehearse = world == 1 and os.environ.get("BENCH_REHEARSE_CABI") == "1"
    strong_cabi_early = args.scaling == "strong" and (world > 1 or rehearse) and args.route == "cabi" and backend == "nccl"
    if strong_cabi_early:
        # one communicator per context, created by the library (ncclCommInitRank); only the 128-byte id travels through torch
        def bcast(mine):
            box = [mine]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        g.comm_init(rank, world, broadcast=bcast if world > 1 else None)
    if args.photons > 0:
        from raytracer_amd.photons import build_photon_map
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        if strong_cabi_early:
            nph = g.photon_generate_multi(args.photons)     # emit shard -> all-gather of counts -> grouped send / recv -> grid, all in the library
        else:
            nph = build_photon_map(g, args.photons, len(scene.lights), dev)
        g.sync()
        photon_info = {"photons_per_light": args.photons, "photons_stored": nph,
                       "photon_pass_ms": round((time.perf_counter() - t0) * 1e3, 3)}

    strong = args.scaling == "strong"          # at N = 1 a label only: one frame per sub-step either way
    # the library's own RCCL route needs one GPU per rank; the gloo rehearsal (all ranks on device 0) takes the torch route
    cabi = strong and (world > 1 or rehearse) and args.route == "cabi" and backend == "nccl"
    K = args.frames_per_step if args.frames_per_step > 0 else (2 if args.photons > 0 else 300)
    # frames in flight inside ONE context (tuning.frame_streams): the tail of one frame's launches is filled by the next frame's waves
    fif = args.frames_in_flight if args.frames_in_flight > 0 else (3 if cabi and args.photons == 0 else 1)
    fif = max(1, min(4, fif)) if args.photons == 0 else 1
    if fif > 1:
        g.set_tuning(frame_streams=fif)
    # frames per pipeline entry (rt_hip_frames_multi_enqueue_device): with the frame split N ways, N / 2 frames per entry keep
    # every launch as large as half a frame's and every all-gather as large as half a frame
    fpe = args.frames_per_entry if args.frames_per_entry > 0 else (max(1, min(4, world // 2)) if cabi and args.photons == 0 else 1)
    fpe = max(1, min(8, fpe)) if (cabi and args.photons == 0) else 1

    # ---- inputs resident in HBM: one G-buffer per frame of a sub-step -------------------------
    nframes = 1 if (strong or world == 1) else world
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
    frame = torch.zeros(npix * 4, dtype=torch.uint8, device=dev)    # N > 1: the frame assembled from all ranks' tiles
    xchg = None
    if world > 1 and not cabi:
        from raytracer_amd.frames import TileExchange, TileAllGather
        xchg = TileAllGather(g, W, H, 64, rank, world, dev) if strong else TileExchange(g, W, H, 64, rank, world, dev)

    render_batch = rt.RtHip.render_tiles_batch_device.__wrapped__
    render_multi = rt.RtHip.render_frame_multi_device.__wrapped__
    enqueue_multi = rt.RtHip.frame_multi_enqueue_device.__wrapped__
    enqueue_multi_batch = rt.RtHip.frames_multi_enqueue_device.__wrapped__
    outs_fif = [tiles[0]] + [torch.zeros(npix * 4, dtype=torch.uint8, device=dev) for _ in range(fif * fpe - 1)]   # one image per frame in flight
    if cabi and (fif > 1 or fpe > 1):
        frame = outs_fif[0]

    def substep():
        # this rank's tiles of all the sub-step's frames in one wavefront (a batch of one at N = 1)
        if cabi:
            # ONE C call per frame: own tiles shaded into the packed chunk, ncclAllGather, unpack; returns with the whole
            # frame assembled on this rank (the call ends with a stream synchronisation)
            render_multi(g, gb_p[0], gb_n[0], W, H, 64, frame)
            rays = g.last_frame_rays()
        elif strong and xchg:
            xchg.render(gb_p[0], gb_n[0])     # this rank's tiles of THE frame, packed
            rays = g.last_frame_rays()
            xchg.gather(frame)                # RCCL all-gather of the chunks + unpack: every rank holds the whole frame
            g.sync()                          # ... before the next frame starts: a frame's latency, not a pipeline's rate
        elif xchg and os.environ.get("BENCH_XCHG") == "pack":   # debugging aid: full-size tile images, then pack kernels
            g.render_tiles_batch_device(gb_p, gb_n, W, H, 64, rank, world, tiles)
            rays = g.last_frame_rays()
            for f in range(nframes):
                xchg.pack(f, tiles[f])
            xchg.exchange(frame)
        elif xchg and os.environ.get("BENCH_XCHG") == "sync":   # collective between the sub-steps instead of under the next one
            xchg.render(gb_p, gb_n)
            rays = g.last_frame_rays()
            xchg.exchange(frame)
        elif xchg:
            xchg.render(gb_p, gb_n)       # ... written straight into the packed chunks: chunk f goes to rank f
            rays = g.last_frame_rays()
            xchg.finish(frame)            # the PREVIOUS sub-step's all-to-all ran under this one's kernels: unpack -> its frame `rank` is complete
            xchg.begin()                  # this sub-step's all-to-all (RCCL, own stream); finished during the next one, or by drain()
        else:
            # the binding's un-fenced entry: the G-buffers were complete before the loop and the call itself
            # returns only after its last kernel, so the per-call torch-stream drain has nothing to wait for
            render_batch(g, gb_p, gb_n, W, H, 64, rank, world, tiles)
            rays = g.last_frame_rays()
        return rays

    def step():
        return sum(substep() for _ in range(K))

    def drain():
        if xchg and not strong and world > 1:
            xchg.finish(frame)            # the last sub-step's exchange belongs to the timed region

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # --inflight M: M - 1 further contexts (scene and BVH replicated, own stream, own output), one host thread each;
    # sub-step k goes to context k % M.  The frames overlap on the GPU: the tail of one frame's kernels and the host's
    # per-frame work disappear under the other frame's kernels.
    inflight = max(1, args.inflight) if world == 1 and args.photons == 0 else 1
    extra = []
    for _ in range(inflight - 1):
        ge = rt.RtHip(scene, semantics="array", accel=ACCEL[args.accel], device=dev_index,
                      profile=0, bounces=2, do_shadows=1, shadow_t_max=shadow_t_max, seed=42)
        extra.append((ge, [torch.zeros(npix * 4, dtype=torch.uint8, device=dev)]))

    # N = 1: frames go through the library's pipeline (rt_hip_frame_enqueue_device / rt_hip_frame_finish): frame k + 1 is
    # queued before frame k's counters are read back, so the host's per-frame work runs under the GPU's.  Every frame is
    # still rendered whole and in order; BENCH_PIPELINE=0 takes the call-per-frame entry instead.
    pipelined = world == 1 and not cabi and args.photons == 0 and os.environ.get("BENCH_PIPELINE", "1") != "0"
    enqueue = rt.RtHip.frame_enqueue_device.__wrapped__

    def run_steps(count):
        if cabi and (fif > 1 or fpe > 1):
            # the split frame through the library's pipeline: up to `fif` entries of `fpe` frames queued (their shares side by side
            # on this GPU, the all-gather of one under the kernels of the next), finished oldest first -- every rank in the same order
            total, rays, pend, k = count * K, 0, 0, 0
            limit = max(2, fif) if fif > 1 else 1
            while total > 0:
                nf = min(fpe, total)
                if pend == limit:
                    rays += g.frame_finish(); pend -= 1
                o = outs_fif[(k % fif) * fpe:(k % fif) * fpe + nf]
                if nf == 1:
                    enqueue_multi(g, gb_p[0], gb_n[0], W, H, 64, o[0])
                else:
                    enqueue_multi_batch(g, [gb_p[0]] * nf, [gb_n[0]] * nf, W, H, 64, o)
                pend += 1; k += 1; total -= nf
            while pend:
                rays += g.frame_finish(); pend -= 1
            return rays
        if not extra and pipelined:
            total, rays = count * K, 0
            if total == 0:
                return 0
            ahead = max(2, fif) - 1 if fif > 1 else 1    # frames queued ahead of the one being finished
            pend = 0
            for k in range(total):
                if pend > ahead:
                    rays += g.frame_finish(); pend -= 1
                enqueue(g, gb_p[0], gb_n[0], W, H, outs_fif[k % fif]); pend += 1
            while pend:
                rays += g.frame_finish(); pend -= 1
            return rays
        if not extra:
            return sum(step() for _ in range(count))
        import threading
        rays = [0] * inflight

        def worker(t):
            ctx, out = (g, tiles) if t == 0 else extra[t - 1]
            for _ in range(t, count * K, inflight):
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
    substeps = args.steps * K
    frames_done = substeps * nframes

    lone_latency = None
    if cabi and (fif > 1 or fpe > 1):
        # latency of ONE split frame (nothing else in flight), untimed: 10 calls of the blocking entry
        render_multi(g, gb_p[0], gb_n[0], W, H, 64, frame)
        fence()
        t1 = time.perf_counter()
        for _ in range(10):
            render_multi(g, gb_p[0], gb_n[0], W, H, 64, frame)
        lone_latency = (time.perf_counter() - t1) / 10 * 1e3
    # ---- roofline of the dominant kernel (rank 0's launches) --------------------------------
    roofline = build_roofline(args, g, st, gb_p, gb_n, W, H, rank, world, tiles, strong)

    # ---- N > 1 (untimed): the frame assembled from all ranks' tiles == the same frame shaded whole on this rank ----
    tiles_verified = None
    if xchg or cabi:
        whole = torch.zeros(npix * 4, dtype=torch.uint8, device=dev)
        g.render_frame_device(gb_p[0 if strong else rank], gb_n[0 if strong else rank], W, H, whole)
        ok = torch.tensor([1.0 if torch.equal(whole, frame) else 0.0], dtype=torch.float64, device=red_dev)
        if world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        tiles_verified = bool(ok.item() == 1.0)

    # ---- untimed extras on rank 0 at N = 1: the spans SURVEY 8d defines around the device-resident frame ----
    extras, target, more = {}, None, {}
    if rank == 0 and world == 1 and not args.no_extras and args.photons == 0 and inflight == 1 and fif == 1 and not cabi:
        extras, target = run_extras(args, rt, scenes, g, scene, gb_p[0], gb_n[0], W, H, dev, dev_index, ACCEL)
        more = run_more_configs(args, rt, scenes, g, scene, gb_p[0], gb_n[0], W, H, dev, dev_index, ACCEL, elapsed / frames_done * 1e3)

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = run_cpu_baseline(scene, gb_n[0], gb_p[0], W, H, args.cpu_sample, more.get("_ref_equiv_sample"))
        re_cpu = cpu_baseline.pop("reference_equivalent", None)
        if re_cpu and "reference_equivalent" in more:
            more["reference_equivalent"]["cpu"] = re_cpu
            gpu_ms = more["reference_equivalent"].get("gpu_host_pointers_ms")
            if gpu_ms:
                more["reference_equivalent"]["gpu_over_cpu_same_call_same_samples"] = round(re_cpu["ms"] / gpu_ms, 1)
        if "process_rays_full_frame" in more:
            more["process_rays_full_frame"]["cpu_mrays_per_s"] = cpu_baseline["value"]
            more["process_rays_full_frame"]["cpu_note"] = "cpu_baseline.value: the same call (RayTracer::processRays) on a strided sample of the same texels, one thread"
    more.pop("_ref_equiv_sample", None)

    if rank == 0:
        ms_per_frame = elapsed / frames_done * 1e3
        latency = lone_latency if lone_latency is not None else elapsed / substeps * 1e3 * (fif if (fif > 1 and world == 1) else 1)
        out = {
            "metric": "secondary Mrays/s", "value": round(total_rays / elapsed / 1e6, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong" if strong else "weak",   # strong: total work fixed (one 1080p frame per sub-step, split over the ranks)
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload + ("+photons" if args.photons > 0 else ""), "scene": scene.name, "triangles": scene.num_tris, "width": W, "height": H,
                       "frames_per_step": K * nframes, "tile": 64, "bounces": 2, "lights": len(scene.lights),
                       "shadows": args.shadow_mode, "accel": args.accel, "parallelism": "tiles x%d" % world,
                       **({"tuning": tuning} if tuning else {}),
                       **({"frames_in_flight": inflight} if inflight > 1 else {}),
                       **({"frames_in_flight": fif, "frames_in_flight_note": "tuning.frame_streams: that many frames of the library's pipeline run side by side on "
                           "streams of their own; a k_shadow launch then shares the GPU with another frame's, so avg_launch_ms is not the launch's own cost"} if fif > 1 else {}),
                       **({"frames_per_entry": fpe, "frames_per_entry_note": "rt_hip_frames_multi_enqueue_device: this rank's tiles of that many frames are shaded by one pass "
                           "of launches and travel in one all-gather; the frames of an entry are finished together (frame_latency_ms is a lone frame's)"} if fpe > 1 else {}),
                       **({"frame_pipeline": "rt_hip_frame_enqueue_device / rt_hip_frame_finish, one frame queued ahead"} if (pipelined and inflight == 1) else {}),
                       "rays_per_frame": round(total_rays / frames_done, 1),
                       "ms_per_frame": round(ms_per_frame, 4),
                       "frame_latency_ms": round(latency, 4),
                       "ms_per_frame_definition": "ms_per_frame = wall time / frames completed by the whole job (G-buffer and image resident in "
                                                  "HBM, BVH build excluded); frame_latency_ms = first kernel to assembled image of one frame"
                                                  + (" (= one frame split over all ranks + all-gather of its tiles)" if (strong and world > 1) else
                                                     " (at N > 1: of the N frames of one sub-step, rendered together)" if world > 1 else ""),
                       "timed_region_s": round(elapsed, 3),
                       "bvh_build_ms": round(build_stats["build_ms"], 3), "bvh_depth": build_stats["bvh_depth"],
                       **extras,
                       **({"multi_gpu_route": ("C ABI: rt_hip_comm_init (id broadcast over torch.distributed) + " +
                                               (("rt_hip_frames_multi_enqueue_device / rt_hip_frame_finish, %d entries of %d frames in flight: " % (fif, fpe)) if fpe > 1 else
                                                ("rt_hip_frame_multi_enqueue_device / rt_hip_frame_finish, %d frames in flight: " % fif) if fif > 1 else "rt_hip_render_frame_multi_device: ") +
                                               "RCCL all-gather of the packed tiles inside the library, every rank holds the whole frame" if cabi else
                                               "torch.distributed: raytracer_amd/frames.py"),
                           "exchange": ("all-gather of packed tiles after every frame" if strong else
                                        "all-to-all of packed tiles, overlapped with the next sub-step") if os.environ.get("BENCH_XCHG") is None else os.environ["BENCH_XCHG"],
                           "assembled_frames_equal_whole_frames": tiles_verified} if (xchg or cabi) else {}),
                       **({"photon_map": photon_info} if photon_info else {})},
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
        }
        if target:
            out["target_config"] = target
        out.update(more)
        print(json.dumps(out))
    if cabi:
        g.comm_destroy()
    g.close()
    if world > 1:
        dist.destroy_process_group()


def build_roofline(args, g, st, gb_p, gb_n, W, H, rank, world, tiles, strong):
    """One flat object for the kernel with the most GPU time among all timed kernels of the frame."""
    any_ms, any_launches, rays_any = st["any_ms"], st["any_launches"], st["rays_any"]
    clo_ms, clo_launches, rays_clo = st["closest_ms"], st["closest_launches"], st["rays_closest"]
    gat_ms, gat_launches, gat_points = st["gather_ms"], st["gather_launches"], st["gather_points"]
    frame_ms = st["frame_ms"]
    # census pass: same frame(s), counting node visits / triangle tests per ray and photons per gather (untimed)
    # The algorithmic work per ray is that of a ray traced by itself from the root: the census runs with the shadow rays' shared descent
    # (tuning.shadow_prefix) switched off; what the kernels execute with it on is reported beside it.
    def census_pass():
        g.set_config(census=1, profile=0)
        g.reset_stats()
        if strong and world > 1:
            g.render_tiles_device(gb_p[0], gb_n[0], W, H, 64, rank, world, tiles[0])
        else:
            g.render_tiles_batch_device(gb_p, gb_n, W, H, 64, rank, world, tiles)
        c = g.stats()
        g.set_config(census=0, profile=1)
        return c
    tun = g.tuning()
    cs_exec = census_pass()
    g.set_tuning(shadow_prefix=0)
    cs = census_pass()
    g.set_tuning(shadow_prefix=tun["shadow_prefix"])
    nodes_any = cs["census_nodes_any"] / max(cs["census_rays_any"], 1)
    tris_any = cs["census_tris_any"] / max(cs["census_rays_any"], 1)
    nodes_clo = cs["census_nodes_closest"] / max(cs["census_rays_closest"], 1)
    tris_clo = cs["census_tris_closest"] / max(cs["census_rays_closest"], 1)
    counters, counters_note = offline_counters(args.workload, args.shadow_mode, args.accel, args.photons) if world == 1 else (None, "N > 1: no counter profile")

    share = lambda ms: round(ms / frame_ms, 4) if frame_ms > 0 else None
    if gat_ms > max(any_ms, clo_ms):
        # photon-mapped frames: radiance_estimate's gather (k_photon_gather + k_photon_gather_wave, timed as one)
        launches = max(gat_launches, 1)
        avg_ms = gat_ms / launches
        visits = cs["census_gather_visits"] / max(cs["gather_launches"], 1)     # photons in the cubes of one launch's points
        points = gat_points / launches
        flops = visits * GATHER_VISIT_FLOP
        achieved = flops / (avg_ms * 1e-3) / 1e12
        roof = {"bound": "valu", "achieved": round(achieved, 3), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / VALU_PEAK_TFLOPS, 4),
                "traffic": counters.get("traffic_bytes_per_launch") if counters else None,
                "kernel": "k_photon_gather + k_photon_gather_wave<pass 1> + k_photon_select_common + k_photon_gather_wave<2> (radiance_estimate, timed as one launch: classification and one lane per sparse point; one wave per point with more than tuning.gather_heavy photons in its cube)",
                "avg_launch_ms": round(avg_ms, 4), "share_of_frame_gpu_time": share(gat_ms),
                "points_per_launch": round(points, 1), "photon_visits_per_launch": round(visits, 1),
                "photon_visits_per_point": round(visits / max(points, 1), 1),
                "flop_per_visit": GATHER_VISIT_FLOP, "bytes_per_visit": GATHER_VISIT_BYTES,
                "work_definition": "the first pass only, priced as a plain grid walk would do it: one distance test (10 FLOP, 16 B) per photon of the "
                                   "coarse-grid cells within the radius, to count the candidates and find max_dist2 (rt_photon_map.cpp:159-169). "
                                   "The kernels do less than that (cells inside the sphere are counted from the cell table, not read) and more "
                                   "(the selection and the sum, not credited)",
                "algorithmic_cache_GBs": round(visits * GATHER_VISIT_BYTES / (avg_ms * 1e-3) / 1e9, 1),
                "algorithmic_cache_note": "16 B x photon visits / launch time: served by L1/L2 (neighbouring points read the same cells), not an HBM rate",
                "bound_note": "a distance test per photon out of cache-resident cells: the L1 data return (td_busy_frac) and the vector ALU (valu_issue_frac "
                              "of the slots at 2 cycles per wave64 instruction) together; the selection passes and the load imbalance between points are why "
                              "the credited first pass reaches a small fraction of the peak"}
    else:
        fused = clo_launches == 0 and rays_clo > 0   # the frame driver folds the closest-hit queues into the shadow launches
        if fused:
            dominant = "k_shadow (shadow any-hit rays + the level's closest-hit queue in the same launch)"
            launches = max(any_launches, 1)
            rays_per_launch = (rays_any + rays_clo) / launches
            avg_ms, kernel_ms = any_ms / launches, any_ms
            nodes_dom = (rays_any * nodes_any + rays_clo * nodes_clo) / max(rays_any + rays_clo, 1)
            tris_dom = (rays_any * tris_any + rays_clo * tris_clo) / max(rays_any + rays_clo, 1)
        elif any_ms >= clo_ms:
            dominant = "k_shadow (any-hit)"
            rays_per_launch = rays_any / max(any_launches, 1)
            avg_ms, kernel_ms = any_ms / max(any_launches, 1), any_ms
            nodes_dom, tris_dom = nodes_any, tris_any
        else:
            dominant = "k_bvh (closest-hit)"
            rays_per_launch = rays_clo / max(clo_launches, 1)
            avg_ms, kernel_ms = clo_ms / max(clo_launches, 1), clo_ms
            nodes_dom, tris_dom = nodes_clo, tris_clo
        flop_per_ray = NODE_FLOP * nodes_dom + TRI_FLOP * tris_dom
        sec = avg_ms * 1e-3
        achieved = rays_per_launch * flop_per_ray / sec / 1e12 if avg_ms > 0 else 0.0
        bpr32 = NODE_BYTES * nodes_dom + 48.0 * tris_dom + 48.0
        bpr64 = NODE_BYTES_SURVEY * nodes_dom + 48.0 * tris_dom + 48.0
        roof = {"bound": "valu", "achieved": round(achieved, 3), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / VALU_PEAK_TFLOPS, 4),
                "traffic": counters.get("traffic_bytes_per_launch") if counters else None,
                "kernel": dominant, "avg_launch_ms": round(avg_ms, 4), "share_of_frame_gpu_time": share(kernel_ms),
                "rays_per_launch": round(rays_per_launch, 1), "flop_per_ray": round(flop_per_ray, 1),
                "nodes_per_ray": round(nodes_dom, 2), "tris_per_ray": round(tris_dom, 2),
                "kernel_mrays_per_s": round(rays_per_launch / sec / 1e6, 1) if avg_ms > 0 else 0.0,
                "work_definition": "34 FLOP per BVH node visit (both child boxes: 12 FMA + 8 min/max + 2 compares) + 45 per triangle test "
                                   "(SURVEY.md 8d), visits and tests counted on the device by a census pass over the identical frame with every ray "
                                   "traced from the root by itself (tuning.shadow_prefix = 0): the algorithm's work per ray.  The timed kernels share "
                                   "the top of the descent between the 64 shadow rays of an 8x8 tile (k_shadow_prefix, its time included in "
                                   "avg_launch_ms) and execute any_hit_nodes_per_ray_executed visits per ray",
                "any_hit_nodes_per_ray_executed": round(cs_exec["census_nodes_any"] / max(cs_exec["census_rays_any"], 1), 2),
                "any_hit_tris_per_ray_executed": round(cs_exec["census_tris_any"] / max(cs_exec["census_rays_any"], 1), 2),
                "shadow_prefix": tun["shadow_prefix"], "shadow_prefix_steps": tun["shadow_prefix_steps"],
                "algorithmic_cache_GBs_32B_nodes": round(rays_per_launch * bpr32 / sec / 1e9, 1) if avg_ms > 0 else 0.0,
                "algorithmic_cache_GBs_64B_nodes": round(rays_per_launch * bpr64 / sec / 1e9, 1) if avg_ms > 0 else 0.0,
                "algorithmic_cache_note": "rays x (node bytes x nodes/ray + 48 x tris/ray + 48): the 32-byte quantised node the kernel reads, and "
                                          "SURVEY 8d's 64-byte price; served by L1/L2/Infinity Cache (re-reads of a ~21 MB tree), NOT an HBM rate "
                                          "-- the HBM side is hbm_counter_GBs",
                "any_hit_rays_per_frame": round(rays_any / max(st["frames"], 1), 1), "any_hit_nodes_per_ray": round(nodes_any, 2),
                "any_hit_tris_per_ray": round(tris_any, 2),
                "closest_rays_per_frame": round(rays_clo / max(st["frames"], 1), 1), "closest_nodes_per_ray": round(nodes_clo, 2),
                "closest_tris_per_ray": round(tris_clo, 2), "closest_standalone_launches": clo_launches}
    avg_ms = roof["avg_launch_ms"]
    if counters:
        t = counters.get("traffic_bytes_per_launch")
        if t and avg_ms > 0:
            roof["hbm_counter_GBs"] = round(t / (avg_ms * 1e-3) / 1e9, 1)
            roof["hbm_frac"] = round(t / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        # cache-level rates: requests x line size (offline counters) / this run's launch time, against the guide's L2 rate (34.5 TB/s over
        # the 8 XCDs) and the L1s' 64 B/clk/CU x 256 CUs x 2.4 GHz
        if counters.get("l2_request_bytes_per_launch") and avg_ms > 0:
            roof["l2_GBs"] = round(counters["l2_request_bytes_per_launch"] / (avg_ms * 1e-3) / 1e9, 1)
            roof["l2_frac"] = round(roof["l2_GBs"] / L2_PEAK_GBS, 4)
        if counters.get("l1_request_bytes_per_launch") and avg_ms > 0:
            roof["l1_GBs"] = round(counters["l1_request_bytes_per_launch"] / (avg_ms * 1e-3) / 1e9, 1)
            roof["l1_frac"] = round(roof["l1_GBs"] / L1_PEAK_GBS, 4)
            roof["cache_rates_note"] = ("l1 / l2: requests x line size (TCP_TOTAL_CACHE_ACCESSES x 64 B, TCC_REQ x 128 B; upper bounds on the bytes moved) / launch time, "
                                        "peaks %.0f GB/s (64 B/clk/CU x 256 CUs x 2.4 GHz) and %.0f GB/s (MI355X_MICROARCH.md, L2)" % (L1_PEAK_GBS, L2_PEAK_GBS))
        for k in ("valu_issue_frac", "lanes_active_frac", "ta_busy_frac", "td_busy_frac", "l1_hit_frac", "l2_hit_frac", "wait_any_frac", "profiled_avg_launch_ms"):
            if k in counters:
                roof[k] = counters[k]
    roof["counters_source"] = counters_note
    roof.setdefault("bound_note", "a launch is as long as its waves' chains of dependent node fetches (wait_any_frac): the busiest unit is the vector-memory return "
                                  "path (td_busy_frac: each 16 B/lane load occupies the CU's L1 data return for 16 cycles whatever the cache hit rate, a node visit "
                                  "is two of them per wave, shared by the CU's four SIMDs), but halving the bytes per ray (two rays per lane, one fetch for both) "
                                  "lengthened the launch by a quarter -- profiles/r3_variants/k_shadow.md 9; the vector ALU issues valu_issue_frac of the slots it "
                                  "has at 2 cycles per wave64 instruction (rounds 1-2 quoted twice that: 4 cycles, the rate of a wave alone), at two thirds of "
                                  "the lanes active; HBM carries a few percent of its peak")
    return roof


def run_extras(args, rt, scenes, g, scene, gb_p, gb_n, W, H, dev, dev_index, ACCEL):
    """Figures SURVEY 8d asks for beside the device-resident frame, none of them `value`: the host-to-host frame
    (G-buffer uploaded with hipMemcpyAsync from pinned memory -> image in pinned host memory), the frame with the
    acceleration structure brought up to date first (what RayTracer::processRays does every call,
    rt_raytracer.cpp:457-503), the literal-shadow mode, and the north_star target scene (~1M triangles, 1080p)."""
    import torch
    npix = W * H
    out_dev = torch.zeros(npix * 4, dtype=torch.uint8, device=dev)
    cfg = {}
    # host-to-host through rt_hip_render_frame (pinned host buffers)
    h_p = gb_p.cpu().pin_memory()
    h_n = gb_n.cpu().pin_memory()
    h_o = torch.zeros(npix * 4, dtype=torch.uint8).pin_memory()
    fp = ctypes.POINTER(ctypes.c_float)
    call = lambda: g._ck(g.lib.rt_hip_render_frame(g.h, ctypes.cast(h_p.data_ptr(), fp), ctypes.cast(h_n.data_ptr(), fp), W, H,
                                                   ctypes.cast(h_o.data_ptr(), ctypes.POINTER(ctypes.c_uint8))))
    for _ in range(3):
        call()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        call()
        ts.append((time.perf_counter() - t0) * 1e3)
    cfg["ms_per_frame_host_to_host"] = round(float(np.median(ts)), 3)
    cfg["host_to_host_note"] = ("rt_hip_render_frame: %.1f MB G-buffer up with hipMemcpyAsync from pinned host memory, frame, %.1f MB image down; "
                                "median of 20; PCIe-bound, never `value`" % (npix * 28 / 1e6, npix * 4 / 1e6))
    # its legs, as SURVEY 8d lists them (upload, trace + shade, download), each by HIP events around the same copies / frame
    try:
        d_p2, d_n2 = torch.empty_like(gb_p), torch.empty_like(gb_n)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        legs = {"upload": [], "frame": [], "download": []}
        for i in range(13):
            torch.cuda.synchronize()
            ev[0].record()
            d_p2.copy_(h_p, non_blocking=True); d_n2.copy_(h_n, non_blocking=True)
            ev[1].record()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            g.render_frame_device(d_p2, d_n2, W, H, out_dev); g.sync()
            t_frame = (time.perf_counter() - t0) * 1e3
            ev[2].record()
            h_o.copy_(out_dev, non_blocking=True)
            ev[3].record()
            torch.cuda.synchronize()
            if i >= 3:
                legs["upload"].append(ev[0].elapsed_time(ev[1])); legs["frame"].append(t_frame); legs["download"].append(ev[2].elapsed_time(ev[3]))
        cfg["host_to_host_legs_ms"] = {k: round(float(np.median(v)), 3) for k, v in legs.items()}
        del d_p2, d_n2
    except Exception as e:
        cfg["host_to_host_legs_ms"] = "not measured: %r" % (e,)
    # the same host-to-host frames through the pipeline entry (rt_hip_frame_enqueue: uploads and downloads on streams of
    # their own, one frame enqueued ahead): what a host-buffer caller that draws frame after frame gets per frame
    try:
        h_o2 = torch.zeros(npix * 4, dtype=torch.uint8).pin_memory()
        outs = (h_o, h_o2)
        n = 40
        g.frame_enqueue_host(h_p, h_n, W, H, outs[0]); g.frame_finish()          # warm: staging buffers, streams
        t0 = time.perf_counter()
        g.frame_enqueue_host(h_p, h_n, W, H, outs[0])
        for i in range(1, n):
            g.frame_enqueue_host(h_p, h_n, W, H, outs[i & 1])
            g.frame_finish()
        g.frame_finish()
        cfg["ms_per_frame_host_to_host_pipelined"] = round((time.perf_counter() - t0) * 1e3 / n, 3)
        cfg["host_to_host_pipelined_note"] = ("rt_hip_frame_enqueue / rt_hip_frame_finish, %d frames, one enqueued ahead: frame k + 1's G-buffer crosses "
                                              "PCIe under frame k's kernels; wall time / frames" % n)
    except Exception as e:
        cfg["host_to_host_pipelined_note"] = "not measured: %r" % (e,)
    # ... and with the G-buffer in the FBO's own 16-bit formats (rt_hip_frame_enqueue_g16: 14 instead of 28 B per texel over PCIe)
    try:
        nr = h_n.numpy().reshape(-1, 4); po = h_p.numpy().reshape(-1, 3)
        n16 = np.empty((npix, 4), np.uint16)
        n16[:, :3] = np.rint(nr[:, :3].astype(np.float64) * 65535.0).astype(np.uint16)
        n16[:, 3] = nr[:, 3].astype(np.uint16)
        h_n16 = torch.from_numpy(n16).pin_memory()
        h_p16 = torch.from_numpy(po.astype(np.float16).view(np.uint16)).pin_memory()
        n = 40
        g.frame_enqueue_host_g16(h_n16, h_p16, W, H, outs[0]); g.frame_finish()
        ref_img = h_o.clone()
        g.frame_enqueue_host(h_p, h_n, W, H, h_o); g.frame_finish(); ref_img.copy_(h_o)
        g.frame_enqueue_host_g16(h_n16, h_p16, W, H, h_o2); g.frame_finish()
        same = bool(torch.equal(ref_img, h_o2))
        t0 = time.perf_counter()
        g.frame_enqueue_host_g16(h_n16, h_p16, W, H, outs[0])
        for i in range(1, n):
            g.frame_enqueue_host_g16(h_n16, h_p16, W, H, outs[i & 1])
            g.frame_finish()
        g.frame_finish()
        cfg["ms_per_frame_host_to_host_g16_pipelined"] = round((time.perf_counter() - t0) * 1e3 / n, 3)
        cfg["host_to_host_g16_note"] = ("rt_hip_frame_enqueue_g16: RGBA16 UNORM normals + RGB16F positions (the rasterizer's FBO formats, ras_rasterizer.cpp:293,301), "
                                        "%.1f MB up per frame, widened on the device; image identical to the float entry's: %s" % (npix * 14 / 1e6, same))
        # ... and with three frames in flight (tuning.frame_streams = 3): uploads, frames on three streams, downloads
        g.set_tuning(frame_streams=3)
        outs3 = (h_o, h_o2, torch.zeros(npix * 4, dtype=torch.uint8).pin_memory())
        for k in range(3):
            g.frame_enqueue_host_g16(h_n16, h_p16, W, H, outs3[k])
        for k in range(3):
            g.frame_finish()
        n, pend = 60, 0
        t0 = time.perf_counter()
        for i in range(n):
            if pend == 3:
                g.frame_finish(); pend -= 1
            g.frame_enqueue_host_g16(h_n16, h_p16, W, H, outs3[i % 3]); pend += 1
        while pend:
            g.frame_finish(); pend -= 1
        cfg["ms_per_frame_host_to_host_g16_3_in_flight"] = round((time.perf_counter() - t0) * 1e3 / n, 3)
        same3 = bool(torch.equal(ref_img, outs3[2]))
        cfg["host_to_host_g16_3_in_flight_note"] = ("the same entry with tuning.frame_streams = 3, %d frames; image identical: %s (no gain expected here: two copy streams "
                                                    "and three lanes share the process's four hardware queues, and the upload is the longer leg)" % (n, same3))
        g.set_tuning(frame_streams=1)
        del h_n16, h_p16
    except Exception as e:
        g.set_tuning(frame_streams=1)
        cfg["host_to_host_g16_note"] = "not measured: %r" % (e,)
    # one frame at a time, as SURVEY 8d words the GPU timing (3 warm-up + 20 timed, median and p95): call -> image complete,
    # host clock around rt_hip_render_frame_device + stream sync, G-buffer and image resident in HBM
    ts = []
    for i in range(23):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.render_frame_device(gb_p, gb_n, W, H, out_dev)
        g.sync()
        if i >= 3:
            ts.append((time.perf_counter() - t0) * 1e3)
    cfg["ms_per_frame_single_median"] = round(float(np.median(ts)), 4)
    cfg["ms_per_frame_single_p95"] = round(float(np.percentile(ts, 95)), 4)
    cfg["single_frame_note"] = "one frame per call with a stream sync after each (latency of a lone frame), median and 95th percentile of 20"
    # The timed region runs with config.profile = 1 (every k_shadow launch is timed through its dispatch packet, which is what the
    # roofline object needs); the same pipelined loop without it, for what a caller that does not time kernels gets
    try:
        enq = type(g).frame_enqueue_device.__wrapped__ if hasattr(type(g).frame_enqueue_device, "__wrapped__") else type(g).frame_enqueue_device
        def piped(n):
            enq(g, gb_p, gb_n, W, H, out_dev)
            for _ in range(n - 1):
                enq(g, gb_p, gb_n, W, H, out_dev)
                g.frame_finish()
            g.frame_finish()
            g.sync()
        g.set_config(profile=0)
        piped(30)
        t0 = time.perf_counter(); piped(300); dt = time.perf_counter() - t0
        g.set_config(profile=1)
        cfg["ms_per_frame_without_kernel_timing"] = round(dt / 300 * 1e3, 4)
        cfg["without_kernel_timing_note"] = "the timed region's loop (rt_hip_frame_enqueue_device / rt_hip_frame_finish, one frame queued ahead), 300 frames, config.profile = 0; never `value`"
    except Exception as e:
        g.set_config(profile=1)
        cfg["without_kernel_timing_note"] = "not measured: %r" % (e,)
    # tuning.frame_streams = 2, 3: frames of the pipeline on streams of their own, side by side (throughput, not `value`: the headline
    # line keeps launches that do not overlap, which is what its roofline object prices)
    try:
        enq = type(g).frame_enqueue_device.__wrapped__ if hasattr(type(g).frame_enqueue_device, "__wrapped__") else type(g).frame_enqueue_device
        outs = [out_dev] + [torch.zeros(npix * 4, dtype=torch.uint8, device=dev) for _ in range(2)]
        for lanes in (2, 3):
            g.set_tuning(frame_streams=lanes)
            def piped(n):
                pend = 0
                for i in range(n):
                    if pend == lanes:
                        g.frame_finish(); pend -= 1
                    enq(g, gb_p, gb_n, W, H, outs[i % lanes]); pend += 1
                while pend:
                    g.frame_finish(); pend -= 1
                g.sync()
            piped(30)
            t0 = time.perf_counter(); piped(300); dt = time.perf_counter() - t0
            cfg["ms_per_frame_%d_in_flight" % lanes] = round(dt / 300 * 1e3, 4)
        g.set_tuning(frame_streams=1)
        cfg["in_flight_note"] = ("tuning.frame_streams = 2 / 3: enqueued frames are dealt to that many streams (level arrays of their own) and run side by side, "
                                 "300 frames, with kernel timing on; throughput of a frame loop, never `value`")
    except Exception as e:
        g.set_tuning(frame_streams=1)
        cfg["in_flight_note"] = "not measured: %r" % (e,)
    # The frame with the scene brought up to date first, on a context that applies model matrices as processRays does
    # (use_model_mat = 1; the scene's matrices are identities, so the geometry and the image are the same).
    render = type(g).render_tiles_batch_device.__wrapped__
    g2 = rt.RtHip(scene, semantics="array", accel=ACCEL[args.accel], device=dev_index, profile=0, bounces=2, do_shadows=1,
                  use_model_mat=1, shadow_t_max=1.0 if args.shadow_mode == "corrected" else 0.0, seed=42)
    mats = np.stack([np.ascontiguousarray(me.model_colmajor(), dtype=np.float32).reshape(16) for me in scene.meshes])
    kinds = {}

    def frame_after(prepare, label, n=20):
        ts = []
        for i in range(n + 3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            prepare(i)
            g2.update()
            render(g2, [gb_p], [gb_n], W, H, 64, 0, 1, [out_dev])
            g2.sync()
            if i >= 3:
                ts.append((time.perf_counter() - t0) * 1e3)
        kinds[label] = ["none", "tables", "refit", "rebuild"][g2.stats()["last_update"]]
        return round(float(np.median(ts)), 3)

    def same(i):
        g2.set_model_mats(0, mats)

    def jitter(i):   # every mesh moved a little, differently each frame: new boxes, same topology
        mm = mats.copy()
        mm[:, 12] += 1e-3 * (1 + (i & 1))
        g2.set_model_mats(0, mm)

    def rebuild(i):
        jitter(i)
        g2.invalidate()

    cfg["ms_per_frame_with_update_unchanged"] = frame_after(same, "unchanged")
    cfg["ms_per_frame_with_update_moved"] = frame_after(jitter, "moved")
    cfg["ms_per_frame_with_rebuild"] = frame_after(rebuild, "rebuild")
    cfg["with_update_kinds"] = kinds
    g2.close()
    cfg["with_update_note"] = ("one rt_hip_set_model_mats call for all %d meshes + rt_hip_update + the device-resident frame, host clock, median of 20 "
                               "(the slot of processRays' per-call soup rebuild, rt_raytracer.cpp:457-503): matrices unchanged (update finds "
                               "nothing to do), every matrix moved (soup + refit on the kept topology), build from scratch forced" % len(scene.meshes))
    # literal reference shadows on the same frame
    if args.shadow_mode == "corrected":
        g.set_config(shadow_t_max=0.0)
        ms, mr, rpf = timed_frames(g, gb_p, gb_n, W, H, out_dev, 100)
        g.set_config(shadow_t_max=1.0)
        cfg["literal_shadows_ms_per_frame"] = round(ms, 4)
        cfg["literal_shadows_mrays_per_s"] = round(mr, 1)
    # SURVEY 8d's incoherent stress set: 2^22 rays, uniform origins in the scene's box, uniform directions, seed 7,
    # through the batch entries (rt_hip_trace_closest_device / rt_hip_trace_any_device), HIP-event time of the kernels
    try:
        n = 1 << 22
        V = np.concatenate([np.asarray(me.verts, np.float32).reshape(-1, 3) for me in scene.meshes], 0)
        lo = torch.from_numpy(V.min(0)).to(dev); hi = torch.from_numpy(V.max(0)).to(dev)
        gen = torch.Generator(device="cpu").manual_seed(7)
        org = (lo + (hi - lo) * torch.rand((n, 3), generator=gen).to(dev)).contiguous()
        d = torch.randn((n, 3), generator=gen).to(dev)
        d = (d / d.norm(dim=1, keepdim=True)).contiguous()
        hits = torch.zeros(n * 4, dtype=torch.float32, device=dev)
        occ = torch.zeros(n, dtype=torch.uint8, device=dev)
        g.set_config(profile=1)
        for name, call in (("closest", lambda: g.trace_closest_device(org, d, n, hits)), ("any", lambda: g.trace_any_device(org, d, n, occ))):
            call(); g.sync(); g.reset_stats()
            for _ in range(5):
                call()
            g.sync()
            st = g.stats()
            ms = (st["closest_ms"] if name == "closest" else st["any_ms"]) / 5
            cfg["incoherent_%s_mrays_per_s" % name] = round(n / (ms * 1e-3) / 1e6, 1)
        prim = hits.view(torch.int32).view(-1, 4)[:, 0]
        cfg["incoherent_note"] = ("SURVEY 8d stress set: 2^22 rays, origins uniform in the scene's box, directions uniform on the sphere, seed 7; "
                                  "kernel time of rt_hip_trace_closest_device / rt_hip_trace_any_device (t_max 1e6), %.0f %% of the rays hit" %
                                  (100.0 * float((prim != -1).float().mean())))
        del org, d, hits, occ
    except Exception as e:   # an extra must not take the line with it
        cfg["incoherent_note"] = "not measured: %r" % (e,)
    # north_star's target: ~1M triangles at 1080p, same shading
    target = None
    if args.workload == "atrium-262k-1080p":
        sc1 = atrium_cached(scenes, 1048576)
        g1 = rt.RtHip(sc1, semantics="array", accel=ACCEL[args.accel], device=dev_index, profile=0, bounces=2, do_shadows=1,
                      shadow_t_max=1.0 if args.shadow_mode == "corrected" else 0.0, seed=42)
        b1 = g1.stats()
        d_n = torch.zeros(npix * 4, dtype=torch.float32, device=dev)
        d_p = torch.zeros(npix * 3, dtype=torch.float32, device=dev)
        g1.render_gbuffer_device(sc1.fov, W, H, d_n, d_p)
        g1.sync()
        ms, mr, rpf = timed_frames(g1, d_p, d_n, W, H, out_dev, 200)
        target = {"workload": "atrium-1m-1080p", "scene": sc1.name, "triangles": sc1.num_tris, "width": W, "height": H,
                  "value": round(mr, 2), "unit": "Mrays/s", "ms_per_frame": round(ms, 4), "frames": 200, "rays_per_frame": int(rpf),
                  "shadows": args.shadow_mode, "accel": args.accel, "bvh_build_ms": round(b1["build_ms"], 3), "bvh_depth": b1["bvh_depth"],
                  "note": "north_star's target scene size (~1M triangles, 1080p), same run, G-buffer and image resident in HBM, 1 GPU"}
        g1.close()
    return cfg, target


def _strong_share(rt, torch, g, d_p, d_n, w, h, whole_ms, reps=6):
    """What a strong-scaling split costs, predicted on ONE GPU: for N = 2, 4, 8 every virtual rank shades its interleaved 64x64
    tiles of the frame into its packed chunk (rt_hip_render_tiles_packed_device, the call rank r of an N-GPU job makes) and the
    N chunks are unpacked (rt_hip_unpack_chunks_device); the frame time of the job is that of its slowest rank + the unpack
    (+ the all-gather, which one GPU cannot measure).  Implied efficiency = whole-frame time / (N x that)."""
    from raytracer_amd.tiles import max_owned_tiles
    packed = type(g).render_tiles_packed_device.__wrapped__
    out = {}
    img = torch.zeros(w * h * 4, dtype=torch.uint8, device=d_p.device)
    for n in (2, 4, 8):
        chunk = max_owned_tiles(w, h, 64, n) * 64 * 64 * 4
        recv = torch.zeros(n * chunk, dtype=torch.uint8, device=d_p.device)
        per_rank = []
        for r in range(n):
            dst = [recv[r * chunk:(r + 1) * chunk]]
            for _ in range(2):
                packed(g, [d_p], [d_n], w, h, 64, r, n, dst)
            g.sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                packed(g, [d_p], [d_n], w, h, 64, r, n, dst)
            g.sync()
            per_rank.append((time.perf_counter() - t0) / reps * 1e3)
        unpack = type(g).unpack_chunks_device.__wrapped__
        unpack(g, recv, chunk, w, h, 64, n, img); g.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            unpack(g, recv, chunk, w, h, 64, n, img)
        g.sync()
        un = (time.perf_counter() - t0) / reps * 1e3
        slow = max(per_rank)
        out[str(n)] = {"slowest_rank_ms": round(slow, 4), "fastest_rank_ms": round(min(per_rank), 4), "unpack_ms": round(un, 4),
                       "implied_efficiency": round(whole_ms / (n * (slow + un)), 3), "all_gather_bytes_per_rank": int(chunk)}
        # the same share with three frames in flight (tuning.frame_streams = 3, rt_hip_tiles_packed_enqueue_device: what
        # rt_hip_frame_multi_enqueue_device queues per frame before its all-gather): wall time per frame of the slowest rank
        try:
            r = per_rank.index(slow)
            enq = type(g).tiles_packed_enqueue_device.__wrapped__
            dsts = [torch.zeros(chunk, dtype=torch.uint8, device=d_p.device) for _ in range(3)]
            g.set_tuning(frame_streams=3)
            def piped(k):
                pend = 0
                for i in range(k):
                    if pend == 3:
                        g.frame_finish(); pend -= 1
                    enq(g, d_p, d_n, w, h, 64, r, n, dsts[i % 3]); pend += 1
                while pend:
                    g.frame_finish(); pend -= 1
                g.sync()
            piped(6)
            t0 = time.perf_counter(); piped(10 * reps); dt = (time.perf_counter() - t0) / (10 * reps) * 1e3
            g.set_tuning(frame_streams=1)
            out[str(n)]["slowest_rank_ms_3_in_flight"] = round(dt, 4)
            out[str(n)]["implied_efficiency_3_in_flight"] = round(whole_ms / (n * (dt + un)), 3)
            # ... and in entries of N / 2 frames (at most 4), three entries in flight: what bench.py --gpus N queues per rank by default
            # (rt_hip_frames_multi_enqueue_device minus its all-gather)
            fpe = max(1, min(4, n // 2))
            if fpe > 1:
                enqb = type(g).tiles_packed_batch_enqueue_device.__wrapped__
                dstb = [[torch.zeros(chunk, dtype=torch.uint8, device=d_p.device) for _ in range(fpe)] for _ in range(3)]
                g.set_tuning(frame_streams=3)
                def pipedb(k):
                    pend = 0
                    for i in range(k):
                        if pend == 3:
                            g.frame_finish(); pend -= 1
                        enqb(g, [d_p] * fpe, [d_n] * fpe, w, h, 64, r, n, dstb[i % 3]); pend += 1
                    while pend:
                        g.frame_finish(); pend -= 1
                    g.sync()
                pipedb(6)
                t0 = time.perf_counter(); pipedb(5 * reps); dtb = (time.perf_counter() - t0) / (5 * reps * fpe) * 1e3
                g.set_tuning(frame_streams=1)
                out[str(n)]["frames_per_entry"] = fpe
                out[str(n)]["slowest_rank_ms_entries_3_in_flight"] = round(dtb, 4)
                out[str(n)]["implied_efficiency_entries_3_in_flight"] = round(whole_ms / (n * (dtb + un)), 3)
        except Exception as e:
            g.set_tuning(frame_streams=1)
            out[str(n)]["in_flight_error"] = repr(e)
    return out


def run_more_configs(args, rt, scenes, g, scene, gb_p, gb_n, W, H, dev, dev_index, ACCEL, whole_ms):
    """Objects beside the headline line, none of them `value` (all on rank 0 at N = 1, untimed by the driver's metric):
    reference_equivalent   SURVEY 8d's run: 0.75x prepass (1440x810), interlace 16 -> what ONE reference frame traces, through
                           RayTracer::processRays' entry with host pointers (rt_hip_process_gbuffer) and device-resident
    process_rays_full_frame  the same call on every covered texel of the 1080p G-buffer (like-for-like with cpu_baseline)
    strong_share_ms        the strong-scaling prediction of _strong_share at 1080p (and at 4K under c4_single_gpu_share)
    c4_single_gpu_share    atrium-1M at 4K: one GPU doing C4's whole frame
    c5                     atrium-262k + 10 M photons emitted: photon pass, photon-mapped frame, gather figures"""
    import torch
    out = {}
    npix = W * H
    shadow_t_max = 1.0 if args.shadow_mode == "corrected" else 0.0
    fp = ctypes.POINTER(ctypes.c_float)
    # ---- processRays: reference-equivalent sample and the full frame, on a context with RayTracer semantics ----
    try:
        g2 = rt.RtHip(scene, accel=ACCEL[args.accel], device=dev_index)      # semantics="raytracer": world space, FLT_EPSILON, t_max = inf
        pw, ph = int(W * 0.75), int(H * 0.75)                                # prepass_resolution 0.75 (ras_rasterizer.cpp:39,167-168)
        d_n = torch.zeros(pw * ph * 4, dtype=torch.float32, device=dev)
        d_p = torch.zeros(pw * ph * 3, dtype=torch.float32, device=dev)
        g.render_gbuffer_device(scene.fov, pw, ph, d_n, d_p)
        g.sync()
        nr = d_n.cpu().numpy().reshape(-1, 4); po = d_p.cpu().numpy().reshape(-1, 3)
        idx = np.arange(pw * ph)
        sel = (idx % 16 == 0) & (nr[:, 3] > 0)                               # INTERLACE 16, phase 0, alpha > 0 (ras_rasterizer.cpp:16,509-519)
        pos = np.ascontiguousarray(po[sel]); nrm = np.ascontiguousarray((nr[sel, :3].astype(np.float64) * 2.0 - 1.0).astype(np.float32))

        def c_call(p_, n_):
            res_p, off_p, nres, nmat = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int(0), ctypes.c_int(0)
            g2._ck(g2.lib.rt_hip_process_gbuffer(g2.h, p_.ctypes.data_as(fp), n_.ctypes.data_as(fp), p_.shape[0], ctypes.byref(res_p),
                                                 ctypes.byref(off_p), ctypes.byref(nres), ctypes.byref(nmat)))
            return nres.value, nmat.value

        def timed(fn, reps):
            for _ in range(3):
                r = fn()
            g2.sync()
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter(); r = fn(); g2.sync(); ts.append((time.perf_counter() - t0) * 1e3)
            return float(np.median(ts)), r

        ms_host, (hits, groups) = timed(lambda: c_call(pos, nrm), 30)
        d_pos = torch.from_numpy(pos).to(dev); d_nrm = torch.from_numpy(nrm).to(dev)
        d_res = torch.zeros(max(pos.shape[0], 1) * 16, dtype=torch.float32, device=dev)
        d_off = torch.zeros(2 * 4096, dtype=torch.int32, device=dev)
        dev_call = type(g2).process_gbuffer_device.__wrapped__
        ms_dev, _ = timed(lambda: dev_call(g2, d_pos, d_nrm, pos.shape[0], d_res, d_off), 30)
        out["reference_equivalent"] = {
            "what": "what ONE frame of the reference traces at 1080p: 0.75x prepass (%dx%d), interlace 16 -> %d reflection rays through "
                    "RayTracer::processRays' entry (ray-gen, closest hit, hit records grouped by material); scene up to date "
                    "(rt_hip_update finds nothing to do; the reference rebuilds its soup every call)" % (pw, ph, pos.shape[0]),
            "rays": int(pos.shape[0]), "hit_records": int(hits), "material_groups": int(groups),
            "gpu_host_pointers_ms": round(ms_host, 4), "gpu_host_pointers_mrays_per_s": round(pos.shape[0] / ms_host / 1e3, 2),
            "gpu_device_resident_ms": round(ms_dev, 4), "gpu_device_resident_mrays_per_s": round(pos.shape[0] / ms_dev / 1e3, 2),
            "entry": "rt_hip_process_gbuffer (host pointers in, pinned host records out: PCIe-inclusive) / rt_hip_process_gbuffer_device; median of 30"}
        out["_ref_equiv_sample"] = (pos, nrm)
        # the whole 1080p G-buffer through the same call
        nrf = gb_n.cpu().numpy().reshape(-1, 4); pof = gb_p.cpu().numpy().reshape(-1, 3)
        self_ = nrf[:, 3] > 0
        posf = np.ascontiguousarray(pof[self_]); nrmf = np.ascontiguousarray((nrf[self_, :3].astype(np.float64) * 2.0 - 1.0).astype(np.float32))
        ms_hostf, (hitsf, groupsf) = timed(lambda: c_call(posf, nrmf), 8)
        d_posf = torch.from_numpy(posf).to(dev); d_nrmf = torch.from_numpy(nrmf).to(dev)
        d_resf = torch.zeros(posf.shape[0] * 16, dtype=torch.float32, device=dev)
        ms_devf, _ = timed(lambda: dev_call(g2, d_posf, d_nrmf, posf.shape[0], d_resf, d_off), 15)
        out["process_rays_full_frame"] = {
            "what": "RayTracer::processRays' entry on all %d covered texels of the 1080p G-buffer (reflection rays, closest hit, 64-byte hit records "
                    "grouped by material): the call cpu_baseline times" % posf.shape[0],
            "rays": int(posf.shape[0]), "hit_records": int(hitsf), "material_groups": int(groupsf),
            "gpu_device_resident_ms": round(ms_devf, 4), "gpu_device_resident_mrays_per_s": round(posf.shape[0] / ms_devf / 1e3, 1),
            "gpu_host_pointers_ms": round(ms_hostf, 4), "gpu_host_pointers_mrays_per_s": round(posf.shape[0] / ms_hostf / 1e3, 1),
            "host_pointers_note": "%.0f MB of positions / normals up, %.0f MB of hit records down per call: PCIe-bound" % (posf.shape[0] * 24 / 1e6, hitsf * 64 / 1e6)}
        del d_posf, d_nrmf, d_resf, d_pos, d_nrm, d_res
        g2.close()
    except Exception as e:   # an extra must not take the line with it
        out["reference_equivalent_error"] = repr(e)
    # ---- strong-scaling share at 1080p on the bench context ----
    try:
        out["strong_share_ms"] = {"1080p": _strong_share(rt, torch, g, gb_p, gb_n, W, H, whole_ms),
                                  "whole_frame_ms": round(whole_ms, 4),
                                  "note": "one GPU playing every rank of an N-GPU strong-scaling job in turn: slowest rank's rt_hip_render_tiles_packed_device "
                                          "(call -> done, host clock) + rt_hip_unpack_chunks_device; implied_efficiency = whole_frame_ms / (N x (slowest + unpack)); "
                                          "the RCCL all-gather between the two is not included (all_gather_bytes_per_rank says what it moves)"}
    except Exception as e:
        out["strong_share_error"] = repr(e)
    # ---- C4's single-GPU share: atrium-1M at 4K ----
    try:
        if args.workload == "atrium-262k-1080p":
            sc1 = atrium_cached(scenes, 1048576)
            g1 = rt.RtHip(sc1, semantics="array", accel=ACCEL[args.accel], device=dev_index, profile=1, bounces=2, do_shadows=1,
                          shadow_t_max=shadow_t_max, seed=42)
            w4, h4 = 3840, 2160
            d_n = torch.zeros(w4 * h4 * 4, dtype=torch.float32, device=dev)
            d_p = torch.zeros(w4 * h4 * 3, dtype=torch.float32, device=dev)
            o4 = torch.zeros(w4 * h4 * 4, dtype=torch.uint8, device=dev)
            g1.render_gbuffer_device(sc1.fov, w4, h4, d_n, d_p)
            g1.sync(); g1.reset_stats()
            ms, mr, rpf = timed_frames(g1, d_p, d_n, w4, h4, o4, 40)
            st1 = g1.stats()
            out["c4_single_gpu_share"] = {"workload": "atrium-1m-4k", "triangles": sc1.num_tris, "width": w4, "height": h4, "ms_per_frame": round(ms, 4),
                                          "value": round(mr, 1), "unit": "Mrays/s", "rays_per_frame": int(rpf), "frames": 40,
                                          "k_shadow_ms_per_launch": round(st1["any_ms"] / max(st1["any_launches"], 1), 4),
                                          "strong_share_ms": _strong_share(rt, torch, g1, d_p, d_n, w4, h4, ms, reps=3),
                                          "note": "BASELINE.json configs[3] (C4: ~1M triangles, 4K, reflections + refractions, 8 GPUs) on ONE GPU: the whole frame, and "
                                                  "under strong_share_ms each of 8 (4, 2) ranks' share of it"}
            g1.close()
            del d_n, d_p, o4
    except Exception as e:
        out["c4_error"] = repr(e)
    # ---- C5: the bench scene + 10 M photons ----
    try:
        if args.workload == "atrium-262k-1080p":
            from raytracer_amd.photons import build_photon_map
            per_light = 2_500_000
            g5 = rt.RtHip(scene, semantics="array", accel=ACCEL[args.accel], device=dev_index, profile=1, bounces=2, do_shadows=1,
                          shadow_t_max=shadow_t_max, use_photon_map=1, photon_estimate=100, photon_radius=1.0, seed=42)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            nph = build_photon_map(g5, per_light, len(scene.lights), dev)
            g5.sync()
            pass_ms = (time.perf_counter() - t0) * 1e3
            o5 = torch.zeros(npix * 4, dtype=torch.uint8, device=dev)
            g5.reset_stats()
            ms, mr, rpf = timed_frames(g5, gb_p, gb_n, W, H, o5, 12, warm=2)
            st5 = g5.stats()
            g5.set_config(census=1, profile=0); g5.reset_stats()
            g5.render_frame_device(gb_p, gb_n, W, H, o5)
            cs = g5.stats()
            launches = max(st5["gather_launches"], 1)
            visits = cs["census_gather_visits"] / max(cs["gather_launches"], 1)
            gms = st5["gather_ms"] / launches
            out["c5"] = {"workload": "atrium-262k-1080p + photons", "photons_emitted": per_light * len(scene.lights), "photons_stored": int(nph),
                         "photon_pass_ms": round(pass_ms, 3), "ms_per_frame": round(ms, 4), "frames": 12, "photon_estimate": 100, "photon_radius": 1.0,
                         "gather_ms_per_launch": round(gms, 4), "gather_launches_per_frame": round(launches / 14.0, 2),
                         "gather_points_per_launch": round(st5["gather_points"] / launches, 1),
                         "gather_roofline": {"bound": "valu", "achieved": round(visits * GATHER_VISIT_FLOP / (gms * 1e-3) / 1e12, 3), "peak": VALU_PEAK_TFLOPS,
                                             "unit": "TFLOP/s", "frac": round(visits * GATHER_VISIT_FLOP / (gms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS, 4),
                                             "photon_visits_per_launch": round(visits, 1),
                                             "work_definition": "first pass priced as a plain grid walk (10 FLOP per photon of the coarse cells within the radius); the kernels "
                                                                "read less (interiors counted from the cell table) and do more (selection, sum)"},
                         "note": "BASELINE.json configs[4] (C5): photon pass = emit (sharded by index) + grids, once; frames shade with radiance_estimate "
                                 "(k = 100, r = 1) instead of shadow rays, through the synchronous frame driver"}
            g5.close()
            del o5
    except Exception as e:
        out["c5_error"] = repr(e)
    return out


def run_cpu_baseline(scene, d_n, d_p, W, H, sample, ref_equiv_sample=None):
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
    ref_equiv = None
    if ref_equiv_sample is not None:
        rp, rn = ref_equiv_sample
        cpu.process_rays(rp, rn)
        ts = []
        for _ in range(3):
            t1 = time.perf_counter(); rr = cpu.process_rays(rp, rn); ts.append((time.perf_counter() - t1) * 1e3)
        ref_equiv = {"ms": round(float(np.median(ts)), 3), "mrays_per_s": round(len(rp) / float(np.median(ts)) / 1e3, 4), "cores": 1, "kind": kind,
                     "hits": int(len(rr["records"])), "sample": "the same %d samples, the same call (soup rebuild + DACRT + records inside it, as every reference frame pays), median of 3" % len(rp)}
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
    if ref_equiv:
        out["reference_equivalent"] = ref_equiv
    return out


if __name__ == "__main__":
    main()
