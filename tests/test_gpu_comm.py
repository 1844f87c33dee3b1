"""The multi-GPU entries of the C ABI (rt_hip_comm_*, rt_hip_render_frame_multi_device,
rt_hip_photon_generate_multi: RCCL called by the library) on the one GPU a test box has: a ONE-rank communicator
runs the same steps as N ranks -- librccl bound at run time, ncclCommInitRank, the tile all-gather and its unpack on
the context's stream, the count all-gather and the grouped send/recv of the photon shards -- and must reproduce the
single-GPU results bit for bit.  N > 1 is covered by index arithmetic on virtual ranks (test_gpu_parity.py,
test_gpu_fullsize.py) and by the gloo tests; RCCL between real peers needs the 8-GPU node."""
import numpy as np
import pytest

from raytracer_amd import scenes

pytestmark = pytest.mark.gpu


def test_one_rank_communicator_reproduces_single_gpu_results():
    import torch
    import raytracer_amd as rt
    sc = scenes.atrium(2048)
    g = rt.RtHip(sc, semantics="array", accel=rt.ACCEL_PLOC, bounces=2, do_shadows=1, shadow_t_max=1.0, seed=42)
    w, h = 200, 150                   # not multiples of the tile: edge tiles carry padding
    d_n = torch.zeros(w * h * 4, dtype=torch.float32, device="cuda")
    d_p = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda")
    g.render_gbuffer_device(sc.fov, w, h, d_n, d_p)
    whole = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
    g.render_frame_device(d_p, d_n, w, h, whole)
    rays = g.last_frame_rays()
    with pytest.raises(rt.RtHipError, match="rt_hip_comm_init"):
        g.render_frame_multi_device(d_p, d_n, w, h, 64, whole)
    g.comm_init(0, 1)
    with pytest.raises(rt.RtHipError, match="already"):
        g.comm_init(0, 1)
    for tile in (64, 16):
        out = torch.full((w * h * 4,), 7, dtype=torch.uint8, device="cuda")
        g.render_frame_multi_device(d_p, d_n, w, h, tile, out)
        assert torch.equal(out, whole), "tile %d" % tile
        assert g.last_frame_rays() == rays
    # the pipelined form: frames queued without waiting, on one and on three lanes, finished in order
    gbs, wholes = [(d_p, d_n)], [(whole, rays)]
    for f in (1, 2, 3):
        n2 = torch.zeros_like(d_n); p2 = torch.zeros_like(d_p)
        g.render_gbuffer_device(sc.fov - 0.05 * f, w, h, n2, p2)
        o2 = torch.zeros_like(whole)
        g.render_frame_device(p2, n2, w, h, o2)
        gbs.append((p2, n2)); wholes.append((o2, g.last_frame_rays()))
    enqueue = rt.RtHip.frame_multi_enqueue_device.__wrapped__ if hasattr(rt.RtHip.frame_multi_enqueue_device, "__wrapped__") else rt.RtHip.frame_multi_enqueue_device
    for streams in (1, 3):
        g.set_tuning(frame_streams=streams)
        outs = [torch.full((w * h * 4,), 7, dtype=torch.uint8, device="cuda") for _ in range(8)]
        torch.cuda.synchronize()
        got, pending = [], 0
        for k in range(8):
            if pending == max(2, streams):
                got.append(g.frame_finish()); pending -= 1
            enqueue(g, gbs[k % 4][0], gbs[k % 4][1], w, h, 64, outs[k]); pending += 1
        while pending:
            got.append(g.frame_finish()); pending -= 1
        g.sync()
        for k in range(8):
            assert torch.equal(outs[k], wholes[k % 4][0]), "streams %d, frame %d" % (streams, k)
            assert got[k] == wholes[k % 4][1]
    # several frames per pipeline entry (one pass of launches, one all-gather of all their chunks, an unpack per frame):
    # entries of 1, 3 and 4 frames, queued on one and on three lanes; an entry's rays are the sum of its frames'
    enqueue_batch = rt.RtHip.frames_multi_enqueue_device.__wrapped__ if hasattr(rt.RtHip.frames_multi_enqueue_device, "__wrapped__") else rt.RtHip.frames_multi_enqueue_device
    for streams in (1, 3):
        g.set_tuning(frame_streams=streams)
        entries = [[0], [1, 2, 3], [2, 0, 1, 3], [3, 3, 0], [1], [0, 1, 2, 3]]
        outs = [[torch.full((w * h * 4,), 7, dtype=torch.uint8, device="cuda") for _ in e] for e in entries]
        torch.cuda.synchronize()
        got, pending = [], 0
        for e, o in zip(entries, outs):
            if pending == max(2, streams):
                got.append(g.frame_finish()); pending -= 1
            enqueue_batch(g, [gbs[f][0] for f in e], [gbs[f][1] for f in e], w, h, 64 if len(e) != 3 else 16, o); pending += 1
        while pending:
            got.append(g.frame_finish()); pending -= 1
        g.sync()
        for k, (e, o) in enumerate(zip(entries, outs)):
            for f, img in zip(e, o):
                assert torch.equal(img, wholes[f][0]), "streams %d, entry %d, frame %d" % (streams, k, f)
            assert got[k] == sum(wholes[f][1] for f in e)
    with pytest.raises(rt.RtHipError, match="1..16 frames"):
        enqueue_batch(g, [d_p] * 17, [d_n] * 17, w, h, 64, [whole] * 17)
    g.set_tuning(frame_streams=1)
    again = torch.zeros_like(whole)
    g.render_frame_device(d_p, d_n, w, h, again)       # the single-GPU entry is undisturbed
    assert torch.equal(again, whole)
    # the photon pass
    per_light = 3000
    n1 = g.photon_generate(per_light)
    a = g.photon_get()
    n2 = g.photon_generate_multi(per_light)
    b = g.photon_get()
    assert n1 == n2 > per_light and a.tobytes() == b.tobytes()
    g.comm_destroy()
    g.comm_destroy()                  # idempotent
    g.close()


def test_batched_entry_that_could_overflow_is_refused_by_every_rank_alike():
    """An entry is refused up front (RT_HIP_ERR_LIMIT) when the LARGEST share of any rank, times its frames, could
    overflow the sync-free driver's level bounds (2^23 nodes) -- before anything is queued, so that no rank is left
    waiting in the all-gather; the pipeline stays usable.  One rank owns the whole 1080p frame: two fit, three do not."""
    import torch
    import raytracer_amd as rt
    sc = scenes.atrium(2048)
    g = rt.RtHip(sc, semantics="array", accel=rt.ACCEL_PLOC, bounces=2, do_shadows=1, shadow_t_max=1.0, seed=42)
    w, h = 1920, 1080
    d_n = torch.zeros(w * h * 4, dtype=torch.float32, device="cuda")
    d_p = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda")
    g.render_gbuffer_device(sc.fov, w, h, d_n, d_p)
    whole = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
    g.render_frame_device(d_p, d_n, w, h, whole)
    rays = g.last_frame_rays()
    g.comm_init(0, 1)
    outs = [torch.zeros_like(whole) for _ in range(3)]
    with pytest.raises(rt.RtHipError, match="cannot be enqueued"):
        g.frames_multi_enqueue_device([d_p] * 3, [d_n] * 3, w, h, 64, outs)
    with pytest.raises(rt.RtHipError, match="no frame is enqueued"):
        g.frame_finish()
    g.frames_multi_enqueue_device([d_p] * 2, [d_n] * 2, w, h, 64, outs[:2])
    assert g.frame_finish() == 2 * rays
    g.sync()
    assert torch.equal(outs[0], whole) and torch.equal(outs[1], whole)
    g.comm_destroy()
    g.close()


def test_torch_rccl_tile_all_gather_single_rank():
    """raytracer_amd.frames.TileAllGather (what bench.py --scaling strong uses under torch.distributed) with the nccl
    (= RCCL) backend and one rank: the collective runs on RCCL's stream, ordered against the context's stream by events
    through torch.cuda.ExternalStream -- no host synchronisation -- and the assembled frame equals the whole frame."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    import raytracer_amd as rt
    from raytracer_amd.frames import TileAllGather
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        sc = scenes.atrium(2048)
        g = rt.RtHip(sc, semantics="array", accel=rt.ACCEL_PLOC, bounces=2, do_shadows=1, shadow_t_max=1.0)
        w, h = 200, 150
        d_n = torch.zeros(w * h * 4, dtype=torch.float32, device=dev)
        d_p = torch.zeros(w * h * 3, dtype=torch.float32, device=dev)
        g.render_gbuffer_device(sc.fov, w, h, d_n, d_p)
        whole = torch.zeros(w * h * 4, dtype=torch.uint8, device=dev)
        g.render_frame_device(d_p, d_n, w, h, whole)
        x = TileAllGather(g, w, h, 64, 0, 1, dev)
        for _ in range(3):                       # back to back: nothing but stream order between the steps
            frame = torch.full((w * h * 4,), 9, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            x.render(d_p, d_n)
            x.gather(frame)
            g.sync()
            assert torch.equal(frame, whole)
        g.close()
    finally:
        dist.destroy_process_group()
