"""rt_hip_frame_enqueue_device / rt_hip_frame_finish: frames queued ahead of the host's read-back render the same images
and report the same ray counts as the call-per-frame entry, in order."""
import numpy as np
import pytest

from raytracer_amd import scenes

pytestmark = pytest.mark.gpu


def test_pipelined_frames_equal_sequential_frames():
    import torch
    import raytracer_amd as rt
    sc = scenes.atrium(8192)
    g = rt.RtHip(sc, semantics="array", accel=rt.ACCEL_PLOC, bounces=2, do_shadows=1, shadow_t_max=1.0, profile=1)
    w, h = 320, 200
    gb, exp = [], []
    for f in range(4):
        d_n = torch.zeros(w * h * 4, dtype=torch.float32, device="cuda")
        d_p = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda")
        g.render_gbuffer_device(sc.fov - 0.05 * f, w, h, d_n, d_p)
        out = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
        g.render_frame_device(d_p, d_n, w, h, out)
        gb.append((d_p, d_n))
        exp.append((out, g.last_frame_rays()))
    assert len({r for _, r in exp}) > 1            # the frames differ
    with pytest.raises(rt.RtHipError, match="no frame is enqueued"):
        g.frame_finish()
    g.reset_stats()
    enqueue = rt.RtHip.frame_enqueue_device.__wrapped__     # the un-fenced binding: returns while the GPU still works
    outs = [torch.full((w * h * 4,), 3, dtype=torch.uint8, device="cuda") for _ in range(4)]
    torch.cuda.synchronize()
    got = []
    enqueue(g, gb[0][0], gb[0][1], w, h, outs[0])
    enqueue(g, gb[1][0], gb[1][1], w, h, outs[1])
    with pytest.raises(rt.RtHipError, match="already enqueued"):
        enqueue(g, gb[2][0], gb[2][1], w, h, outs[2])
    with pytest.raises(rt.RtHipError, match="enqueued"):
        g.render_frame_device(gb[2][0], gb[2][1], w, h, outs[2])
    got.append(g.frame_finish())                            # frame 0
    enqueue(g, gb[2][0], gb[2][1], w, h, outs[2])
    got.append(g.frame_finish())                            # frame 1
    enqueue(g, gb[3][0], gb[3][1], w, h, outs[3])
    got.append(g.frame_finish())
    got.append(g.frame_finish())
    g.sync()
    for f in range(4):
        assert got[f] == exp[f][1], "rays of frame %d" % f
        assert torch.equal(outs[f], exp[f][0]), "image of frame %d" % f
    st = g.stats()
    assert st["frames"] == 4 and st["any_launches"] == 12 and st["any_ms"] > 0 and st["frame_ms"] > 0
    assert st["rays_any"] + st["rays_closest"] == sum(r for _, r in exp)
    # a frame the sync-free driver does not take (forced here) is rendered inside the enqueue call
    g.set_tuning(frame_sync=1)
    o2 = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
    g.frame_enqueue_device(gb[1][0], gb[1][1], w, h, o2)
    assert g.frame_finish() == exp[1][1] and torch.equal(o2, exp[1][0])
    g.close()


def test_host_pipeline_equals_host_frames():
    """rt_hip_frame_enqueue (pinned host buffers, copies on their own streams) against rt_hip_render_frame on the same
    G-buffers: same images, same ray counts, frames finished in order, mixed with the device-pointer entry."""
    import torch
    import raytracer_amd as rt
    sc = scenes.atrium(8192)
    g = rt.RtHip(sc, semantics="array", accel=rt.ACCEL_PLOC, bounces=2, do_shadows=1, shadow_t_max=1.0)
    w, h = 320, 200
    host, exp = [], []
    for f in range(5):
        d_n = torch.zeros(w * h * 4, dtype=torch.float32, device="cuda")
        d_p = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda")
        g.render_gbuffer_device(sc.fov - 0.04 * f, w, h, d_n, d_p)
        g.sync()
        h_p, h_n = d_p.cpu().pin_memory(), d_n.cpu().pin_memory()
        img = g.render_frame(h_p.numpy().reshape(-1, 3), h_n.numpy().reshape(-1, 4), w, h)
        host.append((h_p, h_n, d_p, d_n))
        exp.append((np.asarray(img).reshape(-1).copy(), g.last_frame_rays()))
    outs = [torch.full((w * h * 4,), 7, dtype=torch.uint8).pin_memory() for _ in range(5)]
    d_out = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
    got = []
    g.frame_enqueue_host(host[0][0], host[0][1], w, h, outs[0])
    g.frame_enqueue_host(host[1][0], host[1][1], w, h, outs[1])
    with pytest.raises(rt.RtHipError, match="already enqueued"):
        g.frame_enqueue_host(host[2][0], host[2][1], w, h, outs[2])
    got.append(g.frame_finish())
    assert np.array_equal(outs[0].numpy(), exp[0][0])        # complete when its finish returns, while frame 1 may still run
    g.frame_enqueue_host(host[2][0], host[2][1], w, h, outs[2])
    got.append(g.frame_finish())
    rt.RtHip.frame_enqueue_device.__wrapped__(g, host[3][2], host[3][3], w, h, d_out)   # a device-pointer frame in between
    got.append(g.frame_finish())
    g.frame_enqueue_host(host[4][0], host[4][1], w, h, outs[4])
    got.append(g.frame_finish())
    got.append(g.frame_finish())
    g.sync()
    assert got == [r for _, r in exp]
    for f in (0, 1, 2, 4):
        assert np.array_equal(outs[f].numpy(), exp[f][0]), "image of frame %d" % f
    assert np.array_equal(d_out.cpu().numpy(), exp[3][0])
    # pageable memory works too (no overlap, same result)
    plain = np.full(w * h * 4, 9, np.uint8)
    pos_copy, nrm_copy = host[1][0].numpy().copy(), host[1][1].numpy().copy()   # alive until frame_finish, as the contract asks
    g.frame_enqueue_host(pos_copy, nrm_copy, w, h, plain)
    assert g.frame_finish() == exp[1][1] and np.array_equal(plain, exp[1][0])
    del pos_copy, nrm_copy
    g.close()


def test_g16_pipeline_is_bit_identical_to_the_float_entry():
    """rt_hip_frame_enqueue_g16: the G-buffer in the FBO's own formats (RGBA16 UNORM normals, RGB16F positions:
    ras_rasterizer.cpp:293,301), widened on the device the way glReadPixels(GL_FLOAT) widens it (:497,543) -- c / 65535
    and half -> float, both exact.  The 16-bit planes here are made from the float G-buffer, the round trip is checked to
    be lossless on the host first, and the frames must equal rt_hip_render_frame's on the float planes bit for bit."""
    import torch
    import raytracer_amd as rt
    sc = scenes.atrium(8192)
    g = rt.RtHip(sc, semantics="array", accel=rt.ACCEL_PLOC, bounces=2, do_shadows=1, shadow_t_max=1.0)
    w, h = 320, 200
    frames = []
    for f in range(3):
        d_n = torch.zeros(w * h * 4, dtype=torch.float32, device="cuda")
        d_p = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda")
        g.render_gbuffer_device(sc.fov - 0.05 * f, w, h, d_n, d_p)
        g.sync()
        nr, po = d_n.cpu().numpy().reshape(-1, 4), d_p.cpu().numpy().reshape(-1, 3)
        n16 = np.empty((w * h, 4), np.uint16)
        n16[:, :3] = np.rint(nr[:, :3].astype(np.float64) * 65535.0).astype(np.uint16)
        n16[:, 3] = nr[:, 3].astype(np.uint16)                      # material id + 1
        p16 = po.astype(np.float16)
        # the 16-bit planes hold the G-buffer exactly (it was quantised to these formats when it was made)
        assert np.array_equal((n16[:, :3].astype(np.float32) / np.float32(65535.0)).view(np.uint32), nr[:, :3].view(np.uint32))
        assert np.array_equal(p16.astype(np.float32).view(np.uint32), po.view(np.uint32))
        img = g.render_frame(po, nr, w, h)
        frames.append((torch.from_numpy(n16).pin_memory(), torch.from_numpy(p16.view(np.uint16)).pin_memory(),
                       np.asarray(img).reshape(-1).copy(), g.last_frame_rays()))
    outs = [torch.full((w * h * 4,), 7, dtype=torch.uint8).pin_memory() for _ in range(3)]
    g.frame_enqueue_host_g16(frames[0][0], frames[0][1], w, h, outs[0])
    g.frame_enqueue_host_g16(frames[1][0], frames[1][1], w, h, outs[1])
    rays = [g.frame_finish()]
    g.frame_enqueue_host_g16(frames[2][0], frames[2][1], w, h, outs[2])
    rays += [g.frame_finish(), g.frame_finish()]
    assert rays == [f[3] for f in frames]
    for k in range(3):
        assert np.array_equal(outs[k].numpy(), frames[k][2]), "frame %d" % k
    assert int(outs[0].numpy().reshape(-1, 4)[:, :3].max()) > 40
    g.close()


@pytest.mark.parametrize("streams", [2, 3, 4])
def test_frames_on_several_streams_equal_sequential_frames(streams):
    """tuning.frame_streams > 1: enqueued frames are dealt to that many lanes (a stream and level arrays each) and run side by
    side.  Images and ray counts must be the sequential ones, in order, for device frames, for frames from host buffers
    (float and 16-bit G-buffer), also with a shallow LDS stack (every lane needs its own overflow columns in HBM) and with
    the shadow rays' shared descent forced on."""
    import torch
    import raytracer_amd as rt
    sc = scenes.atrium(8192)
    g = rt.RtHip(sc, semantics="array", accel=rt.ACCEL_PLOC, bounces=2, do_shadows=1, shadow_t_max=1.0, profile=1,
                 tuning=dict(lds_stack_depth=3, shadow_prefix_min_rays=0))
    w, h = 320, 200
    nf = 7
    gb, exp, host = [], [], []
    for f in range(nf):
        d_n = torch.zeros(w * h * 4, dtype=torch.float32, device="cuda")
        d_p = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda")
        g.render_gbuffer_device(sc.fov - 0.04 * f, w, h, d_n, d_p)
        out = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
        g.render_frame_device(d_p, d_n, w, h, out)
        gb.append((d_p, d_n))
        exp.append((out.cpu().numpy(), g.last_frame_rays()))
        host.append((d_p.cpu().pin_memory(), d_n.cpu().pin_memory()))
    assert len({r for _, r in exp}) > 1
    g.set_tuning(frame_streams=streams)
    assert g.tuning()["frame_streams"] == streams
    enqueue = rt.RtHip.frame_enqueue_device.__wrapped__
    outs = [torch.full((w * h * 4,), 3, dtype=torch.uint8, device="cuda") for _ in range(nf)]
    torch.cuda.synchronize()
    got, pending = [], 0
    for f in range(nf):
        if pending == max(2, streams):
            with pytest.raises(rt.RtHipError, match="already enqueued"):
                enqueue(g, gb[f][0], gb[f][1], w, h, outs[f])
            got.append(g.frame_finish()); pending -= 1
        enqueue(g, gb[f][0], gb[f][1], w, h, outs[f]); pending += 1
    while pending:
        got.append(g.frame_finish()); pending -= 1
    g.sync()
    assert got == [r for _, r in exp]
    for f in range(nf):
        assert np.array_equal(outs[f].cpu().numpy(), exp[f][0]), "image of frame %d" % f
    # host buffers: uploads, frames on their lanes, downloads
    h_outs = [torch.full((w * h * 4,), 5, dtype=torch.uint8).pin_memory() for _ in range(nf)]
    got, pending = [], 0
    for f in range(nf):
        if pending == max(2, streams):
            got.append(g.frame_finish()); pending -= 1
        g.frame_enqueue_host(host[f][0], host[f][1], w, h, h_outs[f]); pending += 1
    while pending:
        got.append(g.frame_finish()); pending -= 1
    assert got == [r for _, r in exp]
    for f in range(nf):
        assert np.array_equal(h_outs[f].numpy(), exp[f][0]), "host image of frame %d" % f
    # the other entries still work between pipelines and see the same scene
    again = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
    g.render_frame_device(gb[2][0], gb[2][1], w, h, again)
    assert np.array_equal(again.cpu().numpy(), exp[2][0])
    g.close()


def test_packed_shares_through_the_pipeline_equal_the_blocking_entry():
    """rt_hip_tiles_packed_enqueue_device: every virtual rank's packed share of a frame, three frames in flight on three
    lanes, equals what rt_hip_render_tiles_packed_device writes; unpacked, the chunks are the whole frame."""
    import torch
    import raytracer_amd as rt
    from raytracer_amd.tiles import max_owned_tiles
    sc = scenes.atrium(8192)
    g = rt.RtHip(sc, semantics="array", accel=rt.ACCEL_PLOC, bounces=2, do_shadows=1, shadow_t_max=1.0)
    w, h, tile, nranks = 320, 200, 16, 3
    d_n = torch.zeros(w * h * 4, dtype=torch.float32, device="cuda")
    d_p = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda")
    g.render_gbuffer_device(sc.fov, w, h, d_n, d_p)
    whole = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
    g.render_frame_device(d_p, d_n, w, h, whole)
    chunk = max_owned_tiles(w, h, tile, nranks) * tile * tile * 4
    ref = torch.zeros(nranks * chunk, dtype=torch.uint8, device="cuda")
    rays = []
    for r in range(nranks):
        g.render_tiles_packed_device([d_p], [d_n], w, h, tile, r, nranks, [ref[r * chunk:(r + 1) * chunk]])
        rays.append(g.last_frame_rays())
    g.set_tuning(frame_streams=3)
    enq = rt.RtHip.tiles_packed_enqueue_device.__wrapped__
    got = torch.zeros(nranks * chunk, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for r in range(nranks):                                   # the three ranks' shares side by side
        enq(g, d_p, d_n, w, h, tile, r, nranks, got[r * chunk:(r + 1) * chunk])
    assert [g.frame_finish() for _ in range(nranks)] == rays
    g.sync()
    assert torch.equal(got, ref)
    out = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
    g.unpack_chunks_device(got, chunk, w, h, tile, nranks, out)
    assert torch.equal(out, whole)
    with pytest.raises(rt.RtHipError):
        enq(g, d_p, d_n, w, h, tile, 0, 1, got[:chunk])       # one rank: nothing to pack
    # entries of several frames (rt_hip_tiles_packed_batch_enqueue_device): a second frame beside the first, every rank's
    # entry of (frame A, frame B, frame A) side by side on the three lanes
    n2 = torch.zeros_like(d_n); p2 = torch.zeros_like(d_p)
    g.render_gbuffer_device(sc.fov - 0.1, w, h, n2, p2)
    ref2 = torch.zeros(nranks * chunk, dtype=torch.uint8, device="cuda")
    rays2 = []
    g.set_tuning(frame_streams=1)
    for r in range(nranks):
        g.render_tiles_packed_device([p2], [n2], w, h, tile, r, nranks, [ref2[r * chunk:(r + 1) * chunk]])
        rays2.append(g.last_frame_rays())
    g.set_tuning(frame_streams=3)
    enqb = rt.RtHip.tiles_packed_batch_enqueue_device.__wrapped__
    gots = [[torch.zeros(chunk, dtype=torch.uint8, device="cuda") for _ in range(3)] for _ in range(nranks)]
    torch.cuda.synchronize()
    for r in range(nranks):
        enqb(g, [d_p, p2, d_p], [d_n, n2, d_n], w, h, tile, r, nranks, gots[r])
    assert [g.frame_finish() for _ in range(nranks)] == [2 * rays[r] + rays2[r] for r in range(nranks)]
    g.sync()
    for r in range(nranks):
        assert torch.equal(gots[r][0], ref[r * chunk:(r + 1) * chunk]) and torch.equal(gots[r][2], ref[r * chunk:(r + 1) * chunk])
        assert torch.equal(gots[r][1], ref2[r * chunk:(r + 1) * chunk])
    g.close()
