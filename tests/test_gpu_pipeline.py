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
