"""rt_hip_gbuffer_select_noalloc_device (the entry point that allocates nothing: fixed per-GPU scratch,
include/rt_hip.h) against rt_hip_gbuffer_select_device, through the C ABI: bit-identical outputs on two
calls in a row, the second with a smaller input (the scratch keeps the first call's contents), and the
size limit.  Collected last on purpose (pytest -x)."""
import numpy as np
import pytest

from raytracer_amd import scenes

pytestmark = pytest.mark.gpu


def _select_both(g, torch, d_n, d_p, n, interlace, phase):
    out = []
    for fn in (g.gbuffer_select_device, g.gbuffer_select_noalloc_device):
        o_p = torch.full((n * 3,), -7.0, dtype=torch.float32, device="cuda")
        o_n = torch.full((n * 3,), -7.0, dtype=torch.float32, device="cuda")
        o_i = torch.full((n,), -7, dtype=torch.int32, device="cuda")
        cnt = fn(d_n, d_p, n, interlace, phase, o_p, o_n, o_i)
        out.append((cnt, o_p.cpu().numpy(), o_n.cpu().numpy(), o_i.cpu().numpy()))
    return out


def _assert_same(a, b, what):
    assert a[0] == b[0], "%s: count %d vs %d" % (what, a[0], b[0])
    for x, y, name in zip(a[1:], b[1:], ("positions", "normals", "index")):
        # the whole arrays, i.e. also: nothing is written behind the count
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), "%s: %s differ" % (what, name)


def test_gbuffer_select_noalloc_matches_allocating_entry(oracle_cls):
    import torch
    import raytracer_amd as rt
    assert torch.cuda.is_available()
    sc = scenes.sphere_field()   # plenty of empty texels
    g = rt.RtHip(sc, geometry=rt.SPHERES)
    w, h = 640, 480              # 150 tiles of 2048 texels
    n = w * h
    d_n = torch.zeros(n * 4, dtype=torch.float32, device="cuda")
    d_p = torch.zeros(n * 3, dtype=torch.float32, device="cuda")
    g.render_gbuffer_device(sc.fov, w, h, d_n, d_p)
    covered = int((d_n.view(-1, 4)[:, 3] > 0).sum().item())
    assert 0 < covered < n

    # first call: the whole G-buffer
    a, b = _select_both(g, torch, d_n, d_p, n, 1, 0)
    assert a[0] == covered
    _assert_same(a, b, "full")
    # second call, smaller input, straight after: a prefix that is not a multiple of the tile, other phase rule
    alpha = (d_n.view(-1, 4)[:, 3] > 0).cpu().numpy()
    n2 = int(np.searchsorted(np.cumsum(alpha), covered // 3)) + 77   # about a third of the covered texels
    assert n2 < n and n2 % 2048 != 0
    a2, b2 = _select_both(g, torch, d_n, d_p, n2, 16, 5)
    assert 0 < a2[0] < a[0]
    _assert_same(a2, b2, "prefix, interlace 16")
    # and against the oracle's restatement of getRayTraceData's selection
    nr = d_n.cpu().numpy().reshape(-1, 4)[:n2]
    po = d_p.cpu().numpy().reshape(-1, 3)[:n2]
    ep, en, ei = oracle_cls(sc).gbuffer_select(nr, po, 16, 5)
    assert len(ei) == b2[0]
    np.testing.assert_array_equal(ei, b2[3][:b2[0]])
    assert np.array_equal(ep.view(np.uint32).ravel(), b2[1][:3 * b2[0]].view(np.uint32))
    assert np.array_equal(en.view(np.uint32).ravel(), b2[2][:3 * b2[0]].view(np.uint32))
    # a third, larger again (the scratch now holds the small call's counts)
    a3, b3 = _select_both(g, torch, d_n, d_p, n, 16, 0)
    _assert_same(a3, b3, "full, interlace 16")
    # a second context on the same GPU shares the array
    g2 = rt.RtHip(sc, geometry=rt.SPHERES)
    a4, b4 = _select_both(g2, torch, d_n, d_p, 4099, 1, 0)
    _assert_same(a4, b4, "second context")
    g2.close()

    # the count left on the device: no copy, no wait -- the form a capturing stream takes
    o_p = torch.full((n * 3,), -7.0, dtype=torch.float32, device="cuda")
    o_n = torch.full((n * 3,), -7.0, dtype=torch.float32, device="cuda")
    o_i = torch.full((n,), -7, dtype=torch.int32, device="cuda")
    d_cnt = torch.full((1,), -1, dtype=torch.int32, device="cuda")
    assert g.gbuffer_select_noalloc_device(d_n, d_p, n, 16, 0, o_p, o_n, o_i, d_count=d_cnt) is None
    g.sync()
    _assert_same((int(d_cnt.item()), o_p.cpu().numpy(), o_n.cpu().numpy(), o_i.cpu().numpy()), a3, "count on the device")
    # ... and actually captured into a HIP graph on the context's stream, then replayed on a changed G-buffer
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    stream = ctypes.c_void_p(g.stream())
    graph, gexec = ctypes.c_void_p(), ctypes.c_void_p()
    d_n2 = d_n.clone()
    o_p.fill_(-7.0); o_n.fill_(-7.0); o_i.fill_(-7); d_cnt.fill_(-1)
    torch.cuda.synchronize()
    assert hip.hipStreamBeginCapture(stream, 2) == 0          # hipStreamCaptureModeRelaxed
    # (the binding's un-fenced form: its wrapper would synchronise the stream around the C call)
    type(g).gbuffer_select_noalloc_device.__wrapped__(g, d_n2, d_p, n, 16, 0, o_p, o_n, o_i, d_count=d_cnt)
    assert hip.hipStreamEndCapture(stream, ctypes.byref(graph)) == 0, "the call is not legal under capture"
    assert hip.hipGraphInstantiate(ctypes.byref(gexec), graph, None, None, 0) == 0
    assert int(d_cnt.item()) == -1                             # captured, not executed
    d_n2.view(-1, 4)[: n // 2, 3] = 0.0                        # the replay sees the buffer's contents at launch time
    torch.cuda.synchronize()
    assert hip.hipGraphLaunch(gexec, stream) == 0
    g.sync()
    ref_cnt = g.gbuffer_select_device(d_n2, d_p, n, 16, 0, *(torch.empty_like(t) for t in (o_p, o_n, o_i)))
    assert int(d_cnt.item()) == ref_cnt and 0 < ref_cnt < a3[0]
    hip.hipGraphExecDestroy(gexec); hip.hipGraphDestroy(graph)

    # empty input and the documented limit
    assert g.gbuffer_select_noalloc_device(d_n, d_p, 0, 1, 0, d_p, d_p, d_p) == 0
    with pytest.raises(rt.RtHipError, match="exceed"):
        g.gbuffer_select_noalloc_device(d_n, d_p, rt.NOALLOC_MAX_TEXELS + 1, 1, 0, d_p, d_p, d_p)
    g.close()
