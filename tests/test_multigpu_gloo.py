"""The N>1 path on CPU: two gloo ranks shade their interleaved tiles (with the oracle standing in
for the GPU) and assemble the frame exactly as bench.py does (all_reduce SUM of disjoint RGBA8 tile
images); the result must equal the single-rank frame bit for bit.  Also the photon index shards."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracer_amd import scenes
from raytracer_amd.tiles import photon_shard, tile_owner

W, H, TILE = 96, 64, 16


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _frame_inputs():
    from oracle.pyoracle import Oracle
    sc = scenes.cornell()
    orc = Oracle(sc)
    nr, po = orc.software_gbuffer(W, H)
    mid = nr[:, 3].astype(np.int32) - 1
    nrm = (nr[:, :3].astype(np.float64) * 2.0 - 1.0).astype(np.float32)
    return orc, po, nrm, mid


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc, po, nrm, mid = _frame_inputs()
    owner = tile_owner(W, H, TILE, world)
    mine = owner == rank
    rgba, _, rays = orc.render_texels(po, nrm, np.where(mine, mid, -1), bounces=2, do_shadows=1, shadow_t_max=1.0)
    tiles = np.where(mine[:, None], rgba, 0).astype(np.uint8)      # other ranks' texels stay 0
    frame = torch.from_numpy(tiles.copy())
    dist.all_reduce(frame, op=dist.ReduceOp.SUM)                   # disjoint tiles: SUM assembles
    total_rays = torch.tensor([rays], dtype=torch.int64)
    dist.all_reduce(total_rays, op=dist.ReduceOp.SUM)
    dist.barrier()
    if rank == 0:
        np.save(os.path.join(out_dir, "frame.npy"), frame.numpy())
        np.save(os.path.join(out_dir, "rays.npy"), total_rays.numpy())
    dist.destroy_process_group()


def test_tile_owner_partitions_the_screen():
    for n in (1, 2, 4, 8):
        o = tile_owner(1920, 1080, 64, n)
        assert o.min() == 0 and o.max() == n - 1
        counts = np.bincount(o, minlength=n)
        assert counts.sum() == 1920 * 1080
        assert counts.max() / counts.min() < 1.15      # interleaving balances the load


def test_photon_shards_cover_the_sequence():
    for total in (10, 10000, 10_000_000):
        for n in (1, 2, 3, 8):
            spans = [photon_shard(total, r, n) for r in range(n)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == total
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1


def test_two_rank_frame_equals_single_rank(tmp_path):
    orc, po, nrm, mid = _frame_inputs()
    full, _, rays = orc.render_texels(po, nrm, mid, bounces=2, do_shadows=1, shadow_t_max=1.0)
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), "frame.npy"))
    np.testing.assert_array_equal(got, full)
    assert int(np.load(os.path.join(str(tmp_path), "rays.npy"))[0]) == rays


# ---- collective #1: the photon all-gather ---------------------------------------------------------
PER_LIGHT = 600


def _photon_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.pyoracle import Oracle
    from raytracer_amd.photons import gather_photons
    sc = scenes.cornell()
    orc = Oracle(sc)
    first, count = photon_shard(PER_LIGHT * len(sc.lights), rank, world)
    local = orc.photon_generate_range(PER_LIGHT, first, count, seed=42)   # stands in for the GPU emitter
    full = gather_photons(torch.from_numpy(local.copy()))
    empty = gather_photons(torch.zeros((0, 9), dtype=torch.float32))           # every shard empty
    dist.barrier()
    np.save(os.path.join(out_dir, "photons_%d.npy" % rank), full.numpy())
    np.save(os.path.join(out_dir, "empty_%d.npy" % rank), empty.numpy())
    dist.destroy_process_group()


def test_photon_range_matches_full_sequence():
    from oracle.pyoracle import Oracle
    sc = scenes.cornell()
    orc = Oracle(sc)
    whole = orc.photon_generate(PER_LIGHT, rng=1, seed=42)
    total = PER_LIGHT * len(sc.lights)
    assert np.array_equal(orc.photon_generate_range(PER_LIGHT, 0, total, seed=42), whole)
    parts = [orc.photon_generate_range(PER_LIGHT, *photon_shard(total, r, 3), seed=42) for r in range(3)]
    assert np.array_equal(np.concatenate(parts, 0), whole)


def test_two_rank_photon_gather_equals_single_rank(tmp_path):
    from oracle.pyoracle import Oracle
    sc = scenes.cornell()
    whole = Oracle(sc).photon_generate(PER_LIGHT, rng=1, seed=42)
    assert whole.shape[0] > 100
    mp.spawn(_photon_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        got = np.load(os.path.join(str(tmp_path), "photons_%d.npy" % r))
        assert got.tobytes() == whole.tobytes()          # every rank holds the single-rank array, bit for bit
        assert np.load(os.path.join(str(tmp_path), "empty_%d.npy" % r)).shape == (0, 9)
