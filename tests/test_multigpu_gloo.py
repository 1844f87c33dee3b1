"""The N>1 path on CPU: two gloo ranks shade their interleaved tiles (with the oracle standing in
for the GPU) and assemble the frames exactly as bench.py does (packed tiles, one all-to-all, frame f
assembled on rank f); the result must equal the single-rank frames bit for bit.  Also the older
all_reduce form, the photon index shards and the photon all-gather."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracer_amd import scenes
from raytracer_amd.tiles import max_owned_tiles, owned_tiles, pack_tiles, photon_shard, tile_owner, unpack_tiles

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


@pytest.mark.parametrize("world", [2, 4])
def test_photon_gather_equals_single_rank(tmp_path, world):
    from oracle.pyoracle import Oracle
    sc = scenes.cornell()
    whole = Oracle(sc).photon_generate(PER_LIGHT, rng=1, seed=42)
    assert whole.shape[0] > 100
    mp.spawn(_photon_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "photons_%d.npy" % r))
        assert got.tobytes() == whole.tobytes()          # every rank holds the single-rank array, bit for bit
        assert np.load(os.path.join(str(tmp_path), "empty_%d.npy" % r)).shape == (0, 9)


# ---- collective #2 as bench.py runs it: packed tiles, all-to-all, frame f assembled on rank f --------
W2, H2, TILE2 = 100, 75, 16          # not multiples of the tile: edge tiles carry padding


def _frame_inputs_fov(fov_delta):
    from oracle.pyoracle import Oracle
    sc = scenes.cornell()
    orc = Oracle(sc)
    nr, po = orc.software_gbuffer(W2, H2, fov=sc.fov + fov_delta)
    mid = nr[:, 3].astype(np.int32) - 1
    nrm = (nr[:, :3].astype(np.float64) * 2.0 - 1.0).astype(np.float32)
    return orc, po, nrm, mid


def _a2a_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    owner = tile_owner(W2, H2, TILE2, world)
    mine = owner == rank
    lists = [owned_tiles(W2, H2, TILE2, r, world) for r in range(world)]
    chunk = max_owned_tiles(W2, H2, TILE2, world) * TILE2 * TILE2 * 4
    send = torch.zeros(world * chunk, dtype=torch.uint8)
    for f in range(world):                                         # a step = `world` frames
        orc, po, nrm, mid = _frame_inputs_fov(-0.03 * f)
        rgba, _, _ = orc.render_texels(po, nrm, np.where(mine, mid, -1), bounces=2, do_shadows=1, shadow_t_max=1.0)
        tiles = np.where(mine[:, None], rgba, 7).astype(np.uint8)  # 7 = "untouched" pixels of other ranks: must not travel
        packed = pack_tiles(tiles, W2, H2, TILE2, lists[rank]).reshape(-1)
        send[f * chunk: f * chunk + packed.shape[0]] = torch.from_numpy(packed)
    recv = torch.zeros_like(send)
    dist.all_to_all_single(recv, send)
    frame = np.full((W2 * H2, 4), 9, np.uint8)
    for r in range(world):
        unpack_tiles(recv[r * chunk:(r + 1) * chunk].numpy(), W2, H2, TILE2, lists[r], frame)
    dist.barrier()
    np.save(os.path.join(out_dir, "frame_%d.npy" % rank), frame)
    dist.destroy_process_group()


def test_owned_tiles_match_tile_owner():
    for n in (1, 2, 3, 8):
        owner = tile_owner(W2, H2, TILE2, n).reshape(H2, W2)
        tiles_x = -(-W2 // TILE2)
        seen = []
        for r in range(n):
            ids = owned_tiles(W2, H2, TILE2, r, n)
            assert list(ids) == sorted(ids)
            for t in ids:
                ty, tx = divmod(int(t), tiles_x)
                assert owner[ty * TILE2, tx * TILE2] == r
            seen += list(ids)
        assert sorted(seen) == list(range(tiles_x * -(-H2 // TILE2)))


def test_pack_unpack_roundtrip():
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (W2 * H2, 4)).astype(np.uint8)
    out = np.zeros_like(img)
    for r in range(3):
        ids = owned_tiles(W2, H2, TILE2, r, 3)
        unpack_tiles(pack_tiles(img, W2, H2, TILE2, ids), W2, H2, TILE2, ids, out)
    np.testing.assert_array_equal(out, img)


def _allgather_worker(rank, world, port, out_dir):
    """Strong scaling (bench.py --scaling strong, rt_hip_render_frame_multi_device): ONE frame split over the ranks, the
    packed tiles all-gathered, the whole frame assembled on EVERY rank."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    owner = tile_owner(W2, H2, TILE2, world)
    mine = owner == rank
    lists = [owned_tiles(W2, H2, TILE2, r, world) for r in range(world)]
    chunk = max_owned_tiles(W2, H2, TILE2, world) * TILE2 * TILE2 * 4
    orc, po, nrm, mid = _frame_inputs_fov(0.0)
    rgba, _, rays = orc.render_texels(po, nrm, np.where(mine, mid, -1), bounces=2, do_shadows=1, shadow_t_max=1.0)
    tiles = np.where(mine[:, None], rgba, 7).astype(np.uint8)
    send = torch.zeros(chunk, dtype=torch.uint8)
    packed = pack_tiles(tiles, W2, H2, TILE2, lists[rank]).reshape(-1)
    send[:packed.shape[0]] = torch.from_numpy(packed)
    recv = torch.zeros(world * chunk, dtype=torch.uint8)
    dist.all_gather_into_tensor(recv, send)
    frame = np.full((W2 * H2, 4), 9, np.uint8)
    for r in range(world):
        unpack_tiles(recv[r * chunk:(r + 1) * chunk].numpy(), W2, H2, TILE2, lists[r], frame)
    total = torch.tensor([rays], dtype=torch.int64)
    dist.all_reduce(total, op=dist.ReduceOp.SUM)
    dist.barrier()
    np.save(os.path.join(out_dir, "strong_%d.npy" % rank), frame)
    if rank == 0:
        np.save(os.path.join(out_dir, "strong_rays.npy"), total.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_strong_scaling_all_gather_assembles_the_frame_on_every_rank(tmp_path, world):
    mp.spawn(_allgather_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    orc, po, nrm, mid = _frame_inputs_fov(0.0)
    full, _, rays = orc.render_texels(po, nrm, mid, bounces=2, do_shadows=1, shadow_t_max=1.0)
    for r in range(world):
        np.testing.assert_array_equal(np.load(os.path.join(str(tmp_path), "strong_%d.npy" % r)), full)
    assert int(np.load(os.path.join(str(tmp_path), "strong_rays.npy"))[0]) == rays      # the ranks' rays add up to the frame's


@pytest.mark.parametrize("world", [2, 4])
def test_packed_exchange_assembles_frame_f_on_rank_f(tmp_path, world):
    mp.spawn(_a2a_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for f in range(world):
        orc, po, nrm, mid = _frame_inputs_fov(-0.03 * f)
        full, _, _ = orc.render_texels(po, nrm, mid, bounces=2, do_shadows=1, shadow_t_max=1.0)
        got = np.load(os.path.join(str(tmp_path), "frame_%d.npy" % f))
        np.testing.assert_array_equal(got, full)
