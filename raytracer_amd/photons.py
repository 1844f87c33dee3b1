"""Host side of the sharded photon pass (collective #1).

PhotonMap::generate (rt_photon_map.cpp:86-124) is one sequential loop over lights x count photons.
Here every rank emits a contiguous range of the global photon index sequence
(`tiles.photon_shard`), the counter RNG is keyed by (seed, index), and each rank's stored photons
come out in index order.  Concatenating the shards in rank order therefore reproduces the single-GPU
photon array bit for bit; that concatenation is the only exchange step of the path: a variable-length
all-gather (RCCL on the GPU box, gloo in the CPU tests).  Every rank then installs the full set with
rt_hip_photon_set_device, which builds the same grid everywhere.
"""
import torch
import torch.distributed as dist

from .tiles import photon_shard

PHOTON_FLOATS = 9  # position, energy, direction


def gather_photons(local: torch.Tensor, group=None) -> torch.Tensor:
    """All-gather of (k_r, 9) float32 photon arrays, ranks in order, EXACT sizes -- the same exchange as the C route
    (rt_hip_photon_generate_multi, k_comm.hip): an all-gather of the counts (one int64 per rank), then every rank
    sends its shard to every peer and receives each peer's shard straight into its place in the output, as one batch
    of point-to-point operations (RCCL: one grouped ncclSend / ncclRecv set, each xGMI link carries one shard; gloo in
    the CPU tests).  Nothing is padded: a rank moves (world - 1) x its own shard out and the other shards in."""
    assert local.dim() == 2 and local.shape[1] == PHOTON_FLOATS and local.dtype == torch.float32
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    # the nccl (RCCL) backend needs device tensors, gloo needs host tensors: keep `local`'s device
    counts = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device), group=group)
    counts = [int(c.item()) for c in counts]
    offsets = [0]
    for c in counts:
        offsets.append(offsets[-1] + c)
    out = torch.empty((offsets[-1], PHOTON_FLOATS), dtype=torch.float32, device=local.device)
    local = local.contiguous()
    out[offsets[rank]:offsets[rank + 1]] = local
    ops = []
    for peer in range(world):
        if peer == rank:
            continue
        gpeer = dist.get_global_rank(group, peer) if group is not None else peer
        if counts[rank] > 0:
            ops.append(dist.P2POp(dist.isend, local, gpeer, group))
        if counts[peer] > 0:
            ops.append(dist.P2POp(dist.irecv, out[offsets[peer]:offsets[peer + 1]], gpeer, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out


def build_photon_map(ctx, per_light: int, num_lights: int, device, group=None):
    """Emit this rank's shard on its GPU, all-gather, install the full set on `ctx` (an RtHip).
    Returns the number of photons in the map.  A photon is stored at most once
    (rt_photon_map.cpp:44-52), so the shard's output buffer holds `count` photons."""
    rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    first, count = photon_shard(per_light * num_lights, rank, world)
    cap = max(count, 1)
    buf = torch.empty((cap, PHOTON_FLOATS), dtype=torch.float32, device=device)
    stored = ctx.photon_emit_device(per_light, first, count, buf, cap)
    full = gather_photons(buf[:stored], group)
    ctx.photon_set_device(full, full.shape[0])
    return int(full.shape[0])
