"""Host side of the tile gather (collective #2).

With N GPUs a step renders N frames; every frame is split over the ranks by interleaved tiles
(`tiles.tile_owner`), and frame f is assembled on rank f.  Each rank shades its tiles of frame f straight into one
contiguous chunk (`rt_hip_render_tiles_packed_device`; `rt_hip_pack_tiles_device` does the same from a
full-size image), one all-to-all moves chunk f to rank f, and the receiver scatters the N chunks
into its frame with one kernel (`rt_hip_unpack_chunks_device`).  Only owned pixels
travel: (N-1)/N of one frame per rank and step (7.3 MB at 1080p, N = 8), not N full frames as an
all-reduce of sparse images would move.  RCCL over xGMI on the GPU box, gloo in the CPU tests (where
`tiles.pack_tiles` / `unpack_tiles` stand in for the kernels).
"""
import numpy as np
import torch
import torch.distributed as dist

from .tiles import max_owned_tiles, owned_tiles


class TileExchange:
    """One step: render() this rank's tiles of the N frames straight into the send chunks (one wavefront,
    packed layout), begin() the all-to-all, finish() = wait for it + one unpack kernel: frame `rank` is
    then complete in `d_frame_rgba`.  Send and receive buffers are double-buffered, so a caller may
    render step k+1 between begin() and finish() of step k: the collective (RCCL's own stream and a few
    workgroups) then runs under the next step's trace kernels instead of between steps.  exchange() is
    begin() + finish() back to back."""

    def __init__(self, ctx, width, height, tile, rank, world, device, group=None):
        self.ctx, self.w, self.h, self.tile, self.rank, self.world, self.group = ctx, width, height, tile, rank, world, group
        lists = [owned_tiles(width, height, tile, r, world) for r in range(world)]
        self.ntiles = [int(len(l)) for l in lists]
        self.chunk = max_owned_tiles(width, height, tile, world) * tile * tile * 4   # bytes, equal splits
        self.d_lists = [torch.from_numpy(np.ascontiguousarray(l)).to(device) for l in lists]
        # zeroed once: the padding of a chunk is never written
        self.send = [torch.zeros(world * self.chunk, dtype=torch.uint8, device=device) for _ in range(2)]
        self.recv = [torch.zeros(world * self.chunk, dtype=torch.uint8, device=device) for _ in range(2)]
        self.cur = 0          # the buffer pair render() / pack() fill and begin() sends
        self.pending = None   # (work handle, buffer index, host staging) of the all-to-all in flight

    def chunks(self):
        c = self.chunk
        return [self.send[self.cur][f * c:(f + 1) * c] for f in range(self.world)]

    def render(self, d_pos_list, d_nrm_list):
        """Shade this rank's tiles of every frame of the step into the send chunks (chunk f -> rank f)."""
        # un-fenced entry: the call returns after its last kernel, and begin() / finish() order the streams around the collective
        type(self.ctx).render_tiles_packed_device.__wrapped__(self.ctx, d_pos_list, d_nrm_list, self.w, self.h, self.tile, self.rank,
                                                              self.world, self.chunks())

    def pack(self, frame_index, d_tiles_rgba):
        """Alternative to render(): pack this rank's tiles of a full-size image into chunk `frame_index`."""
        c = self.chunk
        self.ctx.pack_tiles_device(d_tiles_rgba, self.w, self.h, self.tile, self.d_lists[self.rank], self.ntiles[self.rank],
                                   self.send[self.cur][frame_index * c:(frame_index + 1) * c])

    def begin(self):
        """Start the all-to-all of the chunks just rendered; at most one may be in flight."""
        if self.pending is not None:
            raise RuntimeError("TileExchange.begin(): finish() the exchange in flight first")
        b = self.cur
        torch.cuda.current_stream().synchronize()                 # nothing of torch's pending on send / recv
        if dist.get_backend(self.group) == "gloo":                # rehearsal on a box without RCCL peers: stage through the host
            host_recv = torch.zeros(self.recv[b].shape, dtype=torch.uint8)
            work = dist.all_to_all_single(host_recv, self.send[b].cpu(), group=self.group, async_op=True)
        else:
            host_recv = None
            work = dist.all_to_all_single(self.recv[b], self.send[b], group=self.group, async_op=True)
        self.pending = (work, b, host_recv)
        self.cur = b ^ 1

    def finish(self, d_frame_rgba):
        """Wait for the all-to-all begun last and queue the kernel that assembles frame `rank` in d_frame_rgba on the
        context's stream (no-op if none is in flight).  Readers on another stream call ctx.sync() first."""
        if self.pending is None:
            return
        work, b, host_recv = self.pending
        self.pending = None
        work.wait()
        if host_recv is not None:
            self.recv[b].copy_(host_recv)
        torch.cuda.current_stream().synchronize()                 # chunks landed before the context's stream reads them
        # un-fenced entry: the unpack kernel is queued on the context's stream and the host moves on; the next render
        # queues behind it, and recv[b] is not written again before a later render has drained that stream
        type(self.ctx).unpack_chunks_device.__wrapped__(self.ctx, self.recv[b], self.chunk, self.w, self.h, self.tile, self.world,
                                                        d_frame_rgba)

    def exchange(self, d_frame_rgba):
        """All-to-all of the packed chunks, then assemble frame `rank` in d_frame_rgba."""
        self.begin()
        self.finish(d_frame_rgba)
