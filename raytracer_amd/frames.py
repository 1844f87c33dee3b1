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
        # The context's own stream as a torch stream: a collective issued while it is torch's current stream is
        # ordered against the context's kernels by events (RCCL's stream waits for an event recorded on it, and
        # Work.wait() makes it wait for RCCL's stream), so no host-side synchronize brackets the exchange.
        self.ctx_stream = torch.cuda.ExternalStream(ctx.stream(), device=device)

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
        if dist.get_backend(self.group) == "gloo":                # rehearsal on a box without RCCL peers: stage through the host
            self.ctx.sync()
            host_recv = torch.zeros(self.recv[b].shape, dtype=torch.uint8)
            work = dist.all_to_all_single(host_recv, self.send[b].cpu(), group=self.group, async_op=True)
        else:
            host_recv = None
            with torch.cuda.stream(self.ctx_stream):              # RCCL's stream waits for the chunks' kernels: an event, not the host
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
        if host_recv is not None:
            work.wait()
            self.recv[b].copy_(host_recv)
            torch.cuda.current_stream().synchronize()             # chunks landed before the context's stream reads them
        else:
            with torch.cuda.stream(self.ctx_stream):              # the context's stream waits for RCCL's: an event, not the host
                work.wait()
        # un-fenced entry: the unpack kernel is queued on the context's stream and the host moves on; the next render
        # queues behind it, and recv[b] is not written again before a later render has drained that stream
        type(self.ctx).unpack_chunks_device.__wrapped__(self.ctx, self.recv[b], self.chunk, self.w, self.h, self.tile, self.world,
                                                        d_frame_rgba)

    def exchange(self, d_frame_rgba):
        """All-to-all of the packed chunks, then assemble frame `rank` in d_frame_rgba."""
        self.begin()
        self.finish(d_frame_rgba)


class TileAllGather:
    """Strong scaling: ONE frame split over the ranks.  render() shades this rank's tiles of the frame into its
    chunk, gather() all-gathers the chunks (the "all-gather of the final tiles" of SURVEY.md 8e: every rank
    receives (N-1)/N of a frame) and queues the kernel that assembles the whole frame on every rank.  Ordered
    against the context's stream by events, like TileExchange.  With a single rank the chunk is the packed
    whole frame, which keeps the collective and the stream ordering testable on one GPU."""

    def __init__(self, ctx, width, height, tile, rank, world, device, group=None):
        self.ctx, self.w, self.h, self.tile, self.rank, self.world, self.group = ctx, width, height, tile, rank, world, group
        lists = [owned_tiles(width, height, tile, r, world) for r in range(world)]
        self.ntiles = [int(len(l)) for l in lists]
        self.chunk = max_owned_tiles(width, height, tile, world) * tile * tile * 4
        self.d_list = torch.from_numpy(np.ascontiguousarray(lists[rank])).to(device)
        self.send = torch.zeros(self.chunk, dtype=torch.uint8, device=device)       # zeroed once: padding is never written
        self.recv = torch.zeros(world * self.chunk, dtype=torch.uint8, device=device)
        self.whole = torch.zeros(width * height * 4, dtype=torch.uint8, device=device) if world == 1 else None
        self.ctx_stream = torch.cuda.ExternalStream(ctx.stream(), device=device)

    def render(self, d_pos, d_nrm):
        if self.world == 1:
            type(self.ctx).render_frame_device.__wrapped__(self.ctx, d_pos, d_nrm, self.w, self.h, self.whole)
            type(self.ctx).pack_tiles_device.__wrapped__(self.ctx, self.whole, self.w, self.h, self.tile, self.d_list, self.ntiles[0], self.send)
        else:
            type(self.ctx).render_tiles_packed_device.__wrapped__(self.ctx, [d_pos], [d_nrm], self.w, self.h, self.tile, self.rank,
                                                                  self.world, [self.send])

    def gather(self, d_frame_rgba):
        """All-gather + unpack, queued behind render() on the context's stream; ctx.sync() before reading elsewhere."""
        if dist.get_backend(self.group) == "gloo":
            self.ctx.sync()
            host = torch.zeros(self.recv.shape, dtype=torch.uint8)
            dist.all_gather_into_tensor(host, self.send.cpu(), group=self.group)
            self.recv.copy_(host)
            torch.cuda.current_stream().synchronize()
        else:
            with torch.cuda.stream(self.ctx_stream):
                dist.all_gather_into_tensor(self.recv, self.send, group=self.group)   # blocking form: current stream waits for RCCL's
        type(self.ctx).unpack_chunks_device.__wrapped__(self.ctx, self.recv, self.chunk, self.w, self.h, self.tile, self.world,
                                                        d_frame_rgba)
