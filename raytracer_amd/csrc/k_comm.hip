// raytracer_amd/csrc/k_comm.hip -- the multi-GPU path behind the C ABI: one process (and one context) per
// GPU, the frame split by interleaved screen tiles with the scene replicated, the photon pass sharded by
// photon index, and RCCL over xGMI used for exactly two exchange steps: the all-gather of the
// finished tiles and the all-gather of the photon shards.  The reference has no counterpart (one thread, one GL
// context).  RCCL is bound at run time (dlopen) so that single-GPU users of librt_hip.so need no librccl, and so
// that a process which already carries an RCCL (PyTorch bundles one) keeps using that one.
//
// Both collectives are enqueued on the context's own stream, so they are ordered against its kernels by the
// stream itself: no host synchronisation brackets them.
#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>

#include "rt_ctx.h"

namespace {

struct RcclApi {
	void* lib = nullptr;
	ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	const char* (*GetErrorString)(ncclResult_t) = nullptr;
	ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*GroupStart)() = nullptr;
	ncclResult_t (*GroupEnd)() = nullptr;
};
RcclApi g_rccl;

std::mutex g_rccl_mutex;   // contexts of different threads may ask for RCCL at the same time

int load_rccl()
{
	std::lock_guard<std::mutex> lock(g_rccl_mutex);
	if (g_rccl.lib) return RT_HIP_OK;
	const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
	void* lib = nullptr;
	for (const char* nm : names) {
		lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
		if (lib) break;
	}
	if (!lib) { rt_set_error("librccl not found (%s): the multi-GPU entries need RCCL", dlerror()); return RT_HIP_ERR_STATE; }
	RcclApi a;
	a.lib = lib;
#define RT_SYM(field, name)                                                                              \
	a.field = (decltype(a.field))dlsym(lib, name);                                                       \
	if (!a.field) { rt_set_error("librccl lacks %s", name); dlclose(lib); return RT_HIP_ERR_STATE; }
	RT_SYM(GetUniqueId, "ncclGetUniqueId")
	RT_SYM(CommInitRank, "ncclCommInitRank")
	RT_SYM(CommDestroy, "ncclCommDestroy")
	RT_SYM(GetErrorString, "ncclGetErrorString")
	RT_SYM(AllGather, "ncclAllGather")
	RT_SYM(Send, "ncclSend")
	RT_SYM(Recv, "ncclRecv")
	RT_SYM(GroupStart, "ncclGroupStart")
	RT_SYM(GroupEnd, "ncclGroupEnd")
#undef RT_SYM
	g_rccl = a;
	return RT_HIP_OK;
}

#define RT_NCCL_CHECK(expr)                                                                                     \
	do {                                                                                                        \
		ncclResult_t r__ = (expr);                                                                              \
		if (r__ != ncclSuccess) {                                                                               \
			rt_set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(r__), __FILE__, __LINE__);       \
			return RT_HIP_ERR_HIP;                                                                              \
		}                                                                                                       \
	} while (0)

// tiles.py: photon_shard -- [first, first + count) of `total` indices owned by `rank`
void photon_shard(int64_t total, int rank, int nranks, int64_t* first, int64_t* count)
{
	const int64_t base = total / nranks, rem = total % nranks;
	*first = rank * base + std::min<int64_t>(rank, rem);
	*count = base + (rank < rem ? 1 : 0);
}

int max_owned_tiles(int w, int h, int tile, int nranks)
{
	const int tx = (w + tile - 1) / tile, ty = (h + tile - 1) / tile;
	std::vector<int> cnt(nranks, 0);
	for (int y = 0; y < ty; ++y)
		for (int x = 0; x < tx; ++x) ++cnt[(x + y) % nranks];
	return *std::max_element(cnt.begin(), cnt.end());
}

}  // namespace

extern "C" {

int rt_hip_comm_unique_id(void* id128)
{
	if (!id128) { rt_set_error("null id buffer"); return RT_HIP_ERR_ARG; }
	RT_CHECK(load_rccl());
	static_assert(sizeof(ncclUniqueId) == RT_HIP_COMM_ID_BYTES, "RCCL unique id size");
	ncclUniqueId id;
	RT_NCCL_CHECK(g_rccl.GetUniqueId(&id));
	memcpy(id128, &id, sizeof(id));
	return RT_HIP_OK;
}

int rt_hip_comm_init(rt_hip_ctx* ctx, const void* id128, int rank, int nranks)
{
	if (!ctx || !id128) { rt_set_error("null argument"); return RT_HIP_ERR_ARG; }
	if (nranks < 1 || nranks > RT_MAX_BATCH || rank < 0 || rank >= nranks) { rt_set_error("bad rank %d of %d (at most %d ranks)", rank, nranks, RT_MAX_BATCH); return RT_HIP_ERR_ARG; }
	if (ctx->comm) { rt_set_error("the context already has a communicator"); return RT_HIP_ERR_STATE; }
	RT_HIP_CHECK(hipSetDevice(ctx->cfg.device));
	RT_CHECK(load_rccl());
	ncclUniqueId id;
	memcpy(&id, id128, sizeof(id));
	ncclComm_t comm = nullptr;
	RT_NCCL_CHECK(g_rccl.CommInitRank(&comm, nranks, id, rank));
	ctx->comm = comm;
	ctx->comm_rank = rank;
	ctx->comm_nranks = nranks;
	return RT_HIP_OK;
}

int rt_hip_comm_destroy(rt_hip_ctx* ctx)
{
	if (!ctx || !ctx->comm) return RT_HIP_OK;
	(void)hipSetDevice(ctx->cfg.device);
	(void)hipStreamSynchronize(ctx->stream);
	ncclResult_t r = g_rccl.CommDestroy((ncclComm_t)ctx->comm);
	ctx->comm = nullptr;
	ctx->comm_rank = 0;
	ctx->comm_nranks = 1;
	if (r != ncclSuccess) { rt_set_error("ncclCommDestroy failed: %s", g_rccl.GetErrorString(r)); return RT_HIP_ERR_HIP; }
	return RT_HIP_OK;
}

// One frame split over the ranks of the context's communicator (strong scaling): this rank shades its interleaved
// tiles straight into its packed chunk, one all-gather hands every rank all chunks, one kernel scatters them into
// the image.  A one-rank communicator runs exactly the same steps (packed tiles -> all-gather of one chunk -> unpack).
int rt_hip_render_frame_multi_device(rt_hip_ctx* ctx, const float* d_gbuf_pos_rgb, const float* d_gbuf_nrm_rgba, int width,
                                     int height, int tile, uint8_t* d_rgba_out)
{
	if (!ctx) { rt_set_error("null context"); return RT_HIP_ERR_ARG; }
	RT_HIP_CHECK(hipSetDevice(ctx->cfg.device));
	if (ctx->scene_dirty) { rt_set_error("scene changed: call rt_hip_update() before tracing"); return RT_HIP_ERR_STATE; }
	if (!ctx->comm) { rt_set_error("no communicator: call rt_hip_comm_init first"); return RT_HIP_ERR_STATE; }
	if (width <= 0 || height <= 0 || !d_gbuf_pos_rgb || !d_gbuf_nrm_rgba || !d_rgba_out || tile <= 0 || tile % 8 != 0 || tile > 1024) {
		rt_set_error("bad arguments (the tile size must be a positive multiple of 8)");
		return RT_HIP_ERR_ARG;
	}
	const int rank = ctx->comm_rank, nranks = ctx->comm_nranks;
	hipStream_t st = ctx->stream;
	const size_t chunk = (size_t)max_owned_tiles(width, height, tile, nranks) * tile * tile * 4;
	const int key[4] = {width, height, tile, nranks};
	if (memcmp(key, ctx->mg_key, sizeof(key)) != 0 || !ctx->d_mg_send.p) {
		RT_CHECK(ctx->d_mg_send.reserve(chunk));
		RT_CHECK(ctx->d_mg_recv.reserve(chunk * nranks));
		RT_HIP_CHECK(hipMemsetAsync(ctx->d_mg_send.p, 0, chunk, st));  // the padding of edge tiles is never written
		memcpy(ctx->mg_key, key, sizeof(key));
	}
	uint8_t* send = ctx->d_mg_send.as<uint8_t>();
	int64_t rays = 0;
	{	// every communicator size takes this one path (a single rank owns all the tiles).  The all-gather and the unpack are
		// queued behind the frame's kernels before the host waits for the frame's counters: one host wait per frame.
		const float* pos[1] = {d_gbuf_pos_rgb};
		const float* nrm[1] = {d_gbuf_nrm_rgba};
		uint8_t* out[1] = {send};
		ctx->after_frame_queued = [&]() -> int {
			RT_NCCL_CHECK(g_rccl.AllGather(send, ctx->d_mg_recv.p, chunk, ncclUint8, (ncclComm_t)ctx->comm, st));
			return rt_unpack_chunks(ctx, ctx->d_mg_recv.as<uint8_t>(), (int64_t)chunk, width, height, tile, nranks, d_rgba_out);
		};
		const int status = rt_render_frames(ctx, pos, nrm, 1, width, height, tile, rank, nranks, out, 1);
		ctx->after_frame_queued = nullptr;
		RT_CHECK(status);
		rays = ctx->last_frame_rays;
	}
	RT_HIP_CHECK(hipStreamSynchronize(st));
	ctx->last_frame_rays = rays;
	return RT_HIP_OK;
}

// The same frame split, queued without waiting (the frame pipeline of rt_hip_frame_enqueue_device, rt_hip_frame_finish
// completes the oldest entry): the rank's tiles are shaded on the lane the entry is dealt to (tuning.frame_streams), and the
// all-gather and the unpack follow on a stream of their own, in entry order on every rank -- so the next entry's kernels
// start at once and fill the machine under the tail of this one's, which is what a split frame needs: its launches are
// short and each as long as its slowest wave.  For the same reason an entry may hold SEVERAL frames (nframes): one pass of
// launches shades this rank's tiles of all of them, one all-gather carries all their chunks ([rank][frame][chunk] on
// arrival), one unpack per frame -- with the frame split N ways, N / 2 frames per entry keep the launches as large as half
// a frame's.  Exactly one all-gather per call, queued during the call, on every path.
int rt_hip_frames_multi_enqueue_device(rt_hip_ctx* ctx, const float* const* d_gbuf_pos_rgb, const float* const* d_gbuf_nrm_rgba,
                                       int nframes, int width, int height, int tile, uint8_t* const* d_rgba_out)
{
	if (!ctx) { rt_set_error("null context"); return RT_HIP_ERR_ARG; }
	RT_HIP_CHECK(hipSetDevice(ctx->cfg.device));
	if (ctx->scene_dirty) { rt_set_error("scene changed: call rt_hip_update() before tracing"); return RT_HIP_ERR_STATE; }
	if (!ctx->comm) { rt_set_error("no communicator: call rt_hip_comm_init first"); return RT_HIP_ERR_STATE; }
	if (width <= 0 || height <= 0 || !d_gbuf_pos_rgb || !d_gbuf_nrm_rgba || !d_rgba_out || tile <= 0 || tile % 8 != 0 || tile > 1024 ||
	    nframes < 1 || nframes > RT_MAX_BATCH) {
		rt_set_error("bad arguments (1..%d frames; the tile size must be a positive multiple of 8)", RT_MAX_BATCH);
		return RT_HIP_ERR_ARG;
	}
	for (int f = 0; f < nframes; ++f)
		if (!d_gbuf_pos_rgb[f] || !d_gbuf_nrm_rgba[f] || !d_rgba_out[f]) { rt_set_error("frame %d: null buffer", f); return RT_HIP_ERR_ARG; }
	int slot = 0;
	RT_CHECK(rt_pipeline_slot(ctx, &slot));
	const int rank = ctx->comm_rank, nranks = ctx->comm_nranks;
	const int64_t most_tiles = max_owned_tiles(width, height, tile, nranks);
	const size_t chunk = (size_t)most_tiles * tile * tile * 4;
	// Only entries the sync-free driver takes with no level able to overflow its bound (such a frame would have to be redone
	// after its tiles had been sent).  Decided from what the LARGEST share of any rank needs, so that every rank decides alike:
	// a rank that returned here while the others queued their all-gather would leave them waiting.
	const int bounces = ctx->cfg.bounces;
	if (ctx->cfg.use_photon_map || ctx->tuning.frame_sync || (int)ctx->lights.size() > 30 || bounces < 1 || bounces > 8 ||
	    ((most_tiles * tile * tile * nframes) << (bounces - 1)) > ((int64_t)1 << 23)) {
		rt_set_error("%d frame(s) of %d x %d texels split %d ways with this configuration cannot be enqueued: fewer frames per call, or rt_hip_render_frame_multi_device",
		             nframes, width, height, nranks);
		return RT_HIP_ERR_LIMIT;
	}
	if (!ctx->comm_stream) RT_HIP_CHECK(hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
	PendingFrame* pf = &ctx->pending[slot];
	if (!pf->ev_done) RT_HIP_CHECK(hipEventCreateWithFlags(&pf->ev_done, hipEventDisableTiming));
	if (!pf->ev_down) RT_HIP_CHECK(hipEventCreateWithFlags(&pf->ev_down, hipEventDisableTiming));
	// the slot's chunks are free: the entry that used them last has been finished
	DevBuf& send = ctx->d_mg_send_slot[slot];
	DevBuf& recv = ctx->d_mg_recv_slot[slot];
	const size_t mine = chunk * nframes;   // this rank's chunks of the entry's frames, one after the other
	if (send.cap < mine) {
		RT_CHECK(send.reserve(mine));
		RT_HIP_CHECK(hipMemsetAsync(send.p, 0, send.cap, ctx->stream));  // the padding of edge tiles is never written
		RT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
	}
	RT_CHECK(recv.reserve(mine * nranks));
	uint8_t* chunks[RT_MAX_BATCH];
	for (int f = 0; f < nframes; ++f) chunks[f] = send.as<uint8_t>() + (size_t)f * chunk;
	RT_CHECK(rt_frames_enqueue(ctx, d_gbuf_pos_rgb, d_gbuf_nrm_rgba, nframes, width, height, chunks, tile, rank, nranks, 1));
	RT_HIP_CHECK(hipEventRecord(pf->ev_done, pf->stream));
	RT_HIP_CHECK(hipStreamWaitEvent(ctx->comm_stream, pf->ev_done, 0));
	RT_NCCL_CHECK(g_rccl.AllGather(send.p, recv.p, mine, ncclUint8, (ncclComm_t)ctx->comm, ctx->comm_stream));
	for (int f = 0; f < nframes; ++f)
		RT_CHECK(rt_unpack_chunks(ctx, recv.as<uint8_t>() + (size_t)f * chunk, (int64_t)chunk, width, height, tile, nranks, d_rgba_out[f],
		                          ctx->comm_stream, (int64_t)mine));
	RT_HIP_CHECK(hipEventRecord(pf->ev_down, ctx->comm_stream));
	pf->comm = true;
	return RT_HIP_OK;
}

int rt_hip_frame_multi_enqueue_device(rt_hip_ctx* ctx, const float* d_gbuf_pos_rgb, const float* d_gbuf_nrm_rgba, int width,
                                      int height, int tile, uint8_t* d_rgba_out)
{
	if (!d_gbuf_pos_rgb || !d_gbuf_nrm_rgba || !d_rgba_out) { rt_set_error("bad arguments"); return RT_HIP_ERR_ARG; }
	return rt_hip_frames_multi_enqueue_device(ctx, &d_gbuf_pos_rgb, &d_gbuf_nrm_rgba, 1, width, height, tile, &d_rgba_out);
}

// PhotonMap::generate sharded by photon index: this rank emits its contiguous range of the global index
// sequence, the stored counts are all-gathered, every rank sends its shard straight to every other rank
// (grouped point-to-point: exact sizes, each xGMI link carries one shard, no ring), and every rank installs the
// concatenation in rank order -- which is the single-GPU photon array, bit for bit (counter RNG keyed by index).
int rt_hip_photon_generate_multi(rt_hip_ctx* ctx, int photons_per_light, int64_t* out_stored)
{
	if (!ctx) { rt_set_error("null context"); return RT_HIP_ERR_ARG; }
	RT_HIP_CHECK(hipSetDevice(ctx->cfg.device));
	if (ctx->scene_dirty) { rt_set_error("scene changed: call rt_hip_update() before tracing"); return RT_HIP_ERR_STATE; }
	if (!ctx->comm) { rt_set_error("no communicator: call rt_hip_comm_init first"); return RT_HIP_ERR_STATE; }
	if (photons_per_light <= 0 || !out_stored) { rt_set_error("bad arguments"); return RT_HIP_ERR_ARG; }
	*out_stored = 0;
	const int rank = ctx->comm_rank, nranks = ctx->comm_nranks;
	hipStream_t st = ctx->stream;
	const int64_t total = (int64_t)photons_per_light * (int64_t)ctx->lights.size();
	int64_t first = 0, count = 0;
	photon_shard(total, rank, nranks, &first, &count);
	RT_CHECK(ctx->d_photon_emit.reserve(std::max<size_t>((size_t)count, 1) * 36));
	int64_t mine = 0;
	if (count > 0) RT_CHECK(rt_photon_emit(ctx, photons_per_light, first, count, ctx->d_photon_emit.as<float>(), count, &mine));
	// counts
	RT_CHECK(ctx->d_mg_counts.reserve(((size_t)nranks + 1) * sizeof(int64_t)));
	int64_t* d_counts = ctx->d_mg_counts.as<int64_t>();
	int64_t* h = (int64_t*)ctx->h_small + 64;  // pinned
	h[0] = mine;
	RT_HIP_CHECK(hipMemcpyAsync(d_counts + nranks, h, sizeof(int64_t), hipMemcpyHostToDevice, st));
	RT_NCCL_CHECK(g_rccl.AllGather(d_counts + nranks, d_counts, 1, ncclInt64, (ncclComm_t)ctx->comm, st));
	RT_HIP_CHECK(hipMemcpyAsync(h + 1, d_counts, (size_t)nranks * sizeof(int64_t), hipMemcpyDeviceToHost, st));
	RT_HIP_CHECK(hipStreamSynchronize(st));
	std::vector<int64_t> cnt(h + 1, h + 1 + nranks), off(nranks + 1, 0);
	for (int r = 0; r < nranks; ++r) off[r + 1] = off[r] + cnt[r];
	const int64_t all = off[nranks];
	if (cnt[rank] != mine) { rt_set_error("photon count exchange returned %lld for this rank, expected %lld", (long long)cnt[rank], (long long)mine); return RT_HIP_ERR_STATE; }
	// payload: rank r's photons land at [off[r], off[r + 1]) of the full array on every rank
	RT_CHECK(ctx->d_mg_photons.reserve(std::max<size_t>((size_t)all, 1) * 36));
	float* full = ctx->d_mg_photons.as<float>();
	if (mine > 0) RT_HIP_CHECK(hipMemcpyAsync(full + 9 * off[rank], ctx->d_photon_emit.p, (size_t)mine * 36, hipMemcpyDeviceToDevice, st));
	RT_NCCL_CHECK(g_rccl.GroupStart());
	ncclResult_t posted = ncclSuccess;   // a failed call must not leave the group open
	for (int r = 0; r < nranks && posted == ncclSuccess; ++r) {
		if (r == rank) continue;
		if (mine > 0) posted = g_rccl.Send(ctx->d_photon_emit.p, (size_t)mine * 9, ncclFloat32, r, (ncclComm_t)ctx->comm, st);
		if (posted == ncclSuccess && cnt[r] > 0) posted = g_rccl.Recv(full + 9 * off[r], (size_t)cnt[r] * 9, ncclFloat32, r, (ncclComm_t)ctx->comm, st);
	}
	const ncclResult_t ended = g_rccl.GroupEnd();
	RT_NCCL_CHECK(posted);
	RT_NCCL_CHECK(ended);
	RT_CHECK(rt_photon_build_grid(ctx, full, all));
	*out_stored = all;
	return RT_HIP_OK;
}

}  // extern "C"
