// raytracer_amd/csrc/rt_api.hip -- the C ABI declared in include/rt_hip.h: context life cycle,
// scene ingestion, and the host orchestration of the kernels in k_*.hip.  No CPU fallback lives
// here: every entry point either runs HIP kernels or fails with an error code.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <cmath>
#include <limits>

#include "rt_ctx.h"

int rt_fill_sphere_records(rt_hip_ctx* ctx, const int* d_index, const float* d_pos, const float* d_nrm, int n,
                           rt_ray* d_rays, uint8_t* d_flag, uint32_t* d_material);

static thread_local char g_err[512] = "";

void rt_set_error(const char* fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
}

int DevBuf::reserve(size_t bytes)
{
	if (bytes <= cap && p) return RT_HIP_OK;
	size_t want = bytes < 256 ? 256 : bytes;
	if (p) want = std::max(want, cap + cap / 2);  // geometric growth
	void* np = nullptr;
	hipError_t e = hipMalloc(&np, want);
	if (e != hipSuccess) {
		rt_set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
		return RT_HIP_ERR_HIP;
	}
	if (p) {
		// contents are preserved (level arrays of a frame grow while earlier levels are live)
		e = hipDeviceSynchronize();
		if (e == hipSuccess) e = hipMemcpy(np, p, cap, hipMemcpyDeviceToDevice);
		if (e != hipSuccess) {
			rt_set_error("DevBuf grow copy failed: %s", hipGetErrorString(e));
			(void)hipFree(np);
			return RT_HIP_ERR_HIP;
		}
		(void)hipFree(p);
	}
	p = np;
	cap = want;
	return RT_HIP_OK;
}

void DevBuf::release()
{
	if (p) (void)hipFree(p);
	p = nullptr;
	cap = 0;
}

int rt_ensure_pinned(void** p, size_t* cap, size_t bytes)
{
	if (*p && *cap >= bytes) return RT_HIP_OK;
	if (*p) (void)hipHostFree(*p);
	*p = nullptr;
	*cap = 0;
	size_t want = bytes < 4096 ? 4096 : bytes + bytes / 4;
	RT_HIP_CHECK(hipHostMalloc(p, want, hipHostMallocDefault));
	*cap = want;
	return RT_HIP_OK;
}

static int env_int(const char* name, int dflt)
{
	const char* e = getenv(name);
	return e ? atoi(e) : dflt;
}

// Defaults of rt_hip_tuning (include/rt_hip.h) with the RT_HIP_* environment overrides, read once per context
// at rt_hip_create.  xcd_chunk: with one-wave workgroups the 1080p frame measured 1.163 ms at 0 (hardware order),
// 1.153 at 32, 1.139 at 64, 1.142 at 128, 1.158 at 256, 1.171 at 512: each XCD's L2 then holds the tree under 64
// consecutive waves' texels instead of every eighth wave's.
void rt_default_tuning(rt_hip_tuning* t)
{
	memset(t, 0, sizeof(*t));
	t->lds_stack_depth = env_int("RT_HIP_LDS_DEPTH", 16);
	t->xcd_chunk = env_int("RT_HIP_XCD_CHUNK", 64);
	t->trace_oversub = env_int("RT_HIP_OVERSUB", 4);
	t->auto_accel = env_int("RT_HIP_AUTO_PLOC", 1) ? RT_HIP_ACCEL_PLOC : RT_HIP_ACCEL_LBVH;
	t->ploc_radius = env_int("RT_HIP_PLOC_RADIUS", 8);
	t->frame_sync = env_int("RT_HIP_FRAME_SYNC", 0);
	t->gather_heavy = 32;
	if (const char* e = getenv("RT_HIP_GATHER_HEAVY")) t->gather_heavy = (uint32_t)strtoul(e, nullptr, 10);
	t->extra_lds_bytes = env_int("RT_HIP_EXTRA_LDS", 0);
	t->gather_fine = env_int("RT_HIP_GATHER_FINE", 1);
	t->gather_xcd_chunk = env_int("RT_HIP_GATHER_XCD_CHUNK", 1024);
	t->gather_fine_cells = env_int("RT_HIP_GATHER_FINE_CELLS", 8);
	t->gather_medium = 4096;
	if (const char* e = getenv("RT_HIP_GATHER_MEDIUM")) t->gather_medium = (uint32_t)strtoul(e, nullptr, 10);
	t->shadow_prefix = env_int("RT_HIP_SHADOW_PREFIX", 12);
	t->shadow_prefix_steps = env_int("RT_HIP_SHADOW_PREFIX_STEPS", 24);
	t->shadow_prefix_min_rays = env_int("RT_HIP_SHADOW_PREFIX_MIN_RAYS", -1);
	t->frame_streams = env_int("RT_HIP_FRAME_STREAMS", 1);
	t->gather_pair = env_int("RT_HIP_GATHER_PAIR", 100);
	t->gather_sort = env_int("RT_HIP_GATHER_SORT", 1);
	t->max_bvh_depth = 95;
	t->refit_max_growth = 2.0f;
	if (const char* e = getenv("RT_HIP_REFIT_GROWTH")) t->refit_max_growth = (float)atof(e);
}

void rt_clamp_tuning(rt_hip_ctx* ctx)
{
	rt_hip_tuning& t = ctx->tuning;
	t.lds_stack_depth = std::min(32, std::max(1, t.lds_stack_depth));
	t.trace_oversub = std::min(16, std::max(1, t.trace_oversub));
	ctx->max_trace_threads = t.trace_oversub * ctx->resident_threads;
	// the XCD-aware renumbering needs grids that are multiples of 8 * chunk blocks: keep one such unit of the largest
	// trace workgroup (256 threads) inside the grid cap, which is also the number of columns of the overflow stack
	t.xcd_chunk = std::min(std::max(0, t.xcd_chunk), std::max(1, ctx->max_trace_threads / (8 * 256)));
	if (t.auto_accel != RT_HIP_ACCEL_LBVH) t.auto_accel = RT_HIP_ACCEL_PLOC;
	t.ploc_radius = std::min(64, std::max(1, t.ploc_radius));
	t.frame_sync = t.frame_sync ? 1 : 0;
	t.gather_fine = t.gather_fine ? 1 : 0;
	t.gather_xcd_chunk = std::min(8192, std::max(0, t.gather_xcd_chunk));
	t.gather_fine_cells = std::min(32, std::max(2, t.gather_fine_cells));
	t.extra_lds_bytes = std::min(64 * 1024, std::max(0, t.extra_lds_bytes));
	t.max_bvh_depth = std::min(95, std::max(1, t.max_bvh_depth));
	t.shadow_prefix = std::min(30, std::max(0, t.shadow_prefix));
	t.shadow_prefix_steps = std::min(1024, std::max(1, t.shadow_prefix_steps));
	t.frame_streams = std::min(RT_MAX_PENDING, std::max(1, t.frame_streams));
	t.gather_pair = std::min(100000, std::max(0, t.gather_pair));
	t.gather_sort = std::min(2, std::max(0, t.gather_sort));
	if (!(t.refit_max_growth >= 0.0f)) t.refit_max_growth = 0.0f;
}

int rt_trace_blocks(const rt_hip_ctx* ctx, int64_t work_items, int block)
{
	int64_t need = (work_items + block - 1) / block;
	// a multiple of 8 * xcd_chunk, so that the XCD-aware renumbering of the blocks is a bijection
	const int64_t unit = 8 * std::max(1, ctx->tuning.xcd_chunk);
	// never more threads than the overflow stack has columns (rt_clamp_tuning keeps one unit of 256-thread blocks below it)
	int64_t cap = (int64_t)ctx->max_trace_threads / block / unit * unit;
	if (cap < unit) cap = std::max<int64_t>(1, (int64_t)ctx->max_trace_threads / block);
	need = std::max<int64_t>(need, 1);
	// Above the cap every thread loops; size the grid so that all threads make the SAME number of rounds
	// (measured on the 1080p frame: a grid that leaves a third of the threads one ray more than the rest is
	// 5 % slower than either neighbour that divides evenly -- RT_HIP_OVERSUB 6 against 4 and 8).
	if (need > cap) { const int64_t rounds = (need + cap - 1) / cap; need = (need + rounds - 1) / rounds; }
	return (int)std::min((need + unit - 1) / unit * unit, cap);
}

size_t rt_trace_lds(const SceneView& sv, int block)
{
	return sv.use_bvh ? (size_t)sv.lds_depth * block * sizeof(int) + (size_t)sv.extra_lds : 0;
}

SceneView rt_scene_view(const rt_hip_ctx* ctx)
{
	SceneView sv;
	sv.nodes = ctx->d_nodes.as<BvhNode>();
	sv.qnodes = ctx->d_qnodes.as<QNode>();
	sv.qframe = ctx->qframe;
	sv.tris = ctx->d_tri_isect.as<TriIsect>();
	sv.tri_pos = ctx->d_tri_pos.as<TriPos>();
	sv.tri_attr = ctx->d_tri_attr.as<TriAttr>();
	sv.spheres = ctx->d_spheres.as<SphereRec>();
	sv.materials = ctx->d_materials.as<MaterialRec>();
	sv.lights = ctx->d_lights.as<LightRec>();
	sv.tex = ctx->any_texture ? ctx->d_tex.as<TexRec>() : nullptr;
	sv.texels = ctx->d_texels.as<uint32_t>();
	sv.num_tris = ctx->cfg.geometry == RT_HIP_SPHERES ? 0 : ctx->num_tris;
	sv.num_spheres = ctx->cfg.geometry == RT_HIP_SPHERES ? (int)ctx->spheres.size() : 0;
	sv.num_materials = (int)ctx->materials.size() + sv.num_spheres;
	sv.num_lights = (int)ctx->lights.size();
	sv.use_bvh = ctx->use_bvh;
	sv.stack_depth = ctx->bvh_depth + 2;  // + the EXIT sentinel at the bottom (bvh_trace)
	sv.lds_depth = std::min(sv.stack_depth, ctx->tuning.lds_stack_depth);
	sv.spill = ctx->d_spill.as<int>();
	sv.spill_stride = ctx->max_trace_threads;
	sv.xcd_chunk = ctx->tuning.xcd_chunk;
	sv.extra_lds = ctx->tuning.extra_lds_bytes;
	sv.tp.eps_det = ctx->cfg.eps_det;
	sv.tp.eps_t = ctx->cfg.eps_t;
	sv.tp.t_max = ctx->cfg.t_max;
	return sv;
}

// ---- kernel timing (profile mode) -------------------------------------------------------------
int rt_begin_trace_timing(rt_hip_ctx* ctx, hipStream_t st)
{
	if (!st) st = ctx->stream;
	if (!ctx->cfg.profile) return RT_HIP_OK;
	if (ctx->trace_events_used == ctx->trace_events.size()) {
		hipEvent_t a, b;
		RT_HIP_CHECK(hipEventCreate(&a));
		RT_HIP_CHECK(hipEventCreate(&b));
		ctx->trace_events.push_back(std::make_pair(a, b));
		ctx->trace_event_class.push_back(0);
	}
	RT_HIP_CHECK(hipEventRecord(ctx->trace_events[ctx->trace_events_used].first, st));
	return RT_HIP_OK;
}

int rt_end_trace_timing(rt_hip_ctx* ctx, int cls, hipStream_t st)
{
	if (!st) st = ctx->stream;
	if (cls == 2) ctx->stats.gather_launches += 1; else if (cls) ctx->stats.any_launches += 1; else ctx->stats.closest_launches += 1;
	if (!ctx->cfg.profile) return RT_HIP_OK;
	RT_HIP_CHECK(hipEventRecord(ctx->trace_events[ctx->trace_events_used].second, st));
	ctx->trace_event_class[ctx->trace_events_used] = cls;
	++ctx->trace_events_used;
	return RT_HIP_OK;
}

// The event pair for ONE launch timed through its own dispatch packet (hipExtLaunchKernelGGL takes the
// kernel's start and end time stamps from the packet's completion signal): unlike hipEventRecord before and
// after, nothing is queued between the kernels (each recorded event cost ~6 us of idle GPU in the frame's
// timeline).  Both null when profiling is off.
int rt_next_trace_events(rt_hip_ctx* ctx, int cls, hipEvent_t* a, hipEvent_t* b)
{
	*a = *b = nullptr;
	// cls 3: time that belongs to the any-hit launches without being one (the shadow rays' shared descent, k_shadow_prefix)
	if (cls == 3) {} else if (cls) ctx->stats.any_launches += 1; else ctx->stats.closest_launches += 1;
	if (!ctx->cfg.profile) return RT_HIP_OK;
	if (ctx->trace_events_used == ctx->trace_events.size()) {
		hipEvent_t x, y;
		RT_HIP_CHECK(hipEventCreate(&x));
		RT_HIP_CHECK(hipEventCreate(&y));
		ctx->trace_events.push_back(std::make_pair(x, y));
		ctx->trace_event_class.push_back(0);
	}
	*a = ctx->trace_events[ctx->trace_events_used].first;
	*b = ctx->trace_events[ctx->trace_events_used].second;
	ctx->trace_event_class[ctx->trace_events_used] = cls;
	++ctx->trace_events_used;
	return RT_HIP_OK;
}

unsigned long long* rt_census_ptr(rt_hip_ctx* ctx, int cls)
{
	if (!ctx->cfg.census) return nullptr;
	if (!ctx->d_census.p) {
		if (ctx->d_census.reserve(128) != RT_HIP_OK) return nullptr;
		if (hipMemset(ctx->d_census.p, 0, 128) != hipSuccess) return nullptr;  // synchronous: done before any kernel adds to it
	}
	return ctx->d_census.as<unsigned long long>() + (cls == 2 ? 6 : cls ? 3 : 0);
}

// Fold the pending event pairs into stats (requires the stream to be idle or syncs on the events).
int rt_collect_timing(rt_hip_ctx* ctx, size_t upto)
{
	if (!ctx->cfg.profile) ctx->trace_events_used = ctx->trace_events_collected = 0;
	const size_t end = std::min(upto, ctx->trace_events_used);
	for (size_t i = ctx->trace_events_collected; i < end; ++i) {
		RT_HIP_CHECK(hipEventSynchronize(ctx->trace_events[i].second));
		float ms = 0;
		RT_HIP_CHECK(hipEventElapsedTime(&ms, ctx->trace_events[i].first, ctx->trace_events[i].second));
		if (ctx->trace_event_class[i] == 2) ctx->stats.gather_ms += ms;
		else if (ctx->trace_event_class[i]) ctx->stats.any_ms += ms;
		else ctx->stats.closest_ms += ms;
	}
	ctx->trace_events_collected = std::max(ctx->trace_events_collected, end);
	if (ctx->trace_events_collected >= ctx->trace_events_used) ctx->trace_events_used = ctx->trace_events_collected = 0;
	else if (ctx->trace_events_collected > 0) {
		// The frame pipeline keeps one frame's launches recorded ahead, so the pool never drains completely: move the
		// live tail [collected, used) to the front and reuse the folded pairs, instead of creating new events for every
		// frame (the pool is then as large as two frames' launches, for any number of frames).
		const size_t shift = ctx->trace_events_collected;
		std::rotate(ctx->trace_events.begin(), ctx->trace_events.begin() + shift, ctx->trace_events.begin() + ctx->trace_events_used);
		std::rotate(ctx->trace_event_class.begin(), ctx->trace_event_class.begin() + shift, ctx->trace_event_class.begin() + ctx->trace_events_used);
		ctx->trace_events_used -= shift;
		ctx->trace_events_collected = 0;
		for (PendingFrame& pf : ctx->pending) pf.trace_mark = pf.trace_mark > shift ? pf.trace_mark - shift : 0;
	}
	if (ctx->cfg.census && ctx->d_census.p) {
		unsigned long long c[7];
		RT_HIP_CHECK(hipMemcpy(c, ctx->d_census.p, sizeof(c), hipMemcpyDeviceToHost));
		ctx->stats.census_nodes_closest = c[0]; ctx->stats.census_tris_closest = c[1]; ctx->stats.census_rays_closest = c[2];
		ctx->stats.census_nodes_any = c[3]; ctx->stats.census_tris_any = c[4]; ctx->stats.census_rays_any = c[5];
		ctx->stats.census_gather_visits = c[6];
	}
	return RT_HIP_OK;
}

// The float threshold F such that (x < F) == ((double)x < d) for every float x.
static float float_threshold_of_double(double d)
{
	float f = (float)d;
	if ((double)f < d) f = std::nextafter(f, std::numeric_limits<float>::infinity());
	return f;
}

// census sums are fetched even when profiling is off
static int check_ctx(rt_hip_ctx* ctx)
{
	if (!ctx) { rt_set_error("null context"); return RT_HIP_ERR_ARG; }
	RT_HIP_CHECK(hipSetDevice(ctx->cfg.device));
	return RT_HIP_OK;
}

static int check_ready(rt_hip_ctx* ctx)
{
	RT_CHECK(check_ctx(ctx));
	if (ctx->scene_dirty) { rt_set_error("scene changed: call rt_hip_update() before tracing"); return RT_HIP_ERR_STATE; }
	return RT_HIP_OK;
}

extern "C" {

const char* rt_hip_last_error(void) { return g_err; }

int rt_hip_default_config(rt_hip_config* cfg)
{
	if (!cfg) { rt_set_error("null config"); return RT_HIP_ERR_ARG; }
	memset(cfg, 0, sizeof(*cfg));
	cfg->device = 0;
	cfg->geometry = RT_HIP_TRIANGLES;
	cfg->accel = RT_HIP_ACCEL_AUTO;
	cfg->use_model_mat = 1;
	cfg->eps_det = std::numeric_limits<float>::epsilon();
	cfg->eps_t = std::numeric_limits<float>::epsilon();
	cfg->t_max = std::numeric_limits<float>::infinity();
	cfg->bounces = 2;
	cfg->do_shadows = 0;
	cfg->use_photon_map = 0;
	cfg->photon_count = 10000;
	cfg->photon_estimate = 100;
	cfg->photon_radius = 1.0f;
	cfg->photon_max_dist = 12.0f;
	cfg->photon_absorb = 0.6f;
	cfg->seed = 42;
	cfg->profile = 0;
	cfg->shadow_t_max = 0.0f;
	return RT_HIP_OK;
}

int rt_hip_array_config(rt_hip_config* cfg)
{
	RT_CHECK(rt_hip_default_config(cfg));
	cfg->use_model_mat = 0;
	cfg->eps_det = float_threshold_of_double(0.00001);
	cfg->eps_t = float_threshold_of_double(0.00001);
	cfg->t_max = 1000000.0f;
	return RT_HIP_OK;
}

int rt_hip_create(const rt_hip_config* cfg, rt_hip_ctx** out)
{
	if (!cfg || !out) { rt_set_error("null argument"); return RT_HIP_ERR_ARG; }
	*out = nullptr;
	int ndev = 0;
	RT_HIP_CHECK(hipGetDeviceCount(&ndev));
	if (ndev <= 0) { rt_set_error("no HIP device visible: librt_hip has no CPU fallback"); return RT_HIP_ERR_HIP; }
	if (cfg->device < 0 || cfg->device >= ndev) { rt_set_error("device %d out of range (%d devices)", cfg->device, ndev); return RT_HIP_ERR_ARG; }
	RT_HIP_CHECK(hipSetDevice(cfg->device));
	RT_CHECK(rt_noalloc_setup(cfg->device));  // this GPU's address of the allocation-free entry's fixed scratch (k_noalloc.hip)
	rt_hip_ctx* ctx = new rt_hip_ctx();
	ctx->cfg = *cfg;
	memset(&ctx->stats, 0, sizeof(ctx->stats));
	{
		hipDeviceProp_t prop;
		if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess)
			ctx->resident_threads = prop.multiProcessorCount * prop.maxThreadsPerMultiProcessor;
		if (ctx->resident_threads <= 0) ctx->resident_threads = 256 * 2048;
		rt_default_tuning(&ctx->tuning);
		rt_clamp_tuning(ctx);
	}
	// Non-blocking: no implicit ordering against the legacy default stream (a blocking stream measured
	// 4 % slower per frame here).  Callers order their own streams against rt_hip_stream() explicitly;
	// the Python binding drains torch's current stream before every *_device call.
	hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
	if (e == hipSuccess) e = hipEventCreate(&ctx->ev0);
	if (e == hipSuccess) e = hipEventCreate(&ctx->ev1);
	if (e == hipSuccess) e = hipHostMalloc(&ctx->h_small, 4096, hipHostMallocDefault);
	if (e == hipSuccess) e = hipHostMalloc(&ctx->h_pipe, 512 * RT_MAX_PENDING, hipHostMallocDefault);
	for (int k = 0; k < RT_MAX_PENDING && e == hipSuccess; ++k) {
		ctx->pending[k].h_fc = (int*)((char*)ctx->h_pipe + 512 * k);
		e = hipEventCreate(&ctx->pending[k].ev0);
		if (e == hipSuccess) e = hipEventCreate(&ctx->pending[k].ev1);
	}
	if (e != hipSuccess) {
		rt_set_error("context creation failed: %s", hipGetErrorString(e));
		delete ctx;
		return RT_HIP_ERR_HIP;
	}
	*out = ctx;
	return RT_HIP_OK;
}

int rt_hip_destroy(rt_hip_ctx* ctx)
{
	if (!ctx) return RT_HIP_OK;
	(void)hipSetDevice(ctx->cfg.device);
	if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
	if (ctx->comm) (void)rt_hip_comm_destroy(ctx);
	DevBuf* bufs[] = {&ctx->d_verts, &ctx->d_normals, &ctx->d_texcoords, &ctx->d_faces, &ctx->d_mesh_of_tri, &ctx->d_mesh_xf,
	                  &ctx->d_tri_pos, &ctx->d_tri_attr, &ctx->d_tri_isect, &ctx->d_tri_box, &ctx->d_nodes, &ctx->d_qnodes, &ctx->d_spheres,
	                  &ctx->d_materials, &ctx->d_lights, &ctx->d_morton, &ctx->d_morton_tmp, &ctx->d_order, &ctx->d_order_tmp,
	                  &ctx->d_hist, &ctx->d_flags, &ctx->d_parent_leaf, &ctx->d_bounds, &ctx->d_misc, &ctx->d_org, &ctx->d_dir,
	                  &ctx->d_hits, &ctx->d_occ, &ctx->d_in_a, &ctx->d_in_b, &ctx->d_records, &ctx->d_offsets, &ctx->d_keys,
	                  &ctx->d_rank, &ctx->d_counts, &ctx->d_nodes_a, &ctx->d_nodes_b, &ctx->d_rays_q, &ctx->d_ray_parent,
	                  &ctx->d_accum, &ctx->d_qcount, &ctx->d_gate, &ctx->d_photons, &ctx->d_photon_cell, &ctx->d_photon_sorted,
	                  &ctx->d_cell_start, &ctx->d_photon_tmp, &ctx->d_census, &ctx->d_spill, &ctx->d_ploc, &ctx->d_tile_list, &ctx->d_all_tiles,
	                  &ctx->d_tex, &ctx->d_texels, &ctx->d_node_uv, &ctx->d_heavy, &ctx->d_node_inc, &ctx->d_child_slot, &ctx->d_group_boxes, &ctx->d_prefix, &ctx->d_fcount, &ctx->d_leaf_order, &ctx->d_photon_emit, &ctx->d_level_nodes,
	                  &ctx->d_mg_send, &ctx->d_mg_recv, &ctx->d_mg_counts, &ctx->d_mg_photons,
	                  &ctx->d_photons_f, &ctx->d_photon_sorted_f, &ctx->d_cell_start_f};
	for (DevBuf* b : bufs) b->release();
	for (auto& pr : ctx->trace_events) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
	if (ctx->h_records) (void)hipHostFree(ctx->h_records);
	if (ctx->h_offsets) (void)hipHostFree(ctx->h_offsets);
	if (ctx->h_small) (void)hipHostFree(ctx->h_small);
	if (ctx->h_pipe) (void)hipHostFree(ctx->h_pipe);
	if (ctx->up_stream) { (void)hipStreamSynchronize(ctx->up_stream); (void)hipStreamDestroy(ctx->up_stream); }
	if (ctx->down_stream) { (void)hipStreamSynchronize(ctx->down_stream); (void)hipStreamDestroy(ctx->down_stream); }
	for (int k = 0; k < RT_MAX_PENDING; ++k) {
		if (ctx->lane_stream[k]) { (void)hipStreamSynchronize(ctx->lane_stream[k]); (void)hipStreamDestroy(ctx->lane_stream[k]); }
		FrameBufs& fb = ctx->lane_bufs[k];
		for (DevBuf* b : {&fb.nodes_a, &fb.nodes_b, &fb.gate, &fb.node_inc, &fb.node_uv, &fb.rays_q, &fb.ray_parent, &fb.records, &fb.hits, &fb.occ,
		                  &fb.child_slot, &fb.fcount, &fb.group_boxes, &fb.prefix, &ctx->d_mg_send_slot[k], &ctx->d_mg_recv_slot[k]})
			b->release();
		if (ctx->pending[k].ev0) (void)hipEventDestroy(ctx->pending[k].ev0);
		if (ctx->pending[k].ev1) (void)hipEventDestroy(ctx->pending[k].ev1);
		for (hipEvent_t e : {ctx->pending[k].ev_up, ctx->pending[k].ev_done, ctx->pending[k].ev_down})
			if (e) (void)hipEventDestroy(e);
		ctx->d_pipe_pos[k].release(); ctx->d_pipe_nrm[k].release(); ctx->d_pipe_img[k].release(); ctx->d_pipe_g16[k].release();
	}
	if (ctx->comm_stream) { (void)hipStreamSynchronize(ctx->comm_stream); (void)hipStreamDestroy(ctx->comm_stream); }
	if (ctx->ev_order) (void)hipEventDestroy(ctx->ev_order);
	if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
	if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
	if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
	delete ctx;
	return RT_HIP_OK;
}

int rt_hip_set_config(rt_hip_ctx* ctx, const rt_hip_config* cfg)
{
	RT_CHECK(check_ctx(ctx));
	if (!cfg) { rt_set_error("null config"); return RT_HIP_ERR_ARG; }
	ctx->cfg.bounces = cfg->bounces;
	ctx->cfg.do_shadows = cfg->do_shadows;
	ctx->cfg.use_photon_map = cfg->use_photon_map;
	ctx->cfg.photon_count = cfg->photon_count;
	ctx->cfg.photon_estimate = cfg->photon_estimate;
	ctx->cfg.photon_radius = cfg->photon_radius;
	ctx->cfg.photon_max_dist = cfg->photon_max_dist;
	ctx->cfg.photon_absorb = cfg->photon_absorb;
	ctx->cfg.seed = cfg->seed;
	ctx->cfg.profile = cfg->profile;
	ctx->cfg.shadow_t_max = cfg->shadow_t_max;
	ctx->cfg.census = cfg->census;
	ctx->cfg.eps_det = cfg->eps_det;
	ctx->cfg.eps_t = cfg->eps_t;
	ctx->cfg.t_max = cfg->t_max;
	return RT_HIP_OK;
}

int rt_hip_get_tuning(rt_hip_ctx* ctx, rt_hip_tuning* out)
{
	RT_CHECK(check_ctx(ctx));
	if (!out) { rt_set_error("null output"); return RT_HIP_ERR_ARG; }
	*out = ctx->tuning;
	return RT_HIP_OK;
}

int rt_hip_set_tuning(rt_hip_ctx* ctx, const rt_hip_tuning* tuning)
{
	RT_CHECK(check_ctx(ctx));
	if (!tuning) { rt_set_error("null tuning"); return RT_HIP_ERR_ARG; }
	if (ctx->pending_count > 0) { rt_set_error("%d frame(s) are enqueued: rt_hip_frame_finish() them before changing the tuning", ctx->pending_count); return RT_HIP_ERR_STATE; }
	RT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
	const rt_hip_tuning old = ctx->tuning;
	ctx->tuning = *tuning;
	rt_clamp_tuning(ctx);
	const rt_hip_tuning& t = ctx->tuning;
	// what the built structure depends on: rebuild at the next update
	if (t.auto_accel != old.auto_accel || t.ploc_radius != old.ploc_radius || t.max_bvh_depth != old.max_bvh_depth) {
		ctx->dirty_topology = true;
		ctx->scene_dirty = true;
	}
	// The HBM part of the traversal stacks follows the LDS depth and the grid cap (max_trace_threads is both the grid cap
	// and the stride of that array, and rt_clamp_tuning has just changed it): resize it now whenever a tree is in use, also
	// when the scene is dirty for tables or matrices only -- those updates keep the tree (a pending rebuild sizes it itself).
	if (ctx->use_bvh && !ctx->dirty_topology) RT_CHECK(rt_reserve_spill(ctx));
	return RT_HIP_OK;
}

int rt_hip_invalidate(rt_hip_ctx* ctx)
{
	RT_CHECK(check_ctx(ctx));
	ctx->dirty_topology = true;
	ctx->dirty_tables = true;
	ctx->scene_dirty = true;
	return RT_HIP_OK;
}

// ---- scene ---------------------------------------------------------------------------------
int rt_hip_set_materials(rt_hip_ctx* ctx, const rt_material* materials, int n)
{
	RT_CHECK(check_ctx(ctx));
	if (n < 0 || (n > 0 && !materials)) { rt_set_error("bad material table"); return RT_HIP_ERR_ARG; }
	ctx->materials.assign(materials, materials + n);
	ctx->textures.clear();  // a texture belongs to its Material object (data/data.h:25): new table, no textures
	ctx->dirty_tables = true;
	ctx->scene_dirty = true;
	return RT_HIP_OK;
}

int rt_hip_set_texture(rt_hip_ctx* ctx, int material_id, const uint8_t* rgba, int width, int height)
{
	RT_CHECK(check_ctx(ctx));
	if (material_id < 0 || material_id >= (int)ctx->materials.size()) { rt_set_error("material %d out of range", material_id); return RT_HIP_ERR_ARG; }
	if (rgba && (width <= 0 || height <= 0 || (int64_t)width * height > (int64_t)1 << 28)) { rt_set_error("bad texture size %d x %d", width, height); return RT_HIP_ERR_ARG; }
	if ((int)ctx->textures.size() <= material_id) ctx->textures.resize(material_id + 1);
	HostTexture& t = ctx->textures[material_id];
	t.w = t.h = 0; t.texels.clear();
	if (rgba) {
		t.w = width; t.h = height;
		t.texels.resize((size_t)width * height);
		memcpy(t.texels.data(), rgba, (size_t)width * height * 4);
	}
	ctx->dirty_tables = true;
	ctx->scene_dirty = true;
	return RT_HIP_OK;
}

int rt_hip_set_frame_texcoords_device(rt_hip_ctx* ctx, const float* const* d_uv, int nframes)
{
	RT_CHECK(check_ctx(ctx));
	if (nframes < 0 || nframes > RT_MAX_BATCH || (nframes > 0 && !d_uv)) { rt_set_error("bad arguments"); return RT_HIP_ERR_ARG; }
	for (int f = 0; f < RT_MAX_BATCH; ++f) ctx->frame_uv[f] = f < nframes ? d_uv[f] : nullptr;
	ctx->frame_uv_count = nframes;
	return RT_HIP_OK;
}

int rt_hip_set_lights(rt_hip_ctx* ctx, const rt_light* lights, int n)
{
	RT_CHECK(check_ctx(ctx));
	if (n < 0 || (n > 0 && !lights)) { rt_set_error("bad light table"); return RT_HIP_ERR_ARG; }
	ctx->lights.assign(lights, lights + n);
	ctx->dirty_tables = true;
	ctx->scene_dirty = true;
	return RT_HIP_OK;
}

int rt_hip_add_sphere(rt_hip_ctx* ctx, float x, float y, float z, float radius)
{
	RT_CHECK(check_ctx(ctx));
	if (!ctx->sphere_rng_seeded) {
		srand(420);  // AccelSpheres::AccelSpheres (rt_accel_spheres.cpp:11)
		ctx->sphere_rng_seeded = true;
	}
	HostSphere s;
	s.c[0] = x; s.c[1] = y; s.c[2] = z; s.r = radius;
	// Vector3f(rand()%256, rand()%256, rand()%256): g++ evaluates the arguments right to left,
	// so the first draw is the blue channel (pinned against the compiled reference).
	s.diffuse[2] = (float)(rand() % 256);
	s.diffuse[1] = (float)(rand() % 256);
	s.diffuse[0] = (float)(rand() % 256);
	ctx->spheres.push_back(s);
	ctx->dirty_topology = true;
	ctx->scene_dirty = true;
	return (int)ctx->spheres.size() - 1;
}

int rt_hip_add_mesh(rt_hip_ctx* ctx, const float* verts, const float* normals, const float* texcoords, int nverts,
                    const uint32_t* faces, int nfaces, const float* model_mat16, int material_id,
                    const float* position3)
{
	RT_CHECK(check_ctx(ctx));
	if (nverts < 0 || nfaces < 0 || (nverts > 0 && (!verts || !normals || !texcoords)) || (nfaces > 0 && !faces)) {
		rt_set_error("bad mesh arrays");
		return RT_HIP_ERR_ARG;
	}
	for (int i = 0; i < nfaces * 3; ++i)
		if (faces[i] >= (uint32_t)nverts) { rt_set_error("face index %u out of range (%d verts)", faces[i], nverts); return RT_HIP_ERR_ARG; }
	HostMesh m;
	m.verts.assign(verts, verts + (size_t)nverts * 3);
	m.normals.assign(normals, normals + (size_t)nverts * 3);
	m.texcoords.assign(texcoords, texcoords + (size_t)nverts * 2);
	m.faces.assign(faces, faces + (size_t)nfaces * 3);
	if (model_mat16) memcpy(m.model, model_mat16, sizeof(m.model));
	else { memset(m.model, 0, sizeof(m.model)); m.model[0] = m.model[5] = m.model[10] = m.model[15] = 1.0f; }
	m.material = material_id;
	for (int k = 0; k < 3; ++k) m.position[k] = position3 ? position3[k] : 0.0f;
	m.first_tri = 0;
	m.vert_offset = 0;
	ctx->meshes.push_back(m);
	if (ctx->cfg.geometry == RT_HIP_SPHERES) {
		// AccelSpheres::addMesh: a unit sphere at Mesh::position (rt_accel_spheres.cpp:27-31)
		int s = rt_hip_add_sphere(ctx, m.position[0], m.position[1], m.position[2], 1.0f);
		if (s < 0) return s;
	}
	ctx->dirty_topology = true;
	ctx->scene_dirty = true;
	return (int)ctx->meshes.size() - 1;
}

int rt_hip_set_model_mat(rt_hip_ctx* ctx, int mesh_id, const float* model_mat16)
{
	RT_CHECK(check_ctx(ctx));
	if (mesh_id < 0 || mesh_id >= (int)ctx->meshes.size() || !model_mat16) { rt_set_error("bad mesh id"); return RT_HIP_ERR_ARG; }
	// Change detection: the reference's processRays re-reads every Mesh::model_mat each frame (rt_raytracer.cpp:467);
	// a matrix equal to the one in use changes nothing, and without use_model_mat (AccelArray semantics,
	// rt_accel_array.cpp:17-32) the geometry does not depend on it at all.
	if (memcmp(ctx->meshes[mesh_id].model, model_mat16, sizeof(float) * 16) == 0) return RT_HIP_OK;
	memcpy(ctx->meshes[mesh_id].model, model_mat16, sizeof(float) * 16);
	if (!ctx->cfg.use_model_mat || ctx->cfg.geometry != RT_HIP_TRIANGLES) return RT_HIP_OK;
	ctx->dirty_xf = true;
	ctx->scene_dirty = true;
	return RT_HIP_OK;
}

int rt_hip_set_model_mats(rt_hip_ctx* ctx, int first_mesh, int n, const float* model_mats16)
{
	RT_CHECK(check_ctx(ctx));
	if (first_mesh < 0 || n < 0 || (int64_t)first_mesh + n > (int64_t)ctx->meshes.size() || (n > 0 && !model_mats16)) {
		rt_set_error("bad mesh range [%d, %d) of %zu meshes", first_mesh, first_mesh + n, ctx->meshes.size());
		return RT_HIP_ERR_ARG;
	}
	for (int i = 0; i < n; ++i) RT_CHECK(rt_hip_set_model_mat(ctx, first_mesh + i, model_mats16 + 16 * (size_t)i));
	return RT_HIP_OK;
}

int rt_hip_update(rt_hip_ctx* ctx)
{
	RT_CHECK(check_ctx(ctx));
	for (const HostMesh& m : ctx->meshes)
		if (ctx->cfg.geometry == RT_HIP_TRIANGLES && !m.faces.empty() &&
		    (m.material < 0 || m.material >= (int)ctx->materials.size())) {
			rt_set_error("mesh material id %d outside the material table (%zu)", m.material, ctx->materials.size());
			return RT_HIP_ERR_ARG;
		}
	if (ctx->pending_count > 0) { rt_set_error("%d frame(s) are enqueued: rt_hip_frame_finish() them before changing the scene", ctx->pending_count); return RT_HIP_ERR_STATE; }
	if (!ctx->scene_dirty) { ctx->stats.last_update = RT_HIP_UPDATE_NONE; return RT_HIP_OK; }
	RT_CHECK(rt_build_geometry(ctx));
	if (ctx->use_bvh) RT_CHECK(rt_reserve_spill(ctx));  // every path (tables, refit, rebuild) leaves the overflow stacks sized for the tuning in force
	ctx->scene_dirty = ctx->dirty_topology = ctx->dirty_tables = ctx->dirty_xf = false;
	return RT_HIP_OK;
}

// ---- IAccel batch entries ---------------------------------------------------------------------
int rt_hip_intersect(rt_hip_ctx* ctx, rt_ray* rays, int n, uint8_t* hit_out, uint32_t* material_out, rt_hit* hits_out)
{
	RT_CHECK(check_ready(ctx));
	if (n < 0 || (n > 0 && !rays)) { rt_set_error("bad ray array"); return RT_HIP_ERR_ARG; }
	if (n == 0) return RT_HIP_OK;
	hipStream_t st = ctx->stream;
	RT_CHECK(ctx->d_records.reserve((size_t)n * sizeof(rt_ray)));
	RT_CHECK(ctx->d_org.reserve((size_t)n * 12));
	RT_CHECK(ctx->d_dir.reserve((size_t)n * 12));
	RT_CHECK(ctx->d_hits.reserve((size_t)n * sizeof(Hit)));
	RT_CHECK(ctx->d_occ.reserve((size_t)n));
	RT_CHECK(ctx->d_keys.reserve((size_t)n * 4));
	rt_ray* d_rays = ctx->d_records.as<rt_ray>();
	RT_HIP_CHECK(hipMemcpyAsync(d_rays, rays, (size_t)n * sizeof(rt_ray), hipMemcpyHostToDevice, st));
	RT_CHECK(rt_launch_split_rays(ctx, d_rays, n, ctx->d_org.as<float>(), ctx->d_dir.as<float>()));
	if (ctx->cfg.geometry == RT_HIP_SPHERES) {
		RT_CHECK(ctx->d_in_a.reserve((size_t)n * 4));
		RT_CHECK(ctx->d_in_b.reserve((size_t)n * 24));
		int* d_index = ctx->d_in_a.as<int>();
		float* d_pos = ctx->d_in_b.as<float>();
		float* d_nrm = d_pos + (size_t)n * 3;
		RT_CHECK(rt_launch_sphere_intersect(ctx, ctx->d_org.as<float>(), ctx->d_dir.as<float>(), n, d_index, d_pos, d_nrm));
		RT_CHECK(rt_fill_sphere_records(ctx, d_index, d_pos, d_nrm, n, d_rays, ctx->d_occ.as<uint8_t>(), ctx->d_keys.as<uint32_t>()));
		if (hits_out) memset(hits_out, 0xFF, (size_t)n * sizeof(rt_hit));
	} else {
		RT_CHECK(rt_launch_closest(ctx, ctx->d_org.as<float>(), ctx->d_dir.as<float>(), n, ctx->d_hits.as<Hit>()));
		RT_CHECK(rt_launch_fill_records(ctx, ctx->d_org.as<float>(), ctx->d_dir.as<float>(), ctx->d_hits.as<Hit>(), n, d_rays,
		                                ctx->d_occ.as<uint8_t>(), ctx->d_keys.as<uint32_t>()));
		if (hits_out) RT_HIP_CHECK(hipMemcpyAsync(hits_out, ctx->d_hits.p, (size_t)n * sizeof(rt_hit), hipMemcpyDeviceToHost, st));
	}
	RT_HIP_CHECK(hipMemcpyAsync(rays, d_rays, (size_t)n * sizeof(rt_ray), hipMemcpyDeviceToHost, st));
	if (hit_out) RT_HIP_CHECK(hipMemcpyAsync(hit_out, ctx->d_occ.p, (size_t)n, hipMemcpyDeviceToHost, st));
	if (material_out) RT_HIP_CHECK(hipMemcpyAsync(material_out, ctx->d_keys.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
	RT_HIP_CHECK(hipStreamSynchronize(st));
	RT_CHECK(rt_collect_timing(ctx));
	return RT_HIP_OK;
}

int rt_hip_occlude(rt_hip_ctx* ctx, const rt_ray* rays, int n, uint8_t* occluded_out)
{
	RT_CHECK(check_ready(ctx));
	if (n < 0 || (n > 0 && (!rays || !occluded_out))) { rt_set_error("bad arguments"); return RT_HIP_ERR_ARG; }
	if (n == 0) return RT_HIP_OK;
	hipStream_t st = ctx->stream;
	RT_CHECK(ctx->d_records.reserve((size_t)n * sizeof(rt_ray)));
	RT_CHECK(ctx->d_org.reserve((size_t)n * 12));
	RT_CHECK(ctx->d_dir.reserve((size_t)n * 12));
	RT_CHECK(ctx->d_occ.reserve((size_t)n));
	rt_ray* d_rays = ctx->d_records.as<rt_ray>();
	RT_HIP_CHECK(hipMemcpyAsync(d_rays, rays, (size_t)n * sizeof(rt_ray), hipMemcpyHostToDevice, st));
	RT_CHECK(rt_launch_split_rays(ctx, d_rays, n, ctx->d_org.as<float>(), ctx->d_dir.as<float>()));
	if (ctx->cfg.geometry == RT_HIP_SPHERES)
		RT_CHECK(rt_launch_sphere_occlude(ctx, ctx->d_org.as<float>(), ctx->d_dir.as<float>(), n, ctx->d_occ.as<uint8_t>()));
	else
		RT_CHECK(rt_launch_occlude_ref(ctx, ctx->d_org.as<float>(), ctx->d_dir.as<float>(), n, ctx->d_occ.as<uint8_t>()));
	RT_HIP_CHECK(hipMemcpyAsync(occluded_out, ctx->d_occ.p, (size_t)n, hipMemcpyDeviceToHost, st));
	RT_HIP_CHECK(hipStreamSynchronize(st));
	return RT_HIP_OK;
}

int rt_hip_sphere_colors(rt_hip_ctx* ctx, float* rgb_out, int max_spheres)
{
	RT_CHECK(check_ctx(ctx));
	if (!rgb_out) { rt_set_error("null output"); return RT_HIP_ERR_ARG; }
	int n = std::min<int>(max_spheres, (int)ctx->spheres.size());
	for (int i = 0; i < n; ++i)
		for (int k = 0; k < 3; ++k) rgb_out[3 * i + k] = ctx->spheres[i].diffuse[k];
	return n;
}

// ---- processRays ----------------------------------------------------------------------------
int rt_hip_process_gbuffer_device(rt_hip_ctx* ctx, const float* d_positions_xyz, const float* d_normals_xyz, int count,
                                  rt_ray* d_results, rt_result_offset* d_offsets, int* out_result_count,
                                  int* out_material_count)
{
	RT_CHECK(check_ready(ctx));
	if (count < 0 || !out_result_count || !out_material_count) { rt_set_error("bad arguments"); return RT_HIP_ERR_ARG; }
	*out_result_count = 0;
	*out_material_count = 0;
	ctx->last_count = 0;
	if (count == 0) return RT_HIP_OK;  // defined behaviour where the reference has UB (rt_raytracer.cpp:538)
	if (ctx->cfg.geometry != RT_HIP_TRIANGLES) { rt_set_error("processRays needs triangle geometry"); return RT_HIP_ERR_STATE; }
	hipStream_t st = ctx->stream;
	RT_CHECK(ctx->d_org.reserve((size_t)count * 12));
	RT_CHECK(ctx->d_dir.reserve((size_t)count * 12));
	RT_CHECK(ctx->d_hits.reserve((size_t)count * sizeof(Hit)));
	RT_HIP_CHECK(hipEventRecord(ctx->ev0, st));
	RT_CHECK(rt_launch_reflect_raygen(ctx, d_positions_xyz, d_normals_xyz, count, ctx->d_org.as<float>(), ctx->d_dir.as<float>()));
	RT_CHECK(rt_launch_closest(ctx, ctx->d_org.as<float>(), ctx->d_dir.as<float>(), count, ctx->d_hits.as<Hit>()));
	RT_CHECK(rt_group_by_material(ctx, ctx->d_org.as<float>(), ctx->d_dir.as<float>(), ctx->d_hits.as<Hit>(), count, d_results,
	                              d_offsets, out_result_count, out_material_count));
	RT_HIP_CHECK(hipEventRecord(ctx->ev1, st));
	RT_HIP_CHECK(hipEventSynchronize(ctx->ev1));
	float ms = 0;
	RT_HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
	ctx->stats.frame_ms += ms;
	ctx->stats.frames += 1;
	ctx->last_count = count;
	ctx->last_frame_rays = count;
	RT_CHECK(rt_collect_timing(ctx));
	return RT_HIP_OK;
}

int rt_hip_process_gbuffer(rt_hip_ctx* ctx, const float* positions_xyz, const float* normals_xyz, int count,
                           const rt_ray** results, const rt_result_offset** result_offsets, int* out_result_count,
                           int* out_material_count)
{
	RT_CHECK(check_ready(ctx));
	if (count < 0 || !results || !result_offsets || !out_result_count || !out_material_count ||
	    (count > 0 && (!positions_xyz || !normals_xyz))) {
		rt_set_error("bad arguments");
		return RT_HIP_ERR_ARG;
	}
	hipStream_t st = ctx->stream;
	size_t nmat = std::max<size_t>(ctx->materials.size(), 1);
	RT_CHECK(rt_ensure_pinned(&ctx->h_records, &ctx->h_records_cap, std::max<size_t>((size_t)count, 1) * sizeof(rt_ray)));
	RT_CHECK(rt_ensure_pinned(&ctx->h_offsets, &ctx->h_offsets_cap, nmat * sizeof(rt_result_offset)));
	*results = (const rt_ray*)ctx->h_records;
	*result_offsets = (const rt_result_offset*)ctx->h_offsets;
	*out_result_count = 0;
	*out_material_count = 0;
	if (count == 0) { ctx->last_count = 0; return RT_HIP_OK; }
	// G-buffer samples are uploaded once per frame with hipMemcpyAsync (north_star (1))
	RT_CHECK(ctx->d_in_a.reserve((size_t)count * 12));
	RT_CHECK(ctx->d_in_b.reserve((size_t)count * 12));
	// d_in_a/d_in_b are reused by the grouping step, so stage the samples in dedicated buffers
	RT_CHECK(ctx->d_accum.reserve((size_t)count * 24));
	float* d_pos = ctx->d_accum.as<float>();
	float* d_nrm = d_pos + (size_t)count * 3;
	RT_HIP_CHECK(hipMemcpyAsync(d_pos, positions_xyz, (size_t)count * 12, hipMemcpyHostToDevice, st));
	RT_HIP_CHECK(hipMemcpyAsync(d_nrm, normals_xyz, (size_t)count * 12, hipMemcpyHostToDevice, st));
	RT_CHECK(ctx->d_records.reserve((size_t)count * sizeof(rt_ray)));
	RT_CHECK(ctx->d_offsets.reserve(nmat * sizeof(rt_result_offset)));
	int nres = 0, ngroups = 0;
	RT_CHECK(rt_hip_process_gbuffer_device(ctx, d_pos, d_nrm, count, ctx->d_records.as<rt_ray>(),
	                                       ctx->d_offsets.as<rt_result_offset>(), &nres, &ngroups));
	if (nres > 0) RT_HIP_CHECK(hipMemcpyAsync(ctx->h_records, ctx->d_records.p, (size_t)nres * sizeof(rt_ray), hipMemcpyDeviceToHost, st));
	if (ngroups > 0) RT_HIP_CHECK(hipMemcpyAsync(ctx->h_offsets, ctx->d_offsets.p, (size_t)ngroups * sizeof(rt_result_offset), hipMemcpyDeviceToHost, st));
	RT_HIP_CHECK(hipStreamSynchronize(st));
	*out_result_count = nres;
	*out_material_count = ngroups;
	return RT_HIP_OK;
}

int rt_hip_last_hits(rt_hip_ctx* ctx, rt_hit* hits_out, float* directions_out, int count)
{
	RT_CHECK(check_ctx(ctx));
	if (count < 0 || count > ctx->last_count) { rt_set_error("count %d exceeds the last batch (%d)", count, ctx->last_count); return RT_HIP_ERR_ARG; }
	if (count == 0) return RT_HIP_OK;
	if (hits_out) RT_HIP_CHECK(hipMemcpy(hits_out, ctx->d_hits.p, (size_t)count * sizeof(rt_hit), hipMemcpyDeviceToHost));
	if (directions_out) RT_HIP_CHECK(hipMemcpy(directions_out, ctx->d_dir.p, (size_t)count * 12, hipMemcpyDeviceToHost));
	return RT_HIP_OK;
}

// ---- raw batch tracing -----------------------------------------------------------------------
int rt_hip_trace_closest_device(rt_hip_ctx* ctx, const float* d_org, const float* d_dir, int n, rt_hit* d_hits)
{
	RT_CHECK(check_ready(ctx));
	if (n < 0 || (n > 0 && (!d_org || !d_dir || !d_hits))) { rt_set_error("bad arguments"); return RT_HIP_ERR_ARG; }
	if (ctx->cfg.geometry != RT_HIP_TRIANGLES) { rt_set_error("triangle geometry required"); return RT_HIP_ERR_STATE; }
	RT_CHECK(rt_launch_closest(ctx, d_org, d_dir, n, (Hit*)d_hits));
	return RT_HIP_OK;
}

int rt_hip_trace_any_device(rt_hip_ctx* ctx, const float* d_org, const float* d_dir, int n, uint8_t* d_occluded)
{
	RT_CHECK(check_ready(ctx));
	if (n < 0 || (n > 0 && (!d_org || !d_dir || !d_occluded))) { rt_set_error("bad arguments"); return RT_HIP_ERR_ARG; }
	if (ctx->cfg.geometry != RT_HIP_TRIANGLES) { rt_set_error("triangle geometry required"); return RT_HIP_ERR_STATE; }
	RT_CHECK(rt_launch_any(ctx, d_org, d_dir, n, d_occluded));
	return RT_HIP_OK;
}

int rt_hip_trace_census_device(rt_hip_ctx* ctx, const float* d_org, const float* d_dir, int n, int any_hit,
                               uint32_t* d_nodes_visited, uint32_t* d_tris_tested)
{
	RT_CHECK(check_ready(ctx));
	if (n < 0 || (n > 0 && (!d_org || !d_dir || !d_nodes_visited || !d_tris_tested))) { rt_set_error("bad arguments"); return RT_HIP_ERR_ARG; }
	RT_CHECK(rt_launch_census(ctx, d_org, d_dir, n, any_hit, d_nodes_visited, d_tris_tested));
	return RT_HIP_OK;
}

// ---- G-buffer hand-off -----------------------------------------------------------------------
int rt_hip_gbuffer_select_device(rt_hip_ctx* ctx, const float* d_nrm_rgba, const float* d_pos_rgb, int n, int interlace,
                                 int phase, float* d_positions_out, float* d_normals_out, int* d_index_out,
                                 int* out_count)
{
	RT_CHECK(check_ctx(ctx));
	if (n < 0 || !out_count || (n > 0 && (!d_nrm_rgba || !d_pos_rgb || !d_positions_out || !d_normals_out))) {
		rt_set_error("bad arguments");
		return RT_HIP_ERR_ARG;
	}
	return rt_gbuffer_select(ctx, d_nrm_rgba, d_pos_rgb, n, interlace, phase, d_positions_out, d_normals_out, d_index_out, out_count);
}

int rt_hip_gbuffer_select_noalloc_device(rt_hip_ctx* ctx, const float* d_nrm_rgba, const float* d_pos_rgb, int n, int interlace,
                                         int phase, float* d_positions_out, float* d_normals_out, int* d_index_out,
                                         int* d_count_out, int* out_count)
{
	RT_CHECK(check_ctx(ctx));
	if (n < 0 || (!out_count && !d_count_out) || (n > 0 && (!d_nrm_rgba || !d_pos_rgb || !d_positions_out || !d_normals_out))) {
		rt_set_error("bad arguments");
		return RT_HIP_ERR_ARG;
	}
	return rt_gbuffer_select_noalloc(ctx, d_nrm_rgba, d_pos_rgb, n, interlace, phase, d_positions_out, d_normals_out, d_index_out,
	                                 d_count_out, out_count);
}

int rt_hip_render_gbuffer_device(rt_hip_ctx* ctx, float fov, int width, int height, float* d_nrm_rgba, float* d_pos_rgb)
{
	RT_CHECK(check_ready(ctx));
	if (width <= 0 || height <= 0 || !d_nrm_rgba || !d_pos_rgb) { rt_set_error("bad arguments"); return RT_HIP_ERR_ARG; }
	return rt_render_gbuffer(ctx, fov, width, height, d_nrm_rgba, d_pos_rgb);
}

int rt_hip_render_gbuffer_uv_device(rt_hip_ctx* ctx, float fov, int width, int height, float* d_nrm_rgba, float* d_pos_rgb,
                                    float* d_uv)
{
	RT_CHECK(check_ready(ctx));
	if (width <= 0 || height <= 0 || !d_nrm_rgba || !d_pos_rgb || !d_uv) { rt_set_error("bad arguments"); return RT_HIP_ERR_ARG; }
	return rt_render_gbuffer(ctx, fov, width, height, d_nrm_rgba, d_pos_rgb, d_uv);
}

// ---- full frame --------------------------------------------------------------------------------
int rt_hip_render_tiles_device(rt_hip_ctx* ctx, const float* d_gbuf_pos_rgb, const float* d_gbuf_nrm_rgba, int width,
                               int height, int tile, int rank, int nranks, uint8_t* d_rgba_out)
{
	RT_CHECK(check_ready(ctx));
	if (width <= 0 || height <= 0 || !d_gbuf_pos_rgb || !d_gbuf_nrm_rgba || !d_rgba_out || tile <= 0 || nranks <= 0 ||
	    rank < 0 || rank >= nranks) {
		rt_set_error("bad arguments");
		return RT_HIP_ERR_ARG;
	}
	return rt_render_frame(ctx, d_gbuf_pos_rgb, d_gbuf_nrm_rgba, width, height, tile, rank, nranks, d_rgba_out);
}

int rt_hip_render_tiles_batch_device(rt_hip_ctx* ctx, const float* const* d_gbuf_pos_rgb, const float* const* d_gbuf_nrm_rgba,
                                     int nframes, int width, int height, int tile, int rank, int nranks,
                                     uint8_t* const* d_rgba_out)
{
	RT_CHECK(check_ready(ctx));
	if (width <= 0 || height <= 0 || nframes <= 0 || !d_gbuf_pos_rgb || !d_gbuf_nrm_rgba || !d_rgba_out || tile <= 0 ||
	    nranks <= 0 || rank < 0 || rank >= nranks) {
		rt_set_error("bad arguments");
		return RT_HIP_ERR_ARG;
	}
	for (int f = 0; f < nframes; ++f)
		if (!d_gbuf_pos_rgb[f] || !d_gbuf_nrm_rgba[f] || !d_rgba_out[f]) { rt_set_error("frame %d: null buffer", f); return RT_HIP_ERR_ARG; }
	return rt_render_frames(ctx, d_gbuf_pos_rgb, d_gbuf_nrm_rgba, nframes, width, height, tile, rank, nranks, d_rgba_out);
}

int rt_hip_render_tiles_packed_device(rt_hip_ctx* ctx, const float* const* d_gbuf_pos_rgb, const float* const* d_gbuf_nrm_rgba,
                                      int nframes, int width, int height, int tile, int rank, int nranks,
                                      uint8_t* const* d_packed_out)
{
	RT_CHECK(check_ready(ctx));
	if (width <= 0 || height <= 0 || nframes <= 0 || !d_gbuf_pos_rgb || !d_gbuf_nrm_rgba || !d_packed_out || tile <= 0 ||
	    tile % 8 != 0 || nranks <= 1 || nranks > RT_MAX_BATCH || rank < 0 || rank >= nranks) {
		rt_set_error("bad arguments (packed output needs 1 < nranks <= 16 and tile %% 8 == 0)");
		return RT_HIP_ERR_ARG;
	}
	for (int f = 0; f < nframes; ++f)
		if (!d_gbuf_pos_rgb[f] || !d_gbuf_nrm_rgba[f] || !d_packed_out[f]) { rt_set_error("frame %d: null buffer", f); return RT_HIP_ERR_ARG; }
	return rt_render_frames(ctx, d_gbuf_pos_rgb, d_gbuf_nrm_rgba, nframes, width, height, tile, rank, nranks, d_packed_out, 1);
}

// One frame's packed share, queued without waiting (the frame pipeline's form of rt_hip_render_tiles_packed_device for a host
// that runs its own exchange); rt_hip_frame_finish completes the oldest enqueued frame.
int rt_hip_tiles_packed_enqueue_device(rt_hip_ctx* ctx, const float* d_gbuf_pos_rgb, const float* d_gbuf_nrm_rgba, int width, int height,
                                       int tile, int rank, int nranks, uint8_t* d_packed_out)
{
	RT_CHECK(check_ready(ctx));
	if (width <= 0 || height <= 0 || !d_gbuf_pos_rgb || !d_gbuf_nrm_rgba || !d_packed_out || tile <= 0 || tile % 8 != 0 || nranks <= 1 ||
	    nranks > RT_MAX_BATCH || rank < 0 || rank >= nranks) {
		rt_set_error("bad arguments (packed output needs 1 < nranks <= 16 and tile %% 8 == 0)");
		return RT_HIP_ERR_ARG;
	}
	return rt_frame_enqueue(ctx, d_gbuf_pos_rgb, d_gbuf_nrm_rgba, width, height, d_packed_out, tile, rank, nranks, 1);
}

// ... and the packed shares of several frames as one pipeline entry (one pass of launches for all of them)
int rt_hip_tiles_packed_batch_enqueue_device(rt_hip_ctx* ctx, const float* const* d_gbuf_pos_rgb, const float* const* d_gbuf_nrm_rgba,
                                             int nframes, int width, int height, int tile, int rank, int nranks, uint8_t* const* d_packed_out)
{
	RT_CHECK(check_ready(ctx));
	if (width <= 0 || height <= 0 || nframes <= 0 || nframes > RT_MAX_BATCH || !d_gbuf_pos_rgb || !d_gbuf_nrm_rgba || !d_packed_out || tile <= 0 ||
	    tile % 8 != 0 || nranks <= 1 || nranks > RT_MAX_BATCH || rank < 0 || rank >= nranks) {
		rt_set_error("bad arguments (packed output needs 1 < nranks <= 16, tile %% 8 == 0 and 1..16 frames)");
		return RT_HIP_ERR_ARG;
	}
	for (int f = 0; f < nframes; ++f)
		if (!d_gbuf_pos_rgb[f] || !d_gbuf_nrm_rgba[f] || !d_packed_out[f]) { rt_set_error("frame %d: null buffer", f); return RT_HIP_ERR_ARG; }
	return rt_frames_enqueue(ctx, d_gbuf_pos_rgb, d_gbuf_nrm_rgba, nframes, width, height, d_packed_out, tile, rank, nranks, 1);
}

int rt_hip_unpack_chunks_device(rt_hip_ctx* ctx, const uint8_t* d_chunks, int64_t chunk_bytes, int width, int height, int tile,
                                int nranks, uint8_t* d_frame_rgba)
{
	RT_CHECK(check_ctx(ctx));
	if (width <= 0 || height <= 0 || tile <= 0 || nranks <= 0 || nranks > RT_MAX_BATCH || chunk_bytes <= 0 || chunk_bytes % 4 != 0 ||
	    !d_chunks || !d_frame_rgba) {
		rt_set_error("bad arguments");
		return RT_HIP_ERR_ARG;
	}
	return rt_unpack_chunks(ctx, d_chunks, chunk_bytes, width, height, tile, nranks, d_frame_rgba);
}

int rt_hip_render_frame_device(rt_hip_ctx* ctx, const float* d_gbuf_pos_rgb, const float* d_gbuf_nrm_rgba, int width,
                               int height, uint8_t* d_rgba_out)
{
	return rt_hip_render_tiles_device(ctx, d_gbuf_pos_rgb, d_gbuf_nrm_rgba, width, height, 64, 0, 1, d_rgba_out);
}

int rt_hip_render_frame(rt_hip_ctx* ctx, const float* gbuf_pos_rgb, const float* gbuf_nrm_rgba, int width, int height,
                        uint8_t* rgba_out)
{
	RT_CHECK(check_ready(ctx));
	if (width <= 0 || height <= 0 || !gbuf_pos_rgb || !gbuf_nrm_rgba || !rgba_out) { rt_set_error("bad arguments"); return RT_HIP_ERR_ARG; }
	hipStream_t st = ctx->stream;
	size_t npix = (size_t)width * height;
	RT_CHECK(ctx->d_in_a.reserve(npix * 12));
	RT_CHECK(ctx->d_in_b.reserve(npix * 16));
	RT_CHECK(ctx->d_offsets.reserve(npix * 4));
	RT_HIP_CHECK(hipMemcpyAsync(ctx->d_in_a.p, gbuf_pos_rgb, npix * 12, hipMemcpyHostToDevice, st));
	RT_HIP_CHECK(hipMemcpyAsync(ctx->d_in_b.p, gbuf_nrm_rgba, npix * 16, hipMemcpyHostToDevice, st));
	RT_CHECK(rt_render_frame(ctx, ctx->d_in_a.as<float>(), ctx->d_in_b.as<float>(), width, height, 64, 0, 1, ctx->d_offsets.as<uint8_t>()));
	RT_HIP_CHECK(hipMemcpyAsync(rgba_out, ctx->d_offsets.p, npix * 4, hipMemcpyDeviceToHost, st));
	RT_HIP_CHECK(hipStreamSynchronize(st));
	return RT_HIP_OK;
}

int rt_hip_frame_enqueue_device(rt_hip_ctx* ctx, const float* d_gbuf_pos_rgb, const float* d_gbuf_nrm_rgba, int width, int height,
                                uint8_t* d_rgba_out)
{
	RT_CHECK(check_ready(ctx));
	if (width <= 0 || height <= 0 || !d_gbuf_pos_rgb || !d_gbuf_nrm_rgba || !d_rgba_out) { rt_set_error("bad arguments"); return RT_HIP_ERR_ARG; }
	return rt_frame_enqueue(ctx, d_gbuf_pos_rgb, d_gbuf_nrm_rgba, width, height, d_rgba_out);
}

// The same pipeline fed from host memory: the G-buffer of frame k + 1 crosses PCIe while frame k is traced, and frame k's
// image goes back while frame k + 1 is traced.  Three streams (up, the context's, down) chained by events per slot.
int rt_hip_frame_enqueue(rt_hip_ctx* ctx, const float* gbuf_pos_rgb, const float* gbuf_nrm_rgba, int width, int height, uint8_t* rgba_out)
{
	RT_CHECK(check_ready(ctx));
	if (width <= 0 || height <= 0 || !gbuf_pos_rgb || !gbuf_nrm_rgba || !rgba_out) { rt_set_error("bad arguments"); return RT_HIP_ERR_ARG; }
	int slot = 0;
	RT_CHECK(rt_pipeline_slot(ctx, &slot));
	PendingFrame* pf = &ctx->pending[slot];
	if (!ctx->up_stream) RT_HIP_CHECK(hipStreamCreateWithFlags(&ctx->up_stream, hipStreamNonBlocking));
	if (!ctx->down_stream) RT_HIP_CHECK(hipStreamCreateWithFlags(&ctx->down_stream, hipStreamNonBlocking));
	if (!pf->ev_up) RT_HIP_CHECK(hipEventCreateWithFlags(&pf->ev_up, hipEventDisableTiming));
	if (!pf->ev_done) RT_HIP_CHECK(hipEventCreateWithFlags(&pf->ev_done, hipEventDisableTiming));
	if (!pf->ev_down) RT_HIP_CHECK(hipEventCreateWithFlags(&pf->ev_down, hipEventDisableTiming));
	const size_t npix = (size_t)width * height;
	RT_CHECK(ctx->d_pipe_pos[slot].reserve(npix * 12));
	RT_CHECK(ctx->d_pipe_nrm[slot].reserve(npix * 16));
	RT_CHECK(ctx->d_pipe_img[slot].reserve(npix * 4));
	// the slot's staging is free: its previous frame was finished before the slot could be taken again
	RT_HIP_CHECK(hipMemcpyAsync(ctx->d_pipe_pos[slot].p, gbuf_pos_rgb, npix * 12, hipMemcpyHostToDevice, ctx->up_stream));
	RT_HIP_CHECK(hipMemcpyAsync(ctx->d_pipe_nrm[slot].p, gbuf_nrm_rgba, npix * 16, hipMemcpyHostToDevice, ctx->up_stream));
	RT_HIP_CHECK(hipEventRecord(pf->ev_up, ctx->up_stream));
	RT_HIP_CHECK(hipStreamWaitEvent(ctx->stream, pf->ev_up, 0));
	RT_CHECK(rt_frame_enqueue(ctx, ctx->d_pipe_pos[slot].as<float>(), ctx->d_pipe_nrm[slot].as<float>(), width, height,
	                          ctx->d_pipe_img[slot].as<uint8_t>()));
	RT_HIP_CHECK(hipEventRecord(pf->ev_done, pf->stream));  // (the stream of the lane the frame was dealt to)
	RT_HIP_CHECK(hipStreamWaitEvent(ctx->down_stream, pf->ev_done, 0));
	RT_HIP_CHECK(hipMemcpyAsync(rgba_out, ctx->d_pipe_img[slot].p, npix * 4, hipMemcpyDeviceToHost, ctx->down_stream));
	RT_HIP_CHECK(hipEventRecord(pf->ev_down, ctx->down_stream));
	pf->host = true;
	pf->h_rgba = rgba_out;
	pf->img_bytes = npix * 4;
	return RT_HIP_OK;
}

// rt_hip_frame_enqueue with the G-buffer in the formats the rasterizer's FBO holds it in (RGBA16 UNORM normals,
// RGB16F positions: ras_rasterizer.cpp:293,301) instead of the floats glReadPixels widens them to (:497,543): half the
// bytes over PCIe (14 instead of 28 per texel), decoded on the device (exact), same frame bit for bit.
int rt_hip_frame_enqueue_g16(rt_hip_ctx* ctx, const uint16_t* gbuf_nrm_rgba16, const uint16_t* gbuf_pos_rgb16f, int width, int height,
                             uint8_t* rgba_out)
{
	RT_CHECK(check_ready(ctx));
	if (width <= 0 || height <= 0 || !gbuf_nrm_rgba16 || !gbuf_pos_rgb16f || !rgba_out) { rt_set_error("bad arguments"); return RT_HIP_ERR_ARG; }
	int slot = 0;
	RT_CHECK(rt_pipeline_slot(ctx, &slot));
	PendingFrame* pf = &ctx->pending[slot];
	if (!ctx->up_stream) RT_HIP_CHECK(hipStreamCreateWithFlags(&ctx->up_stream, hipStreamNonBlocking));
	if (!ctx->down_stream) RT_HIP_CHECK(hipStreamCreateWithFlags(&ctx->down_stream, hipStreamNonBlocking));
	if (!pf->ev_up) RT_HIP_CHECK(hipEventCreateWithFlags(&pf->ev_up, hipEventDisableTiming));
	if (!pf->ev_done) RT_HIP_CHECK(hipEventCreateWithFlags(&pf->ev_done, hipEventDisableTiming));
	if (!pf->ev_down) RT_HIP_CHECK(hipEventCreateWithFlags(&pf->ev_down, hipEventDisableTiming));
	const size_t npix = (size_t)width * height;
	RT_CHECK(ctx->d_pipe_g16[slot].reserve(npix * 14));
	RT_CHECK(ctx->d_pipe_pos[slot].reserve(npix * 12));
	RT_CHECK(ctx->d_pipe_nrm[slot].reserve(npix * 16));
	RT_CHECK(ctx->d_pipe_img[slot].reserve(npix * 4));
	uint16_t* d_n16 = ctx->d_pipe_g16[slot].as<uint16_t>();
	uint16_t* d_p16 = d_n16 + npix * 4;
	RT_HIP_CHECK(hipMemcpyAsync(d_n16, gbuf_nrm_rgba16, npix * 8, hipMemcpyHostToDevice, ctx->up_stream));
	RT_HIP_CHECK(hipMemcpyAsync(d_p16, gbuf_pos_rgb16f, npix * 6, hipMemcpyHostToDevice, ctx->up_stream));
	RT_HIP_CHECK(hipEventRecord(pf->ev_up, ctx->up_stream));
	RT_HIP_CHECK(hipStreamWaitEvent(ctx->stream, pf->ev_up, 0));
	RT_CHECK(rt_launch_decode_g16(ctx, d_n16, d_p16, (int)npix, ctx->d_pipe_nrm[slot].as<float>(), ctx->d_pipe_pos[slot].as<float>()));
	RT_CHECK(rt_frame_enqueue(ctx, ctx->d_pipe_pos[slot].as<float>(), ctx->d_pipe_nrm[slot].as<float>(), width, height,
	                          ctx->d_pipe_img[slot].as<uint8_t>()));
	RT_HIP_CHECK(hipEventRecord(pf->ev_done, pf->stream));  // (the stream of the lane the frame was dealt to)
	RT_HIP_CHECK(hipStreamWaitEvent(ctx->down_stream, pf->ev_done, 0));
	RT_HIP_CHECK(hipMemcpyAsync(rgba_out, ctx->d_pipe_img[slot].p, npix * 4, hipMemcpyDeviceToHost, ctx->down_stream));
	RT_HIP_CHECK(hipEventRecord(pf->ev_down, ctx->down_stream));
	pf->host = true;
	pf->h_rgba = rgba_out;
	pf->img_bytes = npix * 4;
	return RT_HIP_OK;
}

int rt_hip_frame_finish(rt_hip_ctx* ctx, int64_t* out_rays)
{
	RT_CHECK(check_ctx(ctx));
	PendingFrame* pf = ctx->pending_count > 0 ? &ctx->pending[ctx->pending_head] : nullptr;
	const bool comm = pf && pf->comm;                 // rt_hip_frame_multi_enqueue_device: ev_down = the assembled image is in place
	const bool host = pf && (pf->host || comm);
	hipEvent_t ev_down = host ? pf->ev_down : nullptr;
	uint8_t* h_rgba = host ? pf->h_rgba : nullptr;
	const uint8_t* d_img = host ? pf->fs.rgba[0] : nullptr;
	const size_t bytes = host ? pf->img_bytes : 0;
	if (pf) { pf->host = false; pf->comm = false; }
	const int status = rt_frame_finish(ctx, out_rays);
	if (host) {
		// also when the finish failed: the caller's buffer must not be written to after this call has returned
		RT_HIP_CHECK(hipEventSynchronize(ev_down));
		RT_CHECK(status);
		if (ctx->frame_redone && !comm) {
			// the frame was rendered again after the download had been queued (a level overflowed its bound): fetch it again
			RT_HIP_CHECK(hipMemcpyAsync(h_rgba, d_img, bytes, hipMemcpyDeviceToHost, ctx->stream));
			RT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
		}
	}
	return status;
}

int64_t rt_hip_last_frame_rays(rt_hip_ctx* ctx) { return ctx ? ctx->last_frame_rays : 0; }

// ---- photon map ----------------------------------------------------------------------------------
int rt_hip_photon_emit_device(rt_hip_ctx* ctx, int photons_per_light, int64_t first, int64_t count, float* d_photons_out,
                              int64_t capacity, int64_t* out_stored)
{
	RT_CHECK(check_ready(ctx));
	if (photons_per_light <= 0 || first < 0 || count < 0 || !d_photons_out || capacity < count || !out_stored) {
		rt_set_error("bad arguments");
		return RT_HIP_ERR_ARG;
	}
	return rt_photon_emit(ctx, photons_per_light, first, count, d_photons_out, capacity, out_stored);
}

int rt_hip_photon_generate(rt_hip_ctx* ctx, int photons_per_light, int64_t* out_stored)
{
	RT_CHECK(check_ready(ctx));
	if (photons_per_light <= 0 || !out_stored) { rt_set_error("bad arguments"); return RT_HIP_ERR_ARG; }
	*out_stored = 0;
	const int64_t total = (int64_t)photons_per_light * (int64_t)ctx->lights.size();
	if (total == 0) return rt_photon_build_grid(ctx, nullptr, 0);
	RT_CHECK(ctx->d_photon_emit.reserve((size_t)total * 36));
	int64_t stored = 0;
	RT_CHECK(rt_photon_emit(ctx, photons_per_light, 0, total, ctx->d_photon_emit.as<float>(), total, &stored));
	RT_CHECK(rt_photon_build_grid(ctx, ctx->d_photon_emit.as<float>(), stored));
	*out_stored = stored;
	return RT_HIP_OK;
}

int rt_hip_photon_get(rt_hip_ctx* ctx, float* photons_out, int64_t capacity, int64_t* out_n)
{
	RT_CHECK(check_ctx(ctx));
	if (!out_n || capacity < 0 || (capacity > 0 && !photons_out)) { rt_set_error("bad arguments"); return RT_HIP_ERR_ARG; }
	*out_n = ctx->num_photons;
	if (ctx->num_photons == 0 || capacity == 0) return RT_HIP_OK;  // capacity 0: only the count is asked for
	if (capacity < ctx->num_photons) { rt_set_error("capacity %lld < %lld photons in the map", (long long)capacity, (long long)ctx->num_photons); return RT_HIP_ERR_ARG; }
	RT_CHECK(ctx->d_photon_tmp.reserve((size_t)ctx->num_photons * 36));
	RT_CHECK(rt_photon_unpermute(ctx, ctx->d_photon_tmp.as<float>()));
	RT_HIP_CHECK(hipMemcpyAsync(photons_out, ctx->d_photon_tmp.p, (size_t)ctx->num_photons * 36, hipMemcpyDeviceToHost, ctx->stream));
	RT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
	return RT_HIP_OK;
}

int rt_hip_photon_set_device(rt_hip_ctx* ctx, const float* d_photons, int64_t n)
{
	RT_CHECK(check_ctx(ctx));
	if (n < 0 || (n > 0 && !d_photons)) { rt_set_error("bad arguments"); return RT_HIP_ERR_ARG; }
	return rt_photon_build_grid(ctx, d_photons, n);
}

int rt_hip_photon_set(rt_hip_ctx* ctx, const float* photons, int64_t n)
{
	RT_CHECK(check_ctx(ctx));
	if (n < 0 || (n > 0 && !photons)) { rt_set_error("bad arguments"); return RT_HIP_ERR_ARG; }
	RT_CHECK(ctx->d_photon_tmp.reserve(std::max<size_t>((size_t)n, 1) * 36));
	if (n > 0) RT_HIP_CHECK(hipMemcpy(ctx->d_photon_tmp.p, photons, (size_t)n * 36, hipMemcpyHostToDevice));
	return rt_photon_build_grid(ctx, ctx->d_photon_tmp.as<float>(), n);
}

int rt_hip_radiance_estimate_device(rt_hip_ctx* ctx, const float* d_pos, const float* d_nrm, int npts, int count,
                                    float radius, float* d_rgb_out)
{
	RT_CHECK(check_ctx(ctx));
	if (npts < 0 || count <= 0 || !(radius > 0.0f) || (npts > 0 && (!d_pos || !d_nrm || !d_rgb_out))) {
		rt_set_error("bad arguments");
		return RT_HIP_ERR_ARG;
	}
	return rt_photon_gather(ctx, d_pos, d_nrm, npts, count, radius, d_rgb_out);
}

int rt_hip_radiance_estimate(rt_hip_ctx* ctx, const float* pos, const float* nrm, int npts, int count, float radius,
                             float* rgb_out)
{
	RT_CHECK(check_ctx(ctx));
	if (npts < 0 || (npts > 0 && (!pos || !nrm || !rgb_out))) { rt_set_error("bad arguments"); return RT_HIP_ERR_ARG; }
	if (npts == 0) return RT_HIP_OK;
	hipStream_t st = ctx->stream;
	RT_CHECK(ctx->d_org.reserve((size_t)npts * 12));
	RT_CHECK(ctx->d_dir.reserve((size_t)npts * 12));
	RT_CHECK(ctx->d_accum.reserve((size_t)npts * 12));
	RT_HIP_CHECK(hipMemcpyAsync(ctx->d_org.p, pos, (size_t)npts * 12, hipMemcpyHostToDevice, st));
	RT_HIP_CHECK(hipMemcpyAsync(ctx->d_dir.p, nrm, (size_t)npts * 12, hipMemcpyHostToDevice, st));
	RT_CHECK(rt_hip_radiance_estimate_device(ctx, ctx->d_org.as<float>(), ctx->d_dir.as<float>(), npts, count, radius,
	                                         ctx->d_accum.as<float>()));
	RT_HIP_CHECK(hipMemcpyAsync(rgb_out, ctx->d_accum.p, (size_t)npts * 12, hipMemcpyDeviceToHost, st));
	RT_HIP_CHECK(hipStreamSynchronize(st));
	return RT_HIP_OK;
}

int rt_hip_photon_lookup_device(rt_hip_ctx* ctx, const float* d_pos, int npts, float radius, float* d_rgb_out)
{
	RT_CHECK(check_ctx(ctx));
	if (npts < 0 || !(radius > 0.0f) || (npts > 0 && (!d_pos || !d_rgb_out))) { rt_set_error("bad arguments"); return RT_HIP_ERR_ARG; }
	return rt_photon_lookup(ctx, d_pos, npts, radius, d_rgb_out);
}

int rt_hip_photon_lookup(rt_hip_ctx* ctx, const float* pos, int npts, float radius, float* rgb_out)
{
	RT_CHECK(check_ctx(ctx));
	if (npts < 0 || (npts > 0 && (!pos || !rgb_out))) { rt_set_error("bad arguments"); return RT_HIP_ERR_ARG; }
	if (npts == 0) return RT_HIP_OK;
	hipStream_t st = ctx->stream;
	RT_CHECK(ctx->d_org.reserve((size_t)npts * 12));
	RT_CHECK(ctx->d_accum.reserve((size_t)npts * 12));
	RT_HIP_CHECK(hipMemcpyAsync(ctx->d_org.p, pos, (size_t)npts * 12, hipMemcpyHostToDevice, st));
	RT_CHECK(rt_hip_photon_lookup_device(ctx, ctx->d_org.as<float>(), npts, radius, ctx->d_accum.as<float>()));
	RT_HIP_CHECK(hipMemcpyAsync(rgb_out, ctx->d_accum.p, (size_t)npts * 12, hipMemcpyDeviceToHost, st));
	RT_HIP_CHECK(hipStreamSynchronize(st));
	return RT_HIP_OK;
}

// ---- the live (Gen-2) frame ---------------------------------------------------------------------------
int rt_hip_live_frame_device(rt_hip_ctx* ctx, const float* d_gbuf_pos_rgb, const float* d_gbuf_nrm_rgba, int width,
                             int height, int interlace, int phase, uint8_t* d_raypass_rgb8, uint8_t* d_rgba_out,
                             int* out_rays)
{
	RT_CHECK(check_ready(ctx));
	if (width <= 0 || height <= 0 || !d_gbuf_pos_rgb || !d_gbuf_nrm_rgba || !d_raypass_rgb8 || !d_rgba_out ||
	    (interlace > 1 && (phase < 0 || phase >= interlace))) {
		rt_set_error("bad arguments");
		return RT_HIP_ERR_ARG;
	}
	if (ctx->cfg.geometry != RT_HIP_TRIANGLES) { rt_set_error("the live frame needs triangle geometry"); return RT_HIP_ERR_STATE; }
	if ((int64_t)width * height > (int64_t)1 << 28) { rt_set_error("frame larger than 2^28 texels"); return RT_HIP_ERR_LIMIT; }
	int rays = 0;
	RT_CHECK(rt_live_frame(ctx, d_gbuf_pos_rgb, d_gbuf_nrm_rgba, width, height, interlace, phase, d_raypass_rgb8, d_rgba_out, &rays));
	if (out_rays) *out_rays = rays;
	return RT_HIP_OK;
}

int rt_hip_live_frame_scaled_device(rt_hip_ctx* ctx, const float* d_prepass_pos_rgb, const float* d_prepass_nrm_rgba, int prepass_width,
                                    int prepass_height, const float* d_gbuf_pos_rgb, const float* d_gbuf_nrm_rgba, int width, int height,
                                    float fov, int interlace, int phase, uint8_t* d_raypass_rgb8, uint8_t* d_rgba_out, int* out_rays)
{
	RT_CHECK(check_ready(ctx));
	if (width <= 0 || height <= 0 || prepass_width <= 0 || prepass_height <= 0 || !d_prepass_pos_rgb || !d_prepass_nrm_rgba ||
	    !d_gbuf_pos_rgb || !d_gbuf_nrm_rgba || !d_raypass_rgb8 || !d_rgba_out || (interlace > 1 && (phase < 0 || phase >= interlace))) {
		rt_set_error("bad arguments");
		return RT_HIP_ERR_ARG;
	}
	if (ctx->cfg.geometry != RT_HIP_TRIANGLES) { rt_set_error("the live frame needs triangle geometry"); return RT_HIP_ERR_STATE; }
	if ((int64_t)width * height > (int64_t)1 << 28 || (int64_t)prepass_width * prepass_height > (int64_t)1 << 28) {
		rt_set_error("frame larger than 2^28 texels");
		return RT_HIP_ERR_LIMIT;
	}
	int rays = 0;
	RT_CHECK(rt_live_frame_scaled(ctx, d_prepass_pos_rgb, d_prepass_nrm_rgba, prepass_width, prepass_height, d_gbuf_pos_rgb, d_gbuf_nrm_rgba,
	                              width, height, fov, interlace, phase, d_raypass_rgb8, d_rgba_out, &rays));
	if (out_rays) *out_rays = rays;
	return RT_HIP_OK;
}

// ---- packed tile exchange --------------------------------------------------------------------------
static int check_tiles_args(const void* frame, int w, int h, int tile, const int* d_tile_list, int ntiles, const void* packed)
{
	if (w <= 0 || h <= 0 || tile <= 0 || tile > 1024 || ntiles < 0 || (ntiles > 0 && (!frame || !d_tile_list || !packed))) {
		rt_set_error("bad arguments");
		return RT_HIP_ERR_ARG;
	}
	if ((int64_t)ntiles > (int64_t)((w + tile - 1) / tile) * ((h + tile - 1) / tile)) {
		rt_set_error("%d tiles listed, the frame has %d", ntiles, ((w + tile - 1) / tile) * ((h + tile - 1) / tile));
		return RT_HIP_ERR_ARG;
	}
	return RT_HIP_OK;
}

int rt_hip_pack_tiles_device(rt_hip_ctx* ctx, const uint8_t* d_frame_rgba, int width, int height, int tile,
                             const int* d_tile_list, int ntiles, uint8_t* d_packed)
{
	RT_CHECK(check_ctx(ctx));
	RT_CHECK(check_tiles_args(d_frame_rgba, width, height, tile, d_tile_list, ntiles, d_packed));
	return rt_launch_tiles_pack(ctx, true, (uint8_t*)d_frame_rgba, width, height, tile, d_tile_list, ntiles, d_packed);
}

int rt_hip_unpack_tiles_device(rt_hip_ctx* ctx, const uint8_t* d_packed, int width, int height, int tile,
                               const int* d_tile_list, int ntiles, uint8_t* d_frame_rgba)
{
	RT_CHECK(check_ctx(ctx));
	RT_CHECK(check_tiles_args(d_frame_rgba, width, height, tile, d_tile_list, ntiles, d_packed));
	return rt_launch_tiles_pack(ctx, false, d_frame_rgba, width, height, tile, d_tile_list, ntiles, (uint8_t*)d_packed);
}

// ---- live shading of hit records -----------------------------------------------------------------
static int check_groups(const rt_result_offset* offsets, int n_results, int n_groups, int num_materials)
{
	int prev = 0;
	for (int g = 0; g < n_groups; ++g) {
		if (offsets[g].end < prev || (int)offsets[g].material < 0 || (int)offsets[g].material >= num_materials) {
			rt_set_error("group %d: bad material id or non-monotonic end offset", g);
			return RT_HIP_ERR_ARG;
		}
		prev = offsets[g].end;
	}
	if (prev != n_results) { rt_set_error("groups cover %d records, n_results = %d", prev, n_results); return RT_HIP_ERR_ARG; }
	return RT_HIP_OK;
}

int rt_hip_shade_results_device(rt_hip_ctx* ctx, const rt_ray* d_results, const rt_result_offset* offsets, int n_results,
                                int n_groups, float* d_rgb_out, uint8_t* d_rgb8_out)
{
	RT_CHECK(check_ready(ctx));
	if (n_results < 0 || n_groups < 0 || (n_results > 0 && (!d_results || !offsets || !d_rgb_out || n_groups == 0))) {
		rt_set_error("bad arguments");
		return RT_HIP_ERR_ARG;
	}
	if (n_results == 0) return RT_HIP_OK;
	RT_CHECK(check_groups(offsets, n_results, n_groups, (int)ctx->materials.size()));
	RT_CHECK(ctx->d_offsets.reserve((size_t)n_groups * sizeof(rt_result_offset)));
	RT_HIP_CHECK(hipMemcpyAsync(ctx->d_offsets.p, offsets, (size_t)n_groups * sizeof(rt_result_offset), hipMemcpyHostToDevice,
	                            ctx->stream));
	RT_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // `offsets` is pageable host memory owned by the caller
	return rt_launch_shade_results(ctx, d_results, n_results, ctx->d_offsets.as<rt_result_offset>(), n_groups, d_rgb_out,
	                               d_rgb8_out);
}

int rt_hip_shade_results(rt_hip_ctx* ctx, const rt_ray* results, const rt_result_offset* offsets, int n_results,
                         int n_groups, float* rgb_out, uint8_t* rgb8_out)
{
	RT_CHECK(check_ready(ctx));
	if (n_results < 0 || (n_results > 0 && (!results || !offsets || !rgb_out))) { rt_set_error("bad arguments"); return RT_HIP_ERR_ARG; }
	if (n_results == 0) return RT_HIP_OK;
	hipStream_t st = ctx->stream;
	RT_CHECK(ctx->d_records.reserve((size_t)n_results * sizeof(rt_ray)));
	RT_CHECK(ctx->d_accum.reserve((size_t)n_results * 12));
	RT_CHECK(ctx->d_occ.reserve((size_t)n_results * 3));
	RT_HIP_CHECK(hipMemcpyAsync(ctx->d_records.p, results, (size_t)n_results * sizeof(rt_ray), hipMemcpyHostToDevice, st));
	RT_CHECK(rt_hip_shade_results_device(ctx, ctx->d_records.as<rt_ray>(), offsets, n_results, n_groups,
	                                     ctx->d_accum.as<float>(), rgb8_out ? ctx->d_occ.as<uint8_t>() : nullptr));
	RT_HIP_CHECK(hipMemcpyAsync(rgb_out, ctx->d_accum.p, (size_t)n_results * 12, hipMemcpyDeviceToHost, st));
	if (rgb8_out) RT_HIP_CHECK(hipMemcpyAsync(rgb8_out, ctx->d_occ.p, (size_t)n_results * 3, hipMemcpyDeviceToHost, st));
	RT_HIP_CHECK(hipStreamSynchronize(st));
	return RT_HIP_OK;
}

// ---- introspection -------------------------------------------------------------------------------
int rt_hip_get_stats(rt_hip_ctx* ctx, rt_hip_stats* out)
{
	RT_CHECK(check_ctx(ctx));
	if (!out) { rt_set_error("null output"); return RT_HIP_ERR_ARG; }
	RT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
	RT_CHECK(rt_collect_timing(ctx));
	*out = ctx->stats;
	return RT_HIP_OK;
}

int rt_hip_reset_stats(rt_hip_ctx* ctx)
{
	RT_CHECK(check_ctx(ctx));
	RT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
	RT_CHECK(rt_collect_timing(ctx));
	double build = ctx->stats.build_ms;
	const int last_update = ctx->stats.last_update;
	memset(&ctx->stats, 0, sizeof(ctx->stats));
	ctx->stats.last_update = last_update;
	ctx->stats.bvh_area_cost = ctx->use_bvh ? ctx->bvh_cost_now : 0.0;
	ctx->stats.bvh_area_cost_built = ctx->use_bvh ? ctx->bvh_cost_built : 0.0;
	if (ctx->d_census.p) RT_HIP_CHECK(hipMemset(ctx->d_census.p, 0, 128));
	ctx->stats.build_ms = build;
	ctx->stats.num_tris = ctx->num_tris;
	ctx->stats.num_nodes = ctx->num_nodes;
	ctx->stats.bvh_depth = ctx->bvh_depth;
	return RT_HIP_OK;
}

int rt_hip_sync(rt_hip_ctx* ctx)
{
	RT_CHECK(check_ctx(ctx));
	RT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
	// and the streams of the frame pipeline's other lanes, of its copies and of its collectives
	for (int k = 1; k < RT_MAX_PENDING; ++k)
		if (ctx->lane_stream[k]) RT_HIP_CHECK(hipStreamSynchronize(ctx->lane_stream[k]));
	for (hipStream_t st : {ctx->up_stream, ctx->down_stream, ctx->comm_stream})
		if (st) RT_HIP_CHECK(hipStreamSynchronize(st));
	return RT_HIP_OK;
}

void* rt_hip_stream(rt_hip_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int rt_hip_get_bvh(rt_hip_ctx* ctx, float* nodes_out, int64_t max_nodes, uint32_t* leaf_prims_out, int64_t max_tris)
{
	RT_CHECK(check_ready(ctx));
	if (!ctx->use_bvh) { rt_set_error("no LBVH was built for this scene"); return RT_HIP_ERR_STATE; }
	if (max_nodes < ctx->num_nodes || max_tris < ctx->num_tris) { rt_set_error("output arrays too small"); return RT_HIP_ERR_ARG; }
	RT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
	if (nodes_out) RT_HIP_CHECK(hipMemcpy(nodes_out, ctx->d_nodes.p, (size_t)ctx->num_nodes * sizeof(BvhNode), hipMemcpyDeviceToHost));
	if (leaf_prims_out) {
		std::vector<TriIsect> t((size_t)ctx->num_tris);
		RT_HIP_CHECK(hipMemcpy(t.data(), ctx->d_tri_isect.p, t.size() * sizeof(TriIsect), hipMemcpyDeviceToHost));
		for (size_t i = 0; i < t.size(); ++i) memcpy(&leaf_prims_out[i], &t[i].a.w, 4);
	}
	return RT_HIP_OK;
}

}  // extern "C"
