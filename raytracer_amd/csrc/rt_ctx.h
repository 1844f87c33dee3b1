// raytracer_amd/csrc/rt_ctx.h -- host-side context and the kernel launch interface between the
// translation units of librt_hip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <functional>
#include <string>
#include <vector>

#include "../../include/rt_hip.h"
#include "rt_dev.h"

// ---- error plumbing ---------------------------------------------------------------------------
void rt_set_error(const char* fmt, ...);
#define RT_HIP_CHECK(expr)                                                                        \
	do {                                                                                          \
		hipError_t e__ = (expr);                                                                  \
		if (e__ != hipSuccess) {                                                                  \
			rt_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
			return RT_HIP_ERR_HIP;                                                                \
		}                                                                                         \
	} while (0)
#define RT_CHECK(expr)                    \
	do {                                  \
		int s__ = (expr);                 \
		if (s__ != RT_HIP_OK) return s__; \
	} while (0)

// Growable device buffer (never shrinks; the frame loop allocates nothing in steady state).
struct DevBuf {
	void* p = nullptr;
	size_t cap = 0;
	int reserve(size_t bytes);
	void release();
	template <typename T> T* as() const { return (T*)p; }
};

struct HostMesh {
	std::vector<float> verts, normals, texcoords;
	std::vector<uint32_t> faces;
	float model[16];
	int material;
	float position[3];
	uint32_t first_tri;  // global id of face 0
	size_t vert_offset;  // into the concatenated device vertex arrays
};

struct HostTexture {  // Material::texture, data/data.h:7-16
	int w = 0, h = 0;
	std::vector<uint32_t> texels;  // RGBA8 words, row 0 = bottom scanline
};

struct HostSphere {
	float c[3];
	float r;
	float diffuse[3];
};

// Device scene views handed to kernels.
struct SceneView {
	const BvhNode* nodes;     // float nodes (build output, introspection)
	const QNode* qnodes;      // 32-byte quantised nodes: what the traversal reads
	QFrame qframe;            // their grid
	const TriIsect* tris;     // leaf order
	const TriPos* tri_pos;    // original id order
	const TriAttr* tri_attr;  // original id order
	const SphereRec* spheres;
	const MaterialRec* materials;
	const LightRec* lights;
	const TexRec* tex;        // per material, or nullptr: no material of the scene is textured
	const uint32_t* texels;   // RGBA8 pool of all textures
	int num_tris;
	int num_spheres;
	int num_materials;
	int num_lights;
	int use_bvh;      // 0: brute force over `tris`
	int stack_depth;  // required traversal stack entries (BVH depth + 1, + 1 for the sentinel at the bottom)
	int lds_depth;    // entries [0, lds_depth) of a lane's stack live in LDS, the rest in `spill`
	int* spill;       // [stack_depth - lds_depth][spill_stride] ints in HBM (nullptr if not needed)
	int spill_stride; // = maximum number of threads of a trace launch
	int xcd_chunk;    // XCD-aware block renumbering (rt_trace_dev.h); 0 = hardware order
	int extra_lds;    // host side only: tuning.extra_lds_bytes
	TraceParams tp;
};

// A batch of frames shaded in one wavefront (same size, same scene): the G-buffers in, the RGBA8
// images out.  Level-0 node j belongs to frame j / per_frame; a level-0 node's `parent` is
// frame * (w * h) + pixel.  With a tile list (multi-GPU split, tile % 8 == 0) only the rank's own
// tiles become nodes, so every kernel of the frame shrinks with 1 / nranks.
constexpr int RT_MAX_BATCH = 16;
struct FrameSet {
	const float* pos[RT_MAX_BATCH];
	const float* nrm[RT_MAX_BATCH];
	uint8_t* rgba[RT_MAX_BATCH];
	const float* uv[RT_MAX_BATCH];  // level-0 texture coordinates (w*h x 2) or nullptr: (0,0); read only when the scene has textures
	int nframes;
	int per_frame;         // level-0 nodes per frame
	const int* tile_list;  // owned tile ids (ty * tiles_x + tx), or nullptr: every 8x8 block of the frame
	int ntiles;
	int packed;            // 1: rgba[f] is the PACKED chunk of the rank's tiles (tile after tile, row-major inside), not a frame
};

// The frame pipeline: at most RT_MAX_PENDING frames enqueued and not finished, dealt to tuning.frame_streams lanes.  A lane =
// a HIP stream + the level arrays of one frame (lane 0: the context's stream and the context's own arrays, which the
// synchronous driver uses too).  Frames of one lane run in order; frames of different lanes side by side.
// Four: the box runs four hardware queues per process (GPU_MAX_HW_QUEUES), and a fourth lane already shares one with the others:
// an eighth of the 1080p frame took 0.310 / 0.200 / 0.157 / 0.184 ms per frame with 1 / 2 / 3 / 4 lanes (scripts/share_lanes_probe.py).
constexpr int RT_MAX_PENDING = 4;
struct FrameBufs {
	DevBuf nodes_a, nodes_b, gate, node_inc, node_uv, rays_q, ray_parent, records, hits, occ, child_slot, fcount, group_boxes, prefix;
};
struct FrameLane {  // what frame_async_enqueue works on
	hipStream_t stream;
	DevBuf *nodes_a, *nodes_b, *gate, *node_inc, *node_uv, *rays_q, *ray_parent, *records, *hits, *occ, *child_slot, *fcount, *group_boxes, *prefix;
	const int** fcount_clean;  // the lane's d_fcount[0, *fcount_clean_n) is known to be zero
	int* fcount_clean_n;
	int index;
};

// A frame whose kernels are queued and whose counters the host has not read yet (sync-free frame driver).
struct PendingFrame {
	bool used = false;
	bool complete = false;       // rendered synchronously at enqueue (the sync-free driver declined it): nothing to wait for
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	int* h_fc = nullptr;         // pinned: the frame's counters land here
	size_t trace_mark = 0;       // trace_events index behind this frame's launches
	int n0 = 0, bounces = 0, nframes = 0, num_lights = 0;
	bool shadows = false;
	bool can_overflow = false;   // some level's bound is the level cap: the frame may have to be redone with host-sized levels
	hipStream_t stream = nullptr; // the lane's stream the frame was queued on
	bool comm = false;            // rt_hip_frame_multi_enqueue_device: ev_down = all-gather and unpack done (on the communicator's stream)
	int64_t rays = 0;
	// what a repeat on the synchronous driver needs (a level overflowed its bound)
	FrameSet fs;
	int w = 0, h = 0, tile = 0, rank = 0, nranks = 1, packed = 0;
	// rt_hip_frame_enqueue (host buffers): the copies run on streams of their own, ordered against the frame by events
	bool host = false;
	hipEvent_t ev_up = nullptr, ev_done = nullptr, ev_down = nullptr;
	uint8_t* h_rgba = nullptr;
	size_t img_bytes = 0;
};

struct rt_hip_ctx {
	rt_hip_config cfg;
	rt_hip_tuning tuning;
	hipStream_t stream = nullptr;
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	std::vector<std::pair<hipEvent_t, hipEvent_t>> trace_events;  // pending (profile mode)
	std::vector<int> trace_event_class;                           // 0 closest, 1 any
	size_t trace_events_used = 0;
	size_t trace_events_collected = 0;                            // [collected, used) are still to be folded into the stats
	PendingFrame pending[RT_MAX_PENDING];                         // the frame pipeline (rt_hip_frame_enqueue_device / _finish): a ring
	int pending_head = 0, pending_count = 0;                      // oldest slot, frames enqueued and not finished
	bool frame_redone = false;                                    // the last rt_frame_finish rendered its frame again (level overflow)
	void* h_pipe = nullptr;                                       // pinned: RT_MAX_PENDING x 512 B of frame counters
	int64_t frame_seq = 0;                                        // frames enqueued so far: frame j runs on lane j % lanes
	hipStream_t lane_stream[RT_MAX_PENDING] = {};                 // streams of lanes 1 ..; created on first use
	FrameBufs lane_bufs[RT_MAX_PENDING];                          // level arrays of lanes 1 .. (lane 0: the d_* members below)
	const int* lane_fcount_clean[RT_MAX_PENDING] = {};
	int lane_fcount_clean_n[RT_MAX_PENDING] = {};
	hipEvent_t ev_order = nullptr;                                // orders a lane's stream behind the context's stream at enqueue
	hipStream_t comm_stream = nullptr;                            // rt_hip_frame_multi_enqueue_device: all-gathers and unpacks, in frame order
	DevBuf d_mg_send_slot[RT_MAX_PENDING], d_mg_recv_slot[RT_MAX_PENDING];
	hipStream_t up_stream = nullptr, down_stream = nullptr;       // rt_hip_frame_enqueue: G-buffer up / image down beside the frame's stream
	DevBuf d_pipe_pos[RT_MAX_PENDING], d_pipe_nrm[RT_MAX_PENDING], d_pipe_img[RT_MAX_PENDING];  // its device staging, one set per pending slot
	DevBuf d_pipe_g16[RT_MAX_PENDING];                                         // rt_hip_frame_enqueue_g16: the 14 B / texel planes as they crossed PCIe
	DevBuf d_census, d_spill, d_tile_list;
	int tile_key[5] = {0, 0, 0, 0, 0};                            // (w, h, tile, rank, nranks) of d_tile_list
	int tile_count = 0;
	bool warned_fallback = false, warned_depth = false;
	DevBuf d_all_tiles;                                           // every rank's tile list, concatenated (unpack of a received step)
	int all_tiles_key[4] = {0, 0, 0, 0};                          // (w, h, tile, nranks)
	int all_tiles_offset[RT_MAX_BATCH + 1] = {};
	int resident_threads = 0;   // CUs x threads per CU
	int max_trace_threads = 0;  // grid cap of the trace kernels = tuning.trace_oversub x resident_threads

	std::vector<HostMesh> meshes;
	std::vector<HostSphere> spheres;
	std::vector<rt_material> materials;
	std::vector<HostTexture> textures;                            // by material id (shorter than `materials` is fine)
	bool any_texture = false;
	const float* frame_uv[RT_MAX_BATCH] = {};                     // rt_hip_set_frame_texcoords_device
	int frame_uv_count = 0;
	std::vector<rt_light> lights;
	// What changed since the last rt_hip_update.  scene_dirty = any of them (tracing is refused until the update).
	bool scene_dirty = true;
	bool dirty_topology = true;   // meshes / spheres added, first update, rt_hip_invalidate: everything from scratch
	bool dirty_tables = true;     // materials, lights, textures
	bool dirty_xf = false;        // a model matrix changed (and the context applies model matrices)
	bool sphere_rng_seeded = false;
	float bvh_cost_now = 0.0f;    // the same for the boxes in use (differs from bvh_cost_built after a refit)
	float bvh_cost_built = 0.0f;  // sum of node half-areas / root half-area at the last build (refit quality check)
	DevBuf d_leaf_order;          // leaf slot -> global triangle id of the BVH in use (kept for refits; d_order is scratch)
	DevBuf d_level_nodes;         // internal node ids grouped by depth (the refit's launch lists), then the per-node depth
	std::vector<int> level_offsets;  // level d = d_level_nodes[level_offsets[d], level_offsets[d + 1]); empty: not computed yet

	// geometry in HBM
	DevBuf d_verts, d_normals, d_texcoords, d_faces, d_mesh_of_tri, d_mesh_xf;  // inputs of the soup kernel
	DevBuf d_tri_pos, d_tri_attr, d_tri_isect, d_tri_box, d_nodes, d_qnodes, d_spheres, d_materials, d_lights, d_tex, d_texels;
	DevBuf d_morton, d_morton_tmp, d_order, d_order_tmp, d_hist, d_flags, d_parent_leaf, d_bounds, d_misc, d_ploc;
	int num_tris = 0, num_nodes = 0, bvh_depth = 0, use_bvh = 0;
	float bounds[6];
	QFrame qframe;

	// per-call scratch in HBM
	DevBuf d_org, d_dir, d_hits, d_occ, d_in_a, d_in_b, d_records, d_offsets, d_keys, d_rank, d_counts;
	DevBuf d_nodes_a, d_nodes_b, d_rays_q, d_ray_parent, d_accum, d_qcount, d_gate, d_node_uv, d_node_inc, d_fcount, d_child_slot, d_group_boxes, d_prefix;
	const int* fcount_clean = nullptr;  // sync-free frame driver: d_fcount[0, fcount_clean_n) is known to be zero
	int fcount_clean_n = 0;
	// pinned host staging
	void* h_records = nullptr; size_t h_records_cap = 0;
	void* h_offsets = nullptr; size_t h_offsets_cap = 0;
	void* h_small = nullptr;   // 4 KiB pinned scratch for counters

	// photon map
	DevBuf d_photons, d_photon_cell, d_photon_sorted, d_cell_start, d_photon_tmp, d_heavy, d_photon_emit;
	int64_t num_photons = 0;
	float photon_cell = 1.0f;
	float photon_grid_min[3];
	int photon_grid_dim[3];
	// the same photons in a second, finer grid (cell = radius / 8): what the wave-per-point gather reads
	DevBuf d_photons_f, d_photon_sorted_f, d_cell_start_f;
	float photon_cell_f = 1.0f;
	float photon_grid_min_f[3];
	int photon_grid_dim_f[3];

	// multi-GPU (k_comm.hip)
	void* comm = nullptr;         // ncclComm_t
	int comm_rank = 0, comm_nranks = 1;
	int mg_key[4] = {0, 0, 0, 0}; // (w, h, tile, nranks) the send chunk was cleared for
	DevBuf d_mg_send, d_mg_recv, d_mg_counts, d_mg_photons;

	// rt_render_frames: work to queue behind the frame's kernels BEFORE the host waits for its counters (the multi-GPU entry
	// puts its all-gather and unpack there: one host wait per frame instead of two); run again if the frame had to be redone
	std::function<int()> after_frame_queued;

	int last_count = 0;
	int64_t last_frame_rays = 0;
	rt_hip_stats stats;
};

SceneView rt_scene_view(const rt_hip_ctx* ctx);
void rt_default_tuning(rt_hip_tuning* t);   // defaults, then the RT_HIP_* environment overrides
void rt_clamp_tuning(rt_hip_ctx* ctx);      // ranges; also derives max_trace_threads
// Launch shape of the trace kernels: blocks of `block` threads, at most ctx->max_trace_threads in
// total (the spill stack has exactly that many columns).
int rt_trace_blocks(const rt_hip_ctx* ctx, int64_t work_items, int block);
size_t rt_trace_lds(const SceneView& sv, int block);
int rt_begin_trace_timing(rt_hip_ctx* ctx, hipStream_t st = nullptr);  // nullptr: ctx->stream
int rt_end_trace_timing(rt_hip_ctx* ctx, int cls, hipStream_t st = nullptr);
int rt_next_trace_events(rt_hip_ctx* ctx, int cls, hipEvent_t* a, hipEvent_t* b);  // for hipExtLaunchKernelGGL
unsigned long long* rt_census_ptr(rt_hip_ctx* ctx, int cls);  // nullptr unless cfg.census
int rt_collect_timing(rt_hip_ctx* ctx, size_t upto = (size_t)-1);  // fold the event pairs [collected, min(upto, used))
int rt_ensure_pinned(void** p, size_t* cap, size_t bytes);

// ---- k_geometry.hip ------------------------------------------------------------------------------
int rt_build_geometry(rt_hip_ctx* ctx);  // what the dirty flags ask for: tables, soup + refit, or soup + build; all on ctx->stream

// ---- k_lbvh.hip ----------------------------------------------------------------------------------
int rt_build_lbvh(rt_hip_ctx* ctx);
// Boxes of the kept topology from the current tri_box / d_isect_src (original order); *cost = area sum / root area.
int rt_refit_lbvh(rt_hip_ctx* ctx, const TriIsect* d_isect_src, float* cost);
int rt_reserve_spill(rt_hip_ctx* ctx);  // HBM part of the traversal stacks for the current depth / tuning

// ---- k_trace.hip ---------------------------------------------------------------------------------
// d_count != nullptr: n is the capacity, the ray count is read on the device (brute force ignores it: BVH only)
int rt_launch_closest(rt_hip_ctx* ctx, const float* d_org, const float* d_dir, int n, Hit* d_hits,
                      const int* d_count = nullptr);
int rt_launch_any(rt_hip_ctx* ctx, const float* d_org, const float* d_dir, int n, uint8_t* d_occ);
int rt_launch_occlude_ref(rt_hip_ctx* ctx, const float* d_org, const float* d_dir, int n, uint8_t* d_occ);
int rt_launch_census(rt_hip_ctx* ctx, const float* d_org, const float* d_dir, int n, int any_hit, uint32_t* d_nv,
                     uint32_t* d_nt);
int rt_launch_sphere_intersect(rt_hip_ctx* ctx, const float* d_org, const float* d_dir, int n, int* d_index,
                               float* d_pos, float* d_nrm);
int rt_launch_sphere_occlude(rt_hip_ctx* ctx, const float* d_org, const float* d_dir, int n, uint8_t* d_occ);

// ---- k_frame.hip ---------------------------------------------------------------------------------
int rt_launch_split_rays(rt_hip_ctx* ctx, const rt_ray* d_rays, int n, float* d_org, float* d_dir);
int rt_launch_reflect_raygen(rt_hip_ctx* ctx, const float* d_pos, const float* d_nrm, int n, float* d_org,
                             float* d_dir);
int rt_launch_fill_records(rt_hip_ctx* ctx, const float* d_org, const float* d_dir, const Hit* d_hits, int n,
                           rt_ray* d_rays_inout, uint8_t* d_hit_flag, uint32_t* d_material);
int rt_group_by_material(rt_hip_ctx* ctx, const float* d_org, const float* d_dir, const Hit* d_hits, int n,
                         rt_ray* d_records, rt_result_offset* d_offsets, int* out_results, int* out_groups);
int rt_gbuffer_select(rt_hip_ctx* ctx, const float* d_nrm_rgba, const float* d_pos_rgb, int n, int interlace,
                      int phase, float* d_pos_out, float* d_nrm_out, int* d_index_out, int* out_count);
int rt_launch_shade_results(rt_hip_ctx* ctx, const rt_ray* d_rec, int n, const rt_result_offset* d_groups, int ngroups,
                            float* d_rgb, uint8_t* d_rgb8);
int rt_live_frame(rt_hip_ctx* ctx, const float* d_pos_rgb, const float* d_nrm_rgba, int w, int h, int interlace, int phase,
                  uint8_t* d_raypass, uint8_t* d_rgba, int* out_rays);
int rt_live_frame_scaled(rt_hip_ctx* ctx, const float* d_pre_pos, const float* d_pre_nrm, int pw, int ph, const float* d_pos_rgb,
                         const float* d_nrm_rgba, int w, int h, float fov, int interlace, int phase, uint8_t* d_raypass,
                         uint8_t* d_rgba, int* out_rays);
int rt_launch_tiles_pack(rt_hip_ctx* ctx, bool pack, uint8_t* d_frame, int w, int h, int tile, const int* d_tile_list,
                         int ntiles, uint8_t* d_packed);
int rt_render_gbuffer(rt_hip_ctx* ctx, float fov, int w, int h, float* d_nrm_rgba, float* d_pos_rgb, float* d_uv = nullptr);
int rt_render_frame(rt_hip_ctx* ctx, const float* d_pos_rgb, const float* d_nrm_rgba, int w, int h, int tile,
                    int rank, int nranks, uint8_t* d_rgba);
int rt_frame_enqueue(rt_hip_ctx* ctx, const float* d_pos_rgb, const float* d_nrm_rgba, int w, int h, uint8_t* d_rgba,
                     int tile = 64, int rank = 0, int nranks = 1, int packed = 0);
int rt_frames_enqueue(rt_hip_ctx* ctx, const float* const* d_pos_rgb, const float* const* d_nrm_rgba, int nframes, int w, int h,
                      uint8_t* const* d_rgba, int tile = 64, int rank = 0, int nranks = 1, int packed = 0);  // a batch = one pending entry
int rt_pipeline_slot(rt_hip_ctx* ctx, int* slot);  // the ring slot the next enqueued frame takes, or RT_HIP_ERR_STATE when the pipeline is full
FrameLane rt_frame_lane(rt_hip_ctx* ctx, int lane);
int rt_frame_finish(rt_hip_ctx* ctx, int64_t* out_rays);
int rt_launch_decode_g16(rt_hip_ctx* ctx, const uint16_t* d_nrm16, const uint16_t* d_pos16, int n, float* d_nrm_rgba, float* d_pos_rgb);
int rt_render_frames(rt_hip_ctx* ctx, const float* const* d_pos_rgb, const float* const* d_nrm_rgba, int nframes, int w,
                     int h, int tile, int rank, int nranks, uint8_t* const* d_rgba, int packed = 0);
int rt_unpack_chunks(rt_hip_ctx* ctx, const uint8_t* d_chunks, int64_t chunk_bytes, int w, int h, int tile, int nranks,
                     uint8_t* d_frame, hipStream_t st = nullptr, int64_t rank_stride_bytes = 0);  // nullptr: ctx->stream; 0: the chunks lie back to back

// ---- k_noalloc.hip -------------------------------------------------------------------------------
int rt_noalloc_setup(int device);  // once per device, that device current
int rt_gbuffer_select_noalloc(rt_hip_ctx* ctx, const float* d_nrm_rgba, const float* d_pos_rgb, int n, int interlace,
                              int phase, float* d_pos_out, float* d_nrm_out, int* d_index_out, int* d_count_out, int* out_count);

// ---- k_photon.hip --------------------------------------------------------------------------------
int rt_photon_emit(rt_hip_ctx* ctx, int per_light, int64_t first, int64_t count, float* d_out, int64_t capacity,
                   int64_t* out_stored);
int rt_photon_build_grid(rt_hip_ctx* ctx, const float* d_photons, int64_t n);
int rt_photon_gather(rt_hip_ctx* ctx, const float* d_pos, const float* d_nrm, int npts, int count, float radius,
                     float* d_rgb, bool coherent = false);  // coherent: consecutive points are neighbours in space already
int rt_photon_lookup(rt_hip_ctx* ctx, const float* d_pos, int npts, float radius, float* d_rgb);
int rt_photon_unpermute(rt_hip_ctx* ctx, float* d_out);  // the installed set in photon-index order, num_photons x 9 floats

// ---- shared device-scan / multisplit helpers (k_sort.hip) -----------------------------------------
// Stable counting-sort pass on an 8-bit digit: out_rank[i] = destination of element i.
int rt_multisplit256(rt_hip_ctx* ctx, const uint32_t* d_digit, int n, uint32_t* d_dest, uint32_t* d_bucket_totals);
int rt_radix_sort_pairs64(rt_hip_ctx* ctx, uint64_t* d_keys, uint64_t* d_keys_tmp, uint32_t* d_vals,
                          uint32_t* d_vals_tmp, int n, int key_bits);
