/* include/rt_hip.h -- C ABI of librt_hip.so, the MI355X (gfx950) ray-tracing back end.
 *
 * This is the drop-in boundary for the hot path of Kupoman/raytracer: everything the reference's
 * host code would bind to replace IAccel (raytracer/rt_iaccel.h:4-11) and
 * RayTracer::processRays (raytracer/rt_raytracer.h:58) with HIP kernels.  Plain pointers and
 * sizes only.  Every function returns 0 on success or a negative rt_hip_status; the message of
 * the last failure on the calling thread is available from rt_hip_last_error().
 *
 * Pointer conventions: parameters named `d_*` are DEVICE pointers (HBM resident, used by callers
 * that keep the G-buffer on the GPU); all other pointers are HOST pointers that are borrowed for
 * the duration of the call (the reference borrows the rasterizer's transfer buffers the same
 * way, rasterizer/ras_rasterizer.cpp:172-173,527-528).  Outputs returned through `const T**`
 * point into context-owned pinned host memory and stay valid until the next call on the same
 * context (the reference returns pointers into RayTracer members, rt_raytracer.cpp:536-539).
 *
 * A context is externally synchronised: one caller thread at a time (the reference's
 * processRays is non-re-entrant, rt_raytracer.cpp:27-48).  Independent contexts are independent.
 *
 * Results do not depend on the acceleration structure: every builder returns what the exhaustive
 * search over the triangles returns, bit for bit (exact ties in t go to the lower primitive id), for
 * ray origins at any distance; beyond ~1000 scene extents the triangle test's own float rounding
 * (the reference's Moeller-Trumbore, unchanged) is coarser than a triangle and decides such ties.
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_hip_ctx rt_hip_ctx;

enum rt_hip_status {
	RT_HIP_OK = 0,
	RT_HIP_ERR_ARG = -1,      /* bad argument (null pointer, negative count, unknown id) */
	RT_HIP_ERR_HIP = -2,      /* a HIP runtime call failed; see rt_hip_last_error() */
	RT_HIP_ERR_STATE = -3,    /* call order (e.g. trace before rt_hip_update) */
	RT_HIP_ERR_LIMIT = -4     /* a documented limit was exceeded */
};

/* Byte-identical to the reference's struct Ray (data/data.h:34-42), which is both the ray and
 * the hit record, and the GL vertex format of drawRayData (ras_rasterizer.cpp:586-589). */
typedef struct rt_ray {
	float origin[3];
	float direction[3];
	float position[3];
	float normal[3];
	float texcoord[2];
	float pad[2];
} rt_ray; /* 64 bytes */

/* Field-for-field the reference's struct Material (data/data.h:18-26); `texture` is the
 * Texture* slot, carried opaquely (textures are a "next" row). */
typedef struct rt_material {
	float diffuse_color[3];   /* 0..255 */
	float specular_color[3];  /* 0..255 */
	float shininess;
	float reflectivity;
	float alpha;              /* 1 - opacity (data/loader.cpp:104-105) */
	float ior;
	uint64_t texture;
} rt_material; /* 48 bytes */

/* The reference's struct Light (data/data.h:44-48). */
typedef struct rt_light {
	float position[3];
	float color[3];
	float energy;
} rt_light; /* 28 bytes */

/* The reference's ResultOffset = std::pair<Material*, int> (data/data.h:28) with the Material*
 * replaced by the dense material id; `end` is the CUMULATIVE end offset into the result array
 * (rt_raytracer.cpp:531-533). */
typedef struct rt_result_offset {
	uint32_t material;
	int32_t end;
} rt_result_offset;

/* Per-ray hit, the reference's DACRayResult (rt_raytracer.cpp:38-46) without the ray copy. */
typedef struct rt_hit {
	uint32_t prim;            /* global triangle id in mesh-then-face order; 0xFFFFFFFF = miss */
	float t;
	float u;
	float v;
} rt_hit; /* 16 bytes */

enum rt_hip_geometry { RT_HIP_TRIANGLES = 0, RT_HIP_SPHERES = 1 };
/* LBVH: Morton-split hierarchy (fastest build).  PLOC: agglomerative build over the same Morton
 * order, topology by surface area (fewer node visits per ray, a few ms more to build).  SAH: top-down
 * binned surface-area heuristic on ONE HOST THREAD (hundreds of ms; for scenes that do not change),
 * same node format and traversal. */
enum rt_hip_accel { RT_HIP_ACCEL_AUTO = 0, RT_HIP_ACCEL_BRUTE = 1, RT_HIP_ACCEL_LBVH = 2, RT_HIP_ACCEL_PLOC = 3,
                    RT_HIP_ACCEL_SAH = 4 };

/* Constants the reference hard-codes (file:line beside each). */
typedef struct rt_hip_config {
	int device;               /* HIP device ordinal */
	int geometry;             /* rt_hip_geometry: AccelArray-like triangles or AccelSpheres */
	int accel;                /* rt_hip_accel; AUTO = brute force below 512 triangles, else PLOC */
	int use_model_mat;        /* 1: world space, vertices x model_mat (processRays, rt_raytracer.cpp:466-475)
	                             0: object space (AccelArray::addMesh, rt_accel_array.cpp:17-32) */
	float eps_det;            /* reject det <  eps_det.  FLT_EPSILON (rt_raytracer.cpp:18,336) or the
	                             float equivalent of the double 1e-5 (rt_accel_array.cpp:10,94) */
	float eps_t;              /* reject t   <  eps_t  (rt_raytracer.cpp:356 / rt_accel_array.cpp:114) */
	float t_max;              /* accept t < t_max: +inf (rt_raytracer.cpp:451) or 1e6 (rt_accel_array.cpp:84) */
	int bounces;              /* 2      rt_raytracer.cpp:52 */
	int do_shadows;           /* 0      rt_raytracer.cpp:54 */
	int use_photon_map;       /* 0      rt_raytracer.cpp:56 (photon_map == NULL) */
	int photon_count;         /* 10000  rt_raytracer.cpp:57 */
	int photon_estimate;      /* 100    rt_raytracer.cpp:58 */
	float photon_radius;      /* 1.0    rt_raytracer.cpp:59 */
	float photon_max_dist;    /* 12     rt_photon_map.cpp:92 */
	float photon_absorb;      /* 0.6    rt_photon_map.cpp:45 */
	uint64_t seed;            /* counter-RNG seed of the photon pass */
	int profile;              /* 1: bracket the trace kernels with hipEvents (rt_hip_get_stats) */
	float shadow_t_max;       /* 0: literal reference shadow test -- ANY hit along the un-normalised
	                             light ray shadows, also beyond the light (rt_raytracer.cpp:113-117, whose
	                             distance test compares ray->position with itself).  > 0: a hit shadows
	                             only if t < shadow_t_max in units of |L| (1.0 = "before the light",
	                             the evident intent); runs that use it must say so */
	int census;               /* 1: trace kernels also count BVH node visits / triangle tests (slower) */
	int reserved;
} rt_hip_config;

/* What rt_hip_update found to do (rt_hip_stats.last_update). */
enum rt_hip_update_kind {
	RT_HIP_UPDATE_NONE = 0,    /* nothing changed since the last update */
	RT_HIP_UPDATE_TABLES = 1,  /* materials / lights / textures re-uploaded, geometry untouched */
	RT_HIP_UPDATE_REFIT = 2,   /* model matrices changed: soup rebuilt, boxes refitted on the kept BVH topology */
	RT_HIP_UPDATE_REBUILD = 3  /* meshes added, first update, rt_hip_invalidate, or the refit quality check failed: built from scratch */
};

/* Launch-shape and algorithm knobs that are not part of the reference's semantics (results are the same for
 * every setting; tests flip them to reach otherwise rare code paths).  rt_hip_create fills them with the defaults
 * below, overridden by the environment variable named beside each (read once, at rt_hip_create) -- see
 * INTEGRATION.md.  rt_hip_set_tuning takes effect at the next rt_hip_update where a build is involved. */
typedef struct rt_hip_tuning {
	int lds_stack_depth;      /* 16   RT_HIP_LDS_DEPTH: entries of a lane's traversal stack kept in LDS (1..32); the rest in HBM */
	int xcd_chunk;            /* 64   RT_HIP_XCD_CHUNK: workgroups per XCD and group in the XCD-aware block order; 0 = hardware order */
	int trace_oversub;        /* 4    RT_HIP_OVERSUB: grid cap of the trace kernels in units of the resident threads (1..16) */
	int auto_accel;           /* RT_HIP_ACCEL_PLOC  RT_HIP_AUTO_PLOC=0 -> LBVH: what RT_HIP_ACCEL_AUTO builds from 512 triangles on */
	int ploc_radius;          /* 8    RT_HIP_PLOC_RADIUS: PLOC nearest-neighbour search radius (1..64) */
	int frame_sync;           /* 0    RT_HIP_FRAME_SYNC: 1 = frames are rendered level by level with counts read back */
	uint32_t gather_heavy;    /* 32 RT_HIP_GATHER_HEAVY: a shading point with more photons in its cube takes the wave-per-point gather */
	int extra_lds_bytes;      /* 0    RT_HIP_EXTRA_LDS: extra LDS per trace workgroup (occupancy experiments) */
	int max_bvh_depth;        /* 95   deepest BVH the traversal stack takes (<= 95); a deeper tree is not used: brute force serves the scene */
	float refit_max_growth;   /* 2.0  RT_HIP_REFIT_GROWTH: refit is kept while (sum of node areas / root area) stays below this multiple
	                                 of its value at the last build; 0 = never refit, always rebuild */
	int gather_fine;          /* 1    RT_HIP_GATHER_FINE: the wave-per-point gather walks the fine photon grid (cell = radius / 8) and reads
	                                 only the boundary shell in its first pass; 0 = the coarse grid, every photon of the cube read */
	int gather_xcd_chunk;     /* 1024 RT_HIP_GATHER_XCD_CHUNK: consecutive queued shading points per XCD in the wave-per-point gather (their
	                                 neighbourhoods overlap, so the XCD's L2 serves most of their photon reads); 0 = hardware order */
	int gather_fine_cells;    /* 8    RT_HIP_GATHER_FINE_CELLS: fine photon grid cells per gather radius (2..32); read when a photon set is installed */
	uint32_t gather_medium;   /* 4096 RT_HIP_GATHER_MEDIUM: first pass of the wave-per-point gather: a point whose coarse cube holds at most this many
	                                 photons reads them all from the coarse grid; above it the fine grid's cells inside the sphere are counted from
	                                 the cell table and only the boundary shell is read; 0 = always the shell walk */
	int shadow_prefix;        /* 12   RT_HIP_SHADOW_PREFIX: the descent that the 64 shadow rays of a wave share (from the root to the neighbourhood of
	                                 their origins) is made once per (group of 64 frame nodes, light) and the rays start below it; the value is the
	                                 most sibling subtrees such a start may put on the rays' stacks (<= 30); 0 = every ray descends from the root */
	int shadow_prefix_steps;  /* 24   RT_HIP_SHADOW_PREFIX_STEPS: nodes such a shared descent may visit (one lane walks it, the kernel that makes
	                                 them is as long as its longest walk) */
	int shadow_prefix_min_rays; /* -1  RT_HIP_SHADOW_PREFIX_MIN_RAYS: level-0 shadow rays (nodes x lights) from which the shared descent is used; -1 = six
	                                 times the device's resident threads (below that the shadow launches are as long as their slowest wave and the
	                                 walks cost more than they save); 0 = always */
	int frame_streams;        /* 1    RT_HIP_FRAME_STREAMS: HIP streams the frame pipeline (rt_hip_frame_enqueue*) deals its frames to, each with level
	                                 arrays of its own (1..4).  1: enqueued frames run one after the other.  More: up to that many frames run side by
	                                 side, so the tail of one frame's launches and the gaps between them are filled by another frame's waves -- what a
	                                 job that cares for frames per second rather than for the latency of one frame wants, and what makes a frame that is
	                                 split over several GPUs (short launches, each as long as its slowest wave) scale */
	int gather_pair;          /* 100  RT_HIP_GATHER_PAIR: first pass of the wave-per-point gather for TWO neighbouring queue entries at once when
	                           *      they lie within this many percent of a fine cell of their midpoint: one walk of the grid, one flattening of
	                           *      the runs and one load per photon serve both (the distance arithmetic in packed FP32); 0 = every point alone.
	                           *      Same counts and max_dist2 bit for bit (first pass 3.17 -> 2.69 ms on the C5 frame's level 0). */
	int gather_sort;          /* 1    RT_HIP_GATHER_SORT: the wave-per-point gather's queue sorted by the points' positions (Morton order of their
	                           *      fine-grid cells) before the wave kernels: neighbours in the queue are then neighbours in space whatever order
	                           *      the caller's points come in (pairs form, each part of the map is read through one XCD's L2).  1: except for
	                           *      level 0 of a frame (8 x 8 tiles already), fewer than 16 k points or a map of fewer than 2 M photons (where
	                           *      the sort costs more than it saves); 2: always; 0: never.  Results do not depend on it. */
} rt_hip_tuning;

/* Kernel-time accounting for bench.py.  Times are only filled when config.profile != 0 (hipEvent
 * pairs around every trace launch on the context's stream); the census sums only when
 * config.census != 0 (the trace kernels then also count node visits and triangle tests). */
typedef struct rt_hip_stats {
	double closest_ms;        /* summed device time of closest-hit trace launches */
	int64_t closest_launches;
	int64_t rays_closest;     /* closest-hit rays issued */
	double any_ms;            /* summed device time of any-hit (shadow) trace launches */
	int64_t any_launches;
	int64_t rays_any;         /* any-hit rays issued */
	double build_ms;          /* device time of the last rt_hip_update that touched the geometry (soup + build, or soup + refit) */
	int64_t num_tris;
	int64_t num_nodes;
	int32_t bvh_depth;
	int32_t last_update;      /* what the last rt_hip_update did: rt_hip_update_kind */
	double frame_ms;          /* summed device time of whole frames (render / process entries) */
	int64_t frames;
	uint64_t census_nodes_closest, census_tris_closest, census_rays_closest;
	uint64_t census_nodes_any, census_tris_any, census_rays_any;
	double gather_ms;         /* summed device time of the photon gather (radiance_estimate) launches */
	int64_t gather_launches;
	int64_t gather_points;    /* shading points handed to the gather */
	uint64_t census_gather_visits; /* census: photons in the grid cells the points' mandatory first pass reads */
	double bvh_area_cost;       /* sum of the BVH nodes' surface areas / the root's: the tree in use (after a refit: the refitted boxes) */
	double bvh_area_cost_built; /* the same when the topology was built; a refit is kept while cost <= tuning.refit_max_growth x this */
} rt_hip_stats;

/* ---- life cycle --------------------------------------------------------------------------- */
/* processRays semantics (world space, FLT_EPSILON, t_max = +inf, bounces 2, no shadows). */
int rt_hip_default_config(rt_hip_config* cfg);
/* AccelArray semantics (object space, double 1e-5 thresholds, t_max = 1e6). */
int rt_hip_array_config(rt_hip_config* cfg);
int rt_hip_create(const rt_hip_config* cfg, rt_hip_ctx** out);      /* RayTracer::RayTracer / AccelArray::AccelArray */
int rt_hip_destroy(rt_hip_ctx* ctx);
const char* rt_hip_last_error(void);
int rt_hip_set_config(rt_hip_ctx* ctx, const rt_hip_config* cfg);   /* shading/photon fields only */
int rt_hip_get_tuning(rt_hip_ctx* ctx, rt_hip_tuning* out);
int rt_hip_set_tuning(rt_hip_ctx* ctx, const rt_hip_tuning* tuning);

/* ---- scene -------------------------------------------------------------------------------- */
/* Scene::materials / Scene::lights (data/scene.h:8-10).  Material ids are indices into this table. */
int rt_hip_set_materials(rt_hip_ctx* ctx, const rt_material* materials, int n);
int rt_hip_set_lights(rt_hip_ctx* ctx, const rt_light* lights, int n);
/* Material::texture = new Texture(path) (data/data.h:25, data/loader.cpp:110-121) with the decoded
 * image handed in: RGBA8 (the order Texture::Texture converts to, data/data.cpp:19-26), row 0 = the
 * bottom scanline (FreeImage_GetBits order, also what glTexImage2D receives, ras_rasterizer.cpp:132).
 * The image is copied.  rgba == NULL removes the texture.  rt_hip_set_materials drops all textures.
 * A textured material's diffuse colour becomes Texture::lookup(u, v) (data/data.cpp:35-57, 0..255)
 * in shade() (rt_raytracer.cpp:93-95) and texture2D(GL_NEAREST, repeat) / 255 in the live shading
 * (mesh_fs.cpp:29-30).  Untextured scenes take none of these paths. */
int rt_hip_set_texture(rt_hip_ctx* ctx, int material_id, const uint8_t* rgba, int width, int height);
/* IAccel::addMesh / RayTracer::addMesh (rt_iaccel.h:7, rt_raytracer.cpp:69).  Geometry is copied.
 * model_mat16 is column-major (Eigen default).  position3 is Mesh::position (the sphere centre
 * for RT_HIP_SPHERES, rt_accel_spheres.cpp:27-31).  Returns the mesh id (>= 0). */
int rt_hip_add_mesh(rt_hip_ctx* ctx, const float* verts, const float* normals, const float* texcoords,
                    int nverts, const uint32_t* faces, int nfaces, const float* model_mat16,
                    int material_id, const float* position3);
/* AccelSpheres::addSphere (rt_accel_spheres.cpp:14-26); the colour comes from libc
 * rand()%256 under srand(420) exactly as the reference draws it.  SIDE EFFECT, as in the reference
 * (rt_accel_spheres.cpp:11,24): the first sphere of a context calls srand(420) and every sphere draws three
 * numbers from the PROCESS-GLOBAL libc rand() stream, which perturbs any other user of rand() in the process
 * (the reference's own PhotonMap::generate included). */
int rt_hip_add_sphere(rt_hip_ctx* ctx, float x, float y, float z, float radius);
int rt_hip_set_model_mat(rt_hip_ctx* ctx, int mesh_id, const float* model_mat16);
/* The same for meshes [first_mesh, first_mesh + n) in one call: n x 16 floats (what a frame loop does once per
 * frame, rt_raytracer.cpp:465-467).  Matrices equal to the ones in use change nothing. */
int rt_hip_set_model_mats(rt_hip_ctx* ctx, int first_mesh, int n, const float* model_mats16);
/* IAccel::update (rt_iaccel.h:8): bring the device scene up to date.  Must be called after the scene changes
 * and before tracing.  It does the least that the changes since the last update need (rt_hip_update_kind):
 * nothing; the material / light / texture tables only; for changed model matrices (rt_hip_set_model_mat
 * with a matrix that differs from the mesh's current one -- an equal matrix changes nothing) the world-space
 * triangle soup of rt_raytracer.cpp:457-503 plus a bottom-up REFIT of the boxes on the kept BVH topology; a
 * build from scratch when meshes were added, after rt_hip_invalidate, or when the refitted tree's area sum
 * has grown past tuning.refit_max_growth.  Results never depend on which of these ran. */
int rt_hip_update(rt_hip_ctx* ctx);
/* The next rt_hip_update rebuilds everything from scratch. */
int rt_hip_invalidate(rt_hip_ctx* ctx);

/* ---- IAccel batch entries ------------------------------------------------------------------ */
/* IAccel::intersect for n rays (rt_iaccel.h:10; rt_accel_array.cpp:75-137 / rt_accel_spheres.cpp:58-104).
 * rays[i].origin/direction are read; on a hit position/normal/texcoord are written, on a miss the
 * ray is untouched.  hit_out[i] = 1/0; material_out[i] = material id (sphere index for spheres)
 * or 0xFFFFFFFF; hits_out (optional) receives prim/t/u/v. */
int rt_hip_intersect(rt_hip_ctx* ctx, rt_ray* rays, int n, uint8_t* hit_out, uint32_t* material_out,
                     rt_hit* hits_out);
/* IAccel::occlude for n rays (rt_iaccel.h:9; rt_accel_array.cpp:45-73: t is never tested;
 * rt_accel_spheres.cpp:38-56: infinite line). */
int rt_hip_occlude(rt_hip_ctx* ctx, const rt_ray* rays, int n, uint8_t* occluded_out);
/* Diffuse colours of the sphere materials (the reference hands out Material* per sphere). */
int rt_hip_sphere_colors(rt_hip_ctx* ctx, float* rgb_out, int max_spheres);

/* ---- RayTracer::processRays (rt_raytracer.h:58, rt_raytracer.cpp:428-540) ------------------ */
/* One reflection ray per G-buffer sample, closest hit, hit records grouped by material with
 * cumulative end offsets; hits only; ascending input index inside a group.  Groups are ordered
 * by ascending material id (the reference orders by Material* address, which no caller can
 * rely on).  `camera` is unused by the reference (:447) and has no parameter here. */
int rt_hip_process_gbuffer(rt_hip_ctx* ctx, const float* positions_xyz, const float* normals_xyz, int count,
                           const rt_ray** results, const rt_result_offset** result_offsets,
                           int* out_result_count, int* out_material_count);
/* Per-input-ray hits of the last rt_hip_process_gbuffer (the reference exposes the same data as
 * the global dac_results, rt_raytracer.cpp:48). */
int rt_hip_last_hits(rt_hip_ctx* ctx, rt_hit* hits_out, float* directions_out, int count);
/* Same, G-buffer samples already in HBM and records left in HBM (d_results: count x 64 B,
 * d_offsets: num_materials entries).  Counts are returned on the host. */
int rt_hip_process_gbuffer_device(rt_hip_ctx* ctx, const float* d_positions_xyz, const float* d_normals_xyz,
                                  int count, rt_ray* d_results, rt_result_offset* d_offsets,
                                  int* out_result_count, int* out_material_count);

/* ---- raw batch tracing on device buffers --------------------------------------------------- */
/* d_org / d_dir: n x 3 floats.  Closest hit with the context's eps / t_max. */
int rt_hip_trace_closest_device(rt_hip_ctx* ctx, const float* d_org, const float* d_dir, int n, rt_hit* d_hits);
/* Any hit with intersect's acceptance tests (shadow rays, rt_raytracer.cpp:110-118). */
int rt_hip_trace_any_device(rt_hip_ctx* ctx, const float* d_org, const float* d_dir, int n, uint8_t* d_occluded);

/* ---- G-buffer hand-off (rasterizer -> tracer) ---------------------------------------------- */
/* Rasterizer::getRayTraceData's selection (ras_rasterizer.cpp:509-536,556-561): keep texel i when
 * (interlace <= 1 or i % interlace == phase) and alpha > 0; normals decoded as c*2-1.  Stable.
 * nrm_rgba: n x 4, pos_rgb: n x 3 (the readback layout).  Outputs are n x 3; returns the count. */
int rt_hip_gbuffer_select_device(rt_hip_ctx* ctx, const float* d_nrm_rgba, const float* d_pos_rgb, int n,
                                 int interlace, int phase, float* d_positions_out, float* d_normals_out,
                                 int* d_index_out, int* out_count);
/* The same selection for callers that cannot allocate device memory (tight serving loops, harnesses that
 * forbid hipMalloc, code under HIP graph capture).  It has NO counterpart in the reference.
 * Same arguments, same outputs bit for bit and the same stable order as rt_hip_gbuffer_select_device,
 * but it calls no hipMalloc / hipFree: its only scratch (one count per 2048-texel tile, 33 KB) is a
 * fixed file-scope device array that each GPU holds from the first rt_hip_create on it until the
 * process ends.  n beyond RT_HIP_NOALLOC_MAX_TEXELS fails with RT_HIP_ERR_LIMIT.  The array is shared
 * by every context on a GPU: calls that overlap on one GPU (two contexts, two threads) race.
 * Where the count goes: d_count_out (device word of the caller, may be null) receives it from the scan
 * kernel; out_count (host, may be null) additionally through a 4-byte device-to-host copy and a stream
 * synchronisation.  With out_count == NULL the call is three kernel launches on the context's stream and
 * nothing else -- no copy, no wait -- which is the form a capturing stream takes; with out_count != NULL
 * it is not legal under capture (the copy and the wait are not).  At least one of the two must be given. */
#define RT_HIP_NOALLOC_MAX_TEXELS (1 << 24)
int rt_hip_gbuffer_select_noalloc_device(rt_hip_ctx* ctx, const float* d_nrm_rgba, const float* d_pos_rgb, int n,
                                         int interlace, int phase, float* d_positions_out, float* d_normals_out,
                                         int* d_index_out, int* d_count_out, int* out_count);
/* Software G-buffer producer: primary rays with Camera::getScreenRays
 * geometry (data/camera.cpp:19-44), closest hit, prepass_fs encodings (UNORM16 normal, FP16
 * position; prepass_fs.cpp:11-13, ras_rasterizer.cpp:293,301); alpha = material id + 1, 0 = empty. */
int rt_hip_render_gbuffer_device(rt_hip_ctx* ctx, float fov, int width, int height, float* d_nrm_rgba,
                                 float* d_pos_rgb);
/* The same plus the interpolated texture coordinate of the visible surface (width*height x 2; the
 * rasterizer's outCoord varying, mesh_vs/mesh_fs.cpp:29), (0,0) where nothing is visible. */
int rt_hip_render_gbuffer_uv_device(rt_hip_ctx* ctx, float fov, int width, int height, float* d_nrm_rgba,
                                    float* d_pos_rgb, float* d_uv);
/* Texture coordinates of the G-buffer texels for the frame calls that follow (render_frame,
 * render_tiles*, live_frame): d_uv[f] = width*height x 2 floats of frame f of a batch, in device
 * memory that stays valid until replaced.  nframes = 0 clears: level-0 lookups then use (0,0), a
 * Ray record's zeroed texcoord.  Only read when a material of the scene is textured.  The G-buffer of
 * the reference carries no texcoords (Ray::texcoord is never set by getRayTraceData,
 * ras_rasterizer.cpp:509-536); the forward pass gets them from the rasterizer's interpolation. */
int rt_hip_set_frame_texcoords_device(rt_hip_ctx* ctx, const float* const* d_uv, int nframes);

/* ---- full frame: G-buffer in, RGBA8 out ---------------------------------------------------- */
/* Gen-1 shade() (rt_raytracer.cpp:75-160) evaluated at every G-buffer texel with alpha > 0
 * (material id = alpha - 1), secondary rays (shadow, reflection, refraction) traced on the GPU;
 * RGBA8 written as renderScene does (rt_raytracer.cpp:175-185).  Frame semantics: DESIGN.md.
 * Host entry: uploads with hipMemcpyAsync, downloads the image.
 * Limits (RT_HIP_ERR_LIMIT beyond them): 32 lights, 8 bounces, 2^26 texels per frame.  Two drivers
 * render the same image: the default one sizes every level of the ray tree by its bound and never
 * returns to the host inside a frame; frames with a photon map, with more than 30 lights, or too
 * large to be sized by bounds are rendered level by level with counts read back (so is any frame
 * when tuning.frame_sync = 1). */
int rt_hip_render_frame(rt_hip_ctx* ctx, const float* gbuf_pos_rgb, const float* gbuf_nrm_rgba, int width,
                        int height, uint8_t* rgba_out);
int rt_hip_render_frame_device(rt_hip_ctx* ctx, const float* d_gbuf_pos_rgb, const float* d_gbuf_nrm_rgba,
                               int width, int height, uint8_t* d_rgba_out);
/* Multi-GPU tile split: shade only texels whose tile (tile x tile pixels, row-rotated round
 * robin) belongs to `rank` of `nranks`; other texels of d_rgba_out are left untouched. */
int rt_hip_render_tiles_device(rt_hip_ctx* ctx, const float* d_gbuf_pos_rgb, const float* d_gbuf_nrm_rgba,
                               int width, int height, int tile, int rank, int nranks, uint8_t* d_rgba_out);
/* The same for a batch of frames of one size (at most 16) in ONE wavefront: the three arguments are
 * host arrays of nframes device pointers.  A rank of an N-GPU job shades its tiles of N frames per
 * step; batching them keeps every launch frame-sized instead of 1/N-sized.  Per frame the result is
 * what rt_hip_render_tiles_device writes, bit for bit; rt_hip_last_frame_rays() covers the batch. */
int rt_hip_render_tiles_batch_device(rt_hip_ctx* ctx, const float* const* d_gbuf_pos_rgb,
                                     const float* const* d_gbuf_nrm_rgba, int nframes, int width, int height,
                                     int tile, int rank, int nranks, uint8_t* const* d_rgba_out);
/* The same batch with the output already PACKED for the exchange: d_packed_out[f] is the rank's chunk of
 * frame f in the layout rt_hip_pack_tiles_device produces (own tiles in row-major tile order, each
 * tile x tile pixels row-major; padding pixels of edge tiles are never written, so clear the chunks
 * once).  Needs 1 < nranks <= 16 and tile % 8 == 0.  rt_hip_unpack_chunks_device is the receiving
 * side: d_chunks holds nranks chunks of chunk_bytes each, chunk r = rank r's tiles of this frame;
 * they are scattered into the full-size image (enqueued: rt_hip_sync() before reading it elsewhere). */
int rt_hip_render_tiles_packed_device(rt_hip_ctx* ctx, const float* const* d_gbuf_pos_rgb,
                                      const float* const* d_gbuf_nrm_rgba, int nframes, int width, int height,
                                      int tile, int rank, int nranks, uint8_t* const* d_packed_out);
/* One frame's packed share queued without waiting -- rt_hip_render_tiles_packed_device for one frame in the frame pipeline
 * (rt_hip_frame_enqueue_device's rules: at most max(2, tuning.frame_streams) unfinished, rt_hip_frame_finish completes the
 * oldest), for a host that runs its own exchange of the chunks.  No counterpart in the reference. */
int rt_hip_tiles_packed_enqueue_device(rt_hip_ctx* ctx, const float* d_gbuf_pos_rgb, const float* d_gbuf_nrm_rgba, int width, int height,
                                       int tile, int rank, int nranks, uint8_t* d_packed_out);
/* ... and the packed shares of several frames (1..16, one size) as ONE pipeline entry: a single pass of launches shades all of
 * them, rt_hip_frame_finish completes the entry and returns its rays (no counterpart in the reference; for a host that runs
 * its own exchange and wants launches larger than one rank's share of one frame). */
int rt_hip_tiles_packed_batch_enqueue_device(rt_hip_ctx* ctx, const float* const* d_gbuf_pos_rgb, const float* const* d_gbuf_nrm_rgba,
                                             int nframes, int width, int height, int tile, int rank, int nranks, uint8_t* const* d_packed_out);
int rt_hip_unpack_chunks_device(rt_hip_ctx* ctx, const uint8_t* d_chunks, int64_t chunk_bytes, int width, int height,
                                int tile, int nranks, uint8_t* d_frame_rgba);
/* Tile gather without moving empty pixels (collective #2): pack the tiles listed in
 * d_tile_list (ids ty * tiles_x + tx, ntiles of them; raytracer_amd/tiles.py: owned_tiles) from a
 * full-size RGBA8 frame into one contiguous buffer (tile after tile, tile x tile pixels row-major,
 * zero padding where an edge tile leaves the frame), and the inverse.  The packed buffers are what
 * the ranks exchange (all-to-all: rank r's tiles of frame f go to rank f).  No counterpart in the
 * reference, which has one GPU context and no tiles.  Both are enqueued on the context's stream:
 * call rt_hip_sync() before another stream (a collective) touches the buffers. */
int rt_hip_pack_tiles_device(rt_hip_ctx* ctx, const uint8_t* d_frame_rgba, int width, int height, int tile,
                             const int* d_tile_list, int ntiles, uint8_t* d_packed);
int rt_hip_unpack_tiles_device(rt_hip_ctx* ctx, const uint8_t* d_packed, int width, int height, int tile,
                               const int* d_tile_list, int ntiles, uint8_t* d_frame_rgba);
/* The same frame as rt_hip_render_frame_device, in a pipeline: rt_hip_frame_enqueue_device queues all of the frame's kernels
 * and returns WITHOUT waiting; rt_hip_frame_finish waits for the OLDEST enqueued frame (its image is then complete) and
 * returns its secondary-ray count.  At most two frames (max(2, tuning.frame_streams), up to four) may be enqueued and unfinished; with one enqueued ahead the host's
 * per-frame work runs under the previous frame's kernels.  Frames are executed in order on the context's stream; the other
 * frame entries (rt_hip_render_frame*, rt_hip_render_tiles*) refuse to run while frames are enqueued (RT_HIP_ERR_STATE).  Frames the sync-free driver
 * does not take (photon map, > 30 lights, too large) are rendered inside the enqueue call.  No counterpart in the reference,
 * whose frame loop is strictly sequential (data/scene.cpp:40-58). */
int rt_hip_frame_enqueue_device(rt_hip_ctx* ctx, const float* d_gbuf_pos_rgb, const float* d_gbuf_nrm_rgba, int width, int height,
                                uint8_t* d_rgba_out);
int rt_hip_frame_finish(rt_hip_ctx* ctx, int64_t* out_rays);
/* The pipeline fed from HOST memory (what rt_hip_render_frame does, rt_raytracer.h:58's caller hands host arrays): the G-buffer
 * is uploaded on a copy stream of its own, the frame waits for it, the image goes back on a third stream, so with one frame
 * enqueued ahead the PCIe legs of frame k + 1 and k - 1 run under the kernels of frame k.  The three buffers must stay valid
 * and untouched until rt_hip_frame_finish has returned for this frame, and should be pinned (hipHostMalloc /
 * hipHostRegister): with pageable memory the copies are staged by the runtime and the overlap is lost, not the result.
 * Frames enqueued through either entry finish in order through the same rt_hip_frame_finish. */
int rt_hip_frame_enqueue(rt_hip_ctx* ctx, const float* gbuf_pos_rgb, const float* gbuf_nrm_rgba, int width, int height,
                         uint8_t* rgba_out);
/* The same with the G-buffer in the formats the rasterizer's FBO holds (ras_rasterizer.cpp:293,301): normals RGBA16 UNORM
 * (4 x uint16 per texel, the alpha word = material id + 1 as an integer, 0 = empty), positions RGB16F (3 x IEEE half per
 * texel) -- what the reference widens to floats only in glReadPixels (ras_rasterizer.cpp:497,543).  14 instead of 28 bytes
 * per texel cross PCIe; c / 65535 and half -> float are applied on the device and are exact, so the frame is bit for bit
 * the one rt_hip_frame_enqueue renders from the widened planes.  Same buffer lifetime rules, same rt_hip_frame_finish. */
int rt_hip_frame_enqueue_g16(rt_hip_ctx* ctx, const uint16_t* gbuf_nrm_rgba16, const uint16_t* gbuf_pos_rgb16f, int width, int height,
                             uint8_t* rgba_out);
/* Secondary rays issued by the last frame call (shadow + reflection + refraction). */
int64_t rt_hip_last_frame_rays(rt_hip_ctx* ctx);

/* ---- the live (Gen-2) frame: SURVEY.md 8f row 2 ---------------------------------------------- */
/* Scene::draw (data/scene.cpp:40-56) from the G-buffer on, entirely on the device:
 *   getRayTraceData's selection (texel i with alpha > 0 and, for interlace > 1, i % interlace == phase;
 *     ras_rasterizer.cpp:509-536),
 *   processRays' ray generation and closest hit (rt_raytracer.cpp:444-455, 507),
 *   drawRayData (ras_rasterizer.cpp:619-669): each hit shaded by mesh_fs at the hit point with the
 *     hit's material and written as RGB8 into the raypass image at the texel its ray came from;
 *     misses write nothing; the image persists across calls so the phases accumulate, and the
 *     caller fills it with 255 once (the reference clears it to white on the first pass only, :632-634),
 *   drawMeshes (:237-281, mesh_fs.cpp:46-47): colour = refl * raypass + (1 - refl) * direct at every
 *     covered texel with the visible surface's material (alpha - 1), background = clear colour 0.3.
 * Deliberate differences from the GL path: prepass scale 1 (raypass and frame have one size); the
 * point lands on the texel it came from (no FP16 reprojection jitter); the forward pass shades the
 * G-buffer's position/normal rather than re-rasterised vertex attributes; each group is drawn with
 * its own count (the reference passes the cumulative offset as the count, :660); untextured.
 * d_raypass_rgb8: width * height * 3 bytes; d_rgba_out: width * height * 4; *out_rays = rays traced. */
int rt_hip_live_frame_device(rt_hip_ctx* ctx, const float* d_gbuf_pos_rgb, const float* d_gbuf_nrm_rgba,
                             int width, int height, int interlace, int phase, uint8_t* d_raypass_rgb8,
                             uint8_t* d_rgba_out, int* out_rays);

/* The same with the reference's two sizes (Rasterizer::prepass_resolution = 0.75, ras_rasterizer.cpp:39,167-168): the rays
 * start from a prepass_width x prepass_height G-buffer (select, ray generation and closest hit as above); each hit is drawn
 * as drawRayData draws its point (ras_rasterizer.cpp:619-669, viewport = the prepass size): at the projection of Ray.origin
 * through projectionFromCamera's matrix (:140-153, n = 1, f = 100, the frame's aspect ratio; mesh_vs.cpp:17), GL clip test,
 * the texel that contains the window position -- so an origin that went through FP16 can land beside the texel it came
 * from, and records that share a texel overwrite each other (GL: the last one drawn; here: one of them); the frame is
 * composed at width x height from the full-size G-buffer (the stand-in for drawMeshes' rasterisation), the reflection
 * term read as texture2D(raypassBuffer, gl_FragCoord.st / frame_size) reads it (mesh_fs.cpp:46): bilinear (GL_LINEAR,
 * :600-601) with the default GL_REPEAT wrap.  With the two sizes equal the composite is rt_hip_live_frame_device's.
 * fov as rt_hip_render_gbuffer_device takes it: tan(fov) scales the image plane (data/camera.cpp:19-44), and the matrix here is
 * built from that same scale so that a point projects back onto the texel of a G-buffer made with it.  (The reference's matrix
 * takes tan() of Camera::getFOV(), which already is a tangent -- ras_rasterizer.cpp:143, camera.cpp:41-44,60-63 -- so its raster
 * camera is tan(tan(fov)); it is consistent with itself because its G-buffer is rasterised through that same matrix.  A caller
 * whose G-buffer comes from that rasterizer passes atan(tan(tan(fov))).)
 * d_raypass_rgb8: prepass_width * prepass_height * 3 bytes, persistent across calls, filled with 255 once by the caller.
 * Parity unpinned (GL is not available; filtering precision and the order of overlapping points are GL's own). */
int rt_hip_live_frame_scaled_device(rt_hip_ctx* ctx, const float* d_prepass_pos_rgb, const float* d_prepass_nrm_rgba, int prepass_width,
                                    int prepass_height, const float* d_gbuf_pos_rgb, const float* d_gbuf_nrm_rgba, int width, int height,
                                    float fov, int interlace, int phase, uint8_t* d_raypass_rgb8, uint8_t* d_rgba_out, int* out_rays);

/* ---- live shading of the hit records (the raypass) ------------------------------------------ */
/* rasterizer/shaders/mesh_fs.cpp:26-49 evaluated at each hit record the way Rasterizer::drawRayData
 * (ras_rasterizer.cpp:619-669) draws it: one point per record with its group's material, the
 * reflection term 0 (the raypass target is itself the reflection source), at most LIGHT_COUNT = 8
 * lights (mesh_fs.cpp:5), light and material colours / 255 (ras_rasterizer.cpp:184-188,232).
 * results / offsets are what rt_hip_process_gbuffer returned (groups = (material, cumulative end)).
 * rgb_out: n_results x 3 floats; rgb8_out (may be NULL): the RGB8 target's UNORM8 value.
 * `offsets` is a host pointer in both variants (it is tiny). */
int rt_hip_shade_results(rt_hip_ctx* ctx, const rt_ray* results, const rt_result_offset* offsets, int n_results,
                         int n_groups, float* rgb_out, uint8_t* rgb8_out);
int rt_hip_shade_results_device(rt_hip_ctx* ctx, const rt_ray* d_results, const rt_result_offset* offsets,
                                int n_results, int n_groups, float* d_rgb_out, uint8_t* d_rgb8_out);

/* ---- photon map (raytracer/rt_photon_map.h:13-16) ------------------------------------------ */
/* PhotonMap::generate (rt_photon_map.cpp:86-124) for photon indices [first, first+count) of the
 * global sequence lights x photons_per_light, with the counter RNG keyed by (seed, index) so the
 * result does not depend on how indices are sharded.  Stored photons (9 floats: position,
 * energy, direction) are appended to d_photons_out in index order; *out_stored = how many.
 * A photon is stored at most once, so capacity (in photons) must be >= count. */
int rt_hip_photon_emit_device(rt_hip_ctx* ctx, int photons_per_light, int64_t first, int64_t count,
                              float* d_photons_out, int64_t capacity, int64_t* out_stored);
/* PhotonMap::generate in one call for host-only callers (no device buffer to manage): photons_per_light photons
 * per light of the context's light table, emitted with the counter RNG under config.seed, kept in the context and
 * installed as its photon map (grid built).  *out_stored = photons in the map.  The single-GPU form of
 * rt_hip_photon_emit_device + rt_hip_photon_set_device. */
int rt_hip_photon_generate(rt_hip_ctx* ctx, int photons_per_light, int64_t* out_stored);
/* The installed photon set back on the host, in photon-index order (9 floats each: position, energy, direction),
 * i.e. PhotonMap::photons; *out_n = how many (capacity in photons must cover them; capacity 0 only asks for the count). */
int rt_hip_photon_get(rt_hip_ctx* ctx, float* photons_out, int64_t capacity, int64_t* out_n);
/* Install the (all-gathered) photon set and build the hashed grid (cell = radius). */
int rt_hip_photon_set_device(rt_hip_ctx* ctx, const float* d_photons, int64_t n);
int rt_hip_photon_set(rt_hip_ctx* ctx, const float* photons, int64_t n);
/* PhotonMap::radiance_estimate (rt_photon_map.cpp:149-208) at npts points. */
int rt_hip_radiance_estimate_device(rt_hip_ctx* ctx, const float* d_pos, const float* d_nrm, int npts,
                                    int count, float radius, float* d_rgb_out);
int rt_hip_radiance_estimate(rt_hip_ctx* ctx, const float* pos, const float* nrm, int npts, int count,
                             float radius, float* rgb_out);

/* PhotonMap::lookup (rt_photon_map.cpp:126-142): the summed energy of the photons with |p - pos|^2 < radius^2
 * at npts points (the reference has no caller for it).  The sum runs in grid order, not photon-index order:
 * equal to the reference's to float rounding of the sum. */
int rt_hip_photon_lookup_device(rt_hip_ctx* ctx, const float* d_pos, int npts, float radius, float* d_rgb_out);
int rt_hip_photon_lookup(rt_hip_ctx* ctx, const float* pos, int npts, float radius, float* rgb_out);

/* ---- multi-GPU: one process and one context per GPU, RCCL over xGMI ---------------------- */
/* No counterpart in the reference (one thread, one GL context).  The frame is split by interleaved tile x tile
 * screen tiles (tile k of a row-rotated round robin belongs to rank (tx + ty) % nranks), scene and BVH replicated
 * on every GPU; the photon pass is sharded by photon index.  RCCL is used for exactly two exchange steps: the
 * all-gather of the finished tiles and the all-gather of the photon shards.  librccl is bound at run time
 * (dlopen), only when these entries are used.
 * Bootstrap: rank 0 calls rt_hip_comm_unique_id and hands the RT_HIP_COMM_ID_BYTES bytes to the other ranks by the
 * caller's own means (MPI, a file, a socket, torch.distributed ...); then EVERY rank calls rt_hip_comm_init
 * (collective: it returns when all ranks have called it).  At most 16 ranks. */
#define RT_HIP_COMM_ID_BYTES 128
int rt_hip_comm_unique_id(void* id128);
int rt_hip_comm_init(rt_hip_ctx* ctx, const void* id128, int rank, int nranks);
int rt_hip_comm_destroy(rt_hip_ctx* ctx);   /* also done by rt_hip_destroy */
/* One frame over all ranks (strong scaling; collective): this rank shades its tiles of the frame, the packed tiles
 * are all-gathered, and every rank ends up with the whole RGBA8 image in d_rgba_out -- bit for bit the image
 * rt_hip_render_frame_device renders on one GPU.  tile: a multiple of 8 (64 is what bench.py uses).
 * rt_hip_last_frame_rays() = this rank's rays. */
int rt_hip_render_frame_multi_device(rt_hip_ctx* ctx, const float* d_gbuf_pos_rgb, const float* d_gbuf_nrm_rgba, int width,
                                     int height, int tile, uint8_t* d_rgba_out);
/* The same, queued without waiting: the frame pipeline's form of it (no counterpart in the reference, whose Scene::draw renders
 * one frame at a time, data/scene.cpp:40-58).  rt_hip_frame_finish completes the oldest enqueued frame: on return its assembled
 * image is in d_rgba_out on this rank.  At most max(2, tuning.frame_streams) frames may be unfinished; with frame_streams > 1 the
 * ranks' shares of consecutive frames run side by side on each GPU, and the all-gather of frame k travels under the kernels of
 * frame k + 1 -- the form to use when frames per second matter: an eighth of a 1080p frame is three launches of ~90 us that
 * leave most of the GPU idle, three frames in flight fill it (0.33 -> 0.15 ms per share measured on one GPU).  Every rank must
 * enqueue and finish the same frames in the same order.  Frames too large for the sync-free driver are refused (RT_HIP_ERR_LIMIT):
 * rt_hip_render_frame_multi_device takes those. */
int rt_hip_frame_multi_enqueue_device(rt_hip_ctx* ctx, const float* d_gbuf_pos_rgb, const float* d_gbuf_nrm_rgba, int width,
                                      int height, int tile, uint8_t* d_rgba_out);
/* ... and with SEVERAL frames per pipeline entry (1..16, all of one size; no counterpart in the reference either): one pass of
 * launches shades this rank's tiles of all nframes frames, ONE all-gather carries all their chunks, one unpack per frame; on
 * return of rt_hip_frame_finish (which completes the whole entry and returns its rays) image f is in d_rgba_out[f].  With
 * the frame split N ways, N / 2 frames per entry keep every launch as large as half a frame's: measured on one GPU, the
 * slowest rank's share of a 1080p frame at N = 8 takes 0.36 ms alone, 0.19 with three frames in flight, 0.14 / 0.12 in
 * entries of 4 / 8 frames.  The price is latency: an entry's frames are finished together.  Refused (RT_HIP_ERR_LIMIT)
 * when the largest rank's share of the entry could overflow the sync-free driver's level bounds (2^23 nodes). */
int rt_hip_frames_multi_enqueue_device(rt_hip_ctx* ctx, const float* const* d_gbuf_pos_rgb, const float* const* d_gbuf_nrm_rgba,
                                       int nframes, int width, int height, int tile, uint8_t* const* d_rgba_out);
/* PhotonMap::generate over all ranks (collective): each rank emits its contiguous range of the global photon index
 * sequence, the shards are all-gathered in rank (= index) order, and every rank installs the full set -- the
 * photon array rt_hip_photon_generate makes on one GPU, bit for bit.  *out_stored = photons in the map. */
int rt_hip_photon_generate_multi(rt_hip_ctx* ctx, int photons_per_light, int64_t* out_stored);

/* ---- introspection ------------------------------------------------------------------------- */
int rt_hip_get_stats(rt_hip_ctx* ctx, rt_hip_stats* out);
int rt_hip_reset_stats(rt_hip_ctx* ctx);
int rt_hip_sync(rt_hip_ctx* ctx);
/* The hipStream_t all of the context's work is enqueued on (as an opaque pointer). */
void* rt_hip_stream(rt_hip_ctx* ctx);
/* Copies of the LBVH for validation and for the roofline's nodes/ray census:
 * nodes: num_nodes x 16 floats (64 B each, layout in DESIGN.md); leaf_prims: num_tris ids. */
int rt_hip_get_bvh(rt_hip_ctx* ctx, float* nodes_out, int64_t max_nodes, uint32_t* leaf_prims_out,
                   int64_t max_tris);
/* Per-ray traversal census on the device: nodes visited and triangles tested per ray. */
int rt_hip_trace_census_device(rt_hip_ctx* ctx, const float* d_org, const float* d_dir, int n, int any_hit,
                               uint32_t* d_nodes_visited, uint32_t* d_tris_tested);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
