// raytracer_amd/csrc/k_frame.hip -- everything around the trace kernels on the frame path:
//   * reflection-ray generation from G-buffer samples       rt_raytracer.cpp:444-455
//   * hit-record construction and group-by-material          rt_raytracer.cpp:510-539
//   * G-buffer select / decode                               ras_rasterizer.cpp:509-536,556-561
//   * software G-buffer producer (primary rays)              data/camera.cpp:19-44, prepass_fs.cpp:11-13
//   * Gen-1 shade() as a level-synchronous wavefront         rt_raytracer.cpp:75-160
//     (shadow any-hit rays per (light, node), ballot-compacted reflection / refraction queues,
//     bottom-up colour combine in the reference's nesting order so the image is reproducible)
#include <hip/hip_fp16.h>

#include "rt_ctx.h"
#include "rt_trace_dev.h"

namespace {

constexpr int BLOCK = 256;
// one wave per workgroup for the kernels that traverse: the frame measured 1.145 ms at 64, 1.173 at 128, 1.193 at 256
// (a finished wave frees its slot and its LDS stack at once instead of waiting for its block mates)
constexpr int TRACE_BLOCK = 64;
inline int nblocks(int64_t n, int b) { return (int)((n + b - 1) / b); }

// Resolved hit: what IAccel::intersect leaves in the Ray plus the material.
struct __attribute__((aligned(16))) HitInfo {
	float pos[3];
	int material;  // -1 = miss
	float nrm[3];
	float pad;
};

// 32 bytes: what the shadow rays (read once per light) and the combine need.  The incident direction,
// which only the kernels that spawn reflection / refraction rays read, lives in an array of its own
// (3 floats per node), so the per-light streams carry a third less.
struct __attribute__((aligned(16))) NodeGeo {
	float V[3];
	int parent;    // level 0: frame * pixels + pixel (or -1); deeper: index of the parent node in the previous level | branch << 31
	float N[3];
	int material;  // -1: inactive, -2: no node (padding / another rank's texel)
};
static_assert(sizeof(NodeGeo) == 32, "node record");
__device__ __forceinline__ int node_parent(const NodeGeo& g) { return g.parent & 0x7fffffff; }
__device__ __forceinline__ int node_branch(const NodeGeo& g) { return (int)((unsigned int)g.parent >> 31); }  // 0 reflection child, 1 refraction child
// Sync-free frame driver: a node's shadow mask word also says which children it has (set by
// k_make_children), so that NodeShade needs no zero-initialisation: k_combine reads the colour a child
// returned only where that child exists.  Lights therefore use bits [0, 30).
constexpr unsigned int CHILD_REFL_BIT = 1u << 30, CHILD_REFR_BIT = 1u << 31;
constexpr int FRAME_ASYNC_MAX_LIGHTS = 30;

struct __attribute__((aligned(16))) NodeShade { // 48 B
	float F[3];    // shadow * light
	float pad0;
	float refc[3]; // colour returned by the reflection child (0 if none)
	float pad1;
	float refrc[3];
	float pad2;
};

// Block-level stream compaction: destination slot for every lane whose flag is set, with ONE
// global atomic per 1024-thread workgroup (a single word sustains only ~88 returning atomics/us
// on MI355X, so one atomic per wave made these kernels atomic-bound).  Every thread of the
// workgroup must call it.  Order inside a workgroup is lane order; across workgroups arrival order.
constexpr int CBLOCK = 1024;
__device__ __forceinline__ int block_compact_slot(bool flag, int* __restrict__ counter)
{
	__shared__ int s_wave[CBLOCK / 64];
	__shared__ int s_base;
	const uint64_t m = __ballot(flag);
	const int wave = (int)(threadIdx.x >> 6);
	if ((threadIdx.x & 63) == 0) s_wave[wave] = (int)__popcll(m);
	__syncthreads();
	if (threadIdx.x == 0) {
		int tot = 0;
		for (int w = 0; w < CBLOCK / 64; ++w) { int c = s_wave[w]; s_wave[w] = tot; tot += c; }
		s_base = tot ? atomicAdd(counter, tot) : 0;
	}
	__syncthreads();
	return s_base + s_wave[wave] + (int)__popcll(m & lanemask_lt());
}

// ---- ray generation ------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_reflect_raygen(const float* __restrict__ pos, const float* __restrict__ nrm,
                                                          int n, float* __restrict__ org, float* __restrict__ dir)
{
	int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n) return;
	f3 P = ld3(pos + 3 * (size_t)i), N = ld3(nrm + 3 * (size_t)i);
	// direction = P - 2 * P.dot(N) * N, normalised (rt_raytracer.cpp:445-448)
	f3 d = P - N * (2.0f * dot3(P, N));
	st3(dir + 3 * (size_t)i, normalized3(d));
	st3(org + 3 * (size_t)i, P);
}

__global__ __launch_bounds__(BLOCK) void k_split_rays(const rt_ray* __restrict__ rays, int n, float* __restrict__ org,
                                                      float* __restrict__ dir)
{
	int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n) return;
	const float* r = (const float*)(rays + i);
	for (int k = 0; k < 3; ++k) { org[3 * (size_t)i + k] = r[k]; dir[3 * (size_t)i + k] = r[3 + k]; }
}

// ---- hit resolution --------------------------------------------------------------------------
__device__ __forceinline__ void resolve_tri(const SceneView& sv, const Hit& h, HitInfo& hi, float* texcoord)
{
	if (h.prim == RT_MISS) { hi.material = -1; return; }
	const TriPos tp = sv.tri_pos[h.prim];
	const TriAttr ta = sv.tri_attr[h.prim];
	f3 p = bary3(h.u, h.v, mk3(tp.p0.x, tp.p0.y, tp.p0.z), mk3(tp.p1.x, tp.p1.y, tp.p1.z), mk3(tp.p2.x, tp.p2.y, tp.p2.z));
	f3 nn = bary3(h.u, h.v, ld3(ta.n0), ld3(ta.n1), ld3(ta.n2));
	nn = normalized3(nn);
	st3(hi.pos, p);
	st3(hi.nrm, nn);
	hi.material = __float_as_int(tp.p0.w);
	if (texcoord) {
		float w = (1.0f - h.u - h.v);
		texcoord[0] = (w * ta.t0[0] + h.u * ta.t1[0]) + h.v * ta.t2[0];
		texcoord[1] = (w * ta.t0[1] + h.u * ta.t1[1]) + h.v * ta.t2[1];
	}
}

// ---- textures ---------------------------------------------------------------------
// Texture::lookup (data/data.cpp:35-57): 0..255 per channel.  FreeImage_GetPixelColor fails for
// x == width / y == height (u or v exactly 1) and the reference then returns an uninitialised colour;
// defined here (and in the oracle) as the last texel.
__device__ __forceinline__ f3 tex_lookup_ref(const TexRec t, const uint32_t* __restrict__ texels, float u, float v)
{
	if (fabsf(u) > 1.0f) u = fmodf(u, 1.0f);
	if (fabsf(v) > 1.0f) v = fmodf(v, 1.0f);
	if (u < 0.0f) u = (float)((double)u + 1.0);
	if (v < 0.0f) v = (float)((double)v + 1.0);
	int x = (int)(u * (float)(unsigned)t.w);
	int y = (int)(v * (float)(unsigned)t.h);
	x = min(max(x, 0), t.w - 1);
	y = min(max(y, 0), t.h - 1);
	const uint32_t c = texels[t.offset + (size_t)y * t.w + x];
	return mk3((float)(c & 255u), (float)((c >> 8) & 255u), (float)((c >> 16) & 255u));
}

// texture2D on the GL_NEAREST texture with the default GL_REPEAT wrap (ras_rasterizer.cpp:130-135,
// mesh_fs.cpp:29): i = floor(frac(s) * w), UNORM8 -> c / 255.
__device__ __forceinline__ f3 tex_lookup_gl(const TexRec t, const uint32_t* __restrict__ texels, float u, float v)
{
	const float fu = u - floorf(u), fv = v - floorf(v);
	const int x = min(max((int)floorf(fu * (float)t.w), 0), t.w - 1);
	const int y = min(max((int)floorf(fv * (float)t.h), 0), t.h - 1);
	const uint32_t c = texels[t.offset + (size_t)y * t.w + x];
	return mk3((float)(c & 255u) / 255.0f, (float)((c >> 8) & 255u) / 255.0f, (float)((c >> 16) & 255u) / 255.0f);
}

// mesh_fs.cpp:29-30: albedo = material_textured ? texture2D(...).rgb : material_color (= diffuse / 255)
__device__ __forceinline__ f3 live_albedo(const SceneView& sv, int mat, const MaterialRec& m, float u, float v)
{
	if (sv.tex) {
		const TexRec t = sv.tex[mat];
		if (t.w > 0) return tex_lookup_gl(t, sv.texels, u, v);
	}
	return div3(ld3(m.diffuse), 255.0f);
}

// Count convention of the frame kernels: `n` is the item count, or -- when n_ptr != nullptr -- the
// capacity, and the count is min(n, *n_ptr) read on the device (the sync-free frame path never
// brings a count back to the host between launches).
__device__ __forceinline__ int dev_count(int n, const int* __restrict__ n_ptr)
{
	if (!n_ptr) return n;
	int m = *n_ptr;
	return m < n ? m : n;
}

__global__ __launch_bounds__(BLOCK) void k_resolve(SceneView sv, const Hit* __restrict__ hits, int n,
                                                   const int* __restrict__ n_ptr, HitInfo* __restrict__ out)
{
	n = dev_count(n, n_ptr);
	int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n) return;
	HitInfo hi;
	hi.pad = 0;
	resolve_tri(sv, hits[i], hi, nullptr);
	out[i] = hi;
}

// IAccel::intersect's write-back: hit -> position/normal/texcoord, miss -> untouched.
__global__ __launch_bounds__(BLOCK) void k_fill_records(SceneView sv, const Hit* __restrict__ hits, int n,
                                                        rt_ray* __restrict__ rays, uint8_t* __restrict__ flag,
                                                        uint32_t* __restrict__ material)
{
	int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n) return;
	HitInfo hi;
	float tc[2];
	resolve_tri(sv, hits[i], hi, tc);
	bool hit = hi.material >= 0;
	flag[i] = hit ? 1 : 0;
	material[i] = hit ? (uint32_t)hi.material : RT_MISS;
	if (hit) {
		float* r = (float*)(rays + i);
		for (int k = 0; k < 3; ++k) { r[6 + k] = hi.pos[k]; r[9 + k] = hi.nrm[k]; }
		r[12] = tc[0]; r[13] = tc[1];
	}
}

__global__ __launch_bounds__(BLOCK) void k_fill_sphere_records(const int* __restrict__ index,
                                                               const float* __restrict__ pos,
                                                               const float* __restrict__ nrm, int n,
                                                               rt_ray* __restrict__ rays, uint8_t* __restrict__ flag,
                                                               uint32_t* __restrict__ material)
{
	int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n) return;
	bool hit = index[i] >= 0;
	flag[i] = hit ? 1 : 0;
	material[i] = hit ? (uint32_t)index[i] : RT_MISS;
	if (hit) {
		float* r = (float*)(rays + i);
		for (int k = 0; k < 3; ++k) { r[6 + k] = pos[3 * (size_t)i + k]; r[9 + k] = nrm[3 * (size_t)i + k]; }
	}
}

// ---- group by material -----------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_material_digit(SceneView sv, const Hit* __restrict__ hits, int n, int shift,
                                                          uint32_t* __restrict__ digit)
{
	int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n) return;
	uint32_t p = hits[i].prim;
	uint32_t key = (p == RT_MISS) ? 0xFFFFu : (uint32_t)__float_as_int(sv.tri_pos[p].p0.w);
	digit[i] = (key >> shift) & 255u;
}

__global__ __launch_bounds__(128) void k_zero_ints(int* __restrict__ p, int n)
{
	for (int k = threadIdx.x; k < n; k += 128) p[k] = 0;
}

__global__ __launch_bounds__(BLOCK) void k_iota(uint32_t* p, int n)
{
	int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i < n) p[i] = (uint32_t)i;
}

// perm_out[dest[i]] = perm_in[i]; digits for the next pass are gathered through the permutation
__global__ __launch_bounds__(BLOCK) void k_apply_dest(const uint32_t* __restrict__ perm_in,
                                                      const uint32_t* __restrict__ dest, int n,
                                                      uint32_t* __restrict__ perm_out)
{
	int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i < n) perm_out[dest[i]] = perm_in[i];
}

__global__ __launch_bounds__(BLOCK) void k_material_digit_perm(SceneView sv, const Hit* __restrict__ hits,
                                                               const uint32_t* __restrict__ perm, int n, int shift,
                                                               uint32_t* __restrict__ digit)
{
	int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n) return;
	uint32_t p = hits[perm[i]].prim;
	uint32_t key = (p == RT_MISS) ? 0xFFFFu : (uint32_t)__float_as_int(sv.tri_pos[p].p0.w);
	digit[i] = (key >> shift) & 255u;
}

// records[slot] for slot < nres: the ray that landed in slot `slot` is perm[slot]
__global__ __launch_bounds__(BLOCK) void k_emit_records(SceneView sv, const float* __restrict__ org,
                                                        const float* __restrict__ dir, const Hit* __restrict__ hits,
                                                        const uint32_t* __restrict__ perm, int nres,
                                                        rt_ray* __restrict__ records)
{
	int s = blockIdx.x * BLOCK + threadIdx.x;
	if (s >= nres) return;
	uint32_t i = perm[s];
	HitInfo hi;
	float tc[2];
	resolve_tri(sv, hits[i], hi, tc);
	float* r = (float*)(records + s);
	for (int k = 0; k < 3; ++k) {
		r[k] = org[3 * (size_t)i + k];
		r[3 + k] = dir[3 * (size_t)i + k];
		r[6 + k] = hi.pos[k];
		r[9 + k] = hi.nrm[k];
	}
	r[12] = tc[0]; r[13] = tc[1]; r[14] = 0.0f; r[15] = 0.0f;
}

// The same from the ray's side: ray i, if it hit, writes its record to slot dest[i] (its rank in the
// one-pass grouping; misses rank last and write nothing).  Needs neither the inverse permutation nor
// the hit count, so it is queued before the host reads the group sizes.
__global__ __launch_bounds__(BLOCK) void k_emit_records_scatter(SceneView sv, const float* __restrict__ org,
                                                                const float* __restrict__ dir, const Hit* __restrict__ hits,
                                                                const uint32_t* __restrict__ dest, int n,
                                                                rt_ray* __restrict__ records)
{
	int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n) return;
	const Hit h = hits[i];
	if (h.prim == RT_MISS) return;
	HitInfo hi;
	float tc[2];
	resolve_tri(sv, h, hi, tc);
	float* r = (float*)(records + dest[i]);
	for (int k = 0; k < 3; ++k) {
		r[k] = org[3 * (size_t)i + k];
		r[3 + k] = dir[3 * (size_t)i + k];
		r[6 + k] = hi.pos[k];
		r[9 + k] = hi.nrm[k];
	}
	r[12] = tc[0]; r[13] = tc[1]; r[14] = 0.0f; r[15] = 0.0f;
}

// ---- G-buffer select -------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_select_digit(const float* __restrict__ nrm_rgba, int n, int interlace,
                                                        int phase, uint32_t* __restrict__ digit)
{
	int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n) return;
	bool keep = nrm_rgba[4 * (size_t)i + 3] > 0.0f;
	if (interlace > 1 && (i % interlace) != phase) keep = false;
	digit[i] = keep ? 0u : 1u;
}

__global__ __launch_bounds__(BLOCK) void k_select_scatter(const float* __restrict__ nrm_rgba,
                                                          const float* __restrict__ pos_rgb,
                                                          const uint32_t* __restrict__ digit,
                                                          const uint32_t* __restrict__ dest, int n,
                                                          float* __restrict__ pos_out, float* __restrict__ nrm_out,
                                                          int* __restrict__ index_out)
{
	int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n || digit[i] != 0u) return;
	uint32_t d = dest[i];
	for (int k = 0; k < 3; ++k) {
		// (*normals)[i][k] = buffer * 2.0 - 1.0 in double, stored to float (ras_rasterizer.cpp:533-535)
		nrm_out[3 * (size_t)d + k] = (float)((double)nrm_rgba[4 * (size_t)i + k] * 2.0 - 1.0);
		pos_out[3 * (size_t)d + k] = pos_rgb[3 * (size_t)i + k];
	}
	if (index_out) index_out[d] = i;
}

// ---- the G-buffer in its native 16-bit formats (ras_rasterizer.cpp:293,301) --------------------------
// What glReadPixels(GL_FLOAT) does on the way out of the GL (ras_rasterizer.cpp:497,543), on this side of PCIe:
// UNORM16 -> c / 65535 and half -> float, both exact, so the float planes are the ones the float entries take.
// The alpha word carries material id + 1 as an integer (this build's G-buffer convention; the reference's is 0 / 1).
__global__ __launch_bounds__(BLOCK) void k_decode_g16(const ushort4* __restrict__ nrm16, const uint16_t* __restrict__ pos16, int n,
                                                      float4* __restrict__ nrm_rgba, float* __restrict__ pos_rgb)
{
	const int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n) return;
	const ushort4 q = nrm16[i];
	nrm_rgba[i] = make_float4((float)q.x / 65535.0f, (float)q.y / 65535.0f, (float)q.z / 65535.0f, (float)q.w);
	for (int k = 0; k < 3; ++k) {
		__half_raw hr; hr.x = pos16[3 * (size_t)i + k];
		pos_rgb[3 * (size_t)i + k] = __half2float(__half(hr));
	}
}

// ---- software G-buffer (primary rays) -----------------------------------------------------------
__device__ __forceinline__ float unorm16(float c)
{
	c = fminf(fmaxf(c, 0.0f), 1.0f);
	return rintf(c * 65535.0f) / 65535.0f;
}
__device__ __forceinline__ float half_round(float f) { return __half2float(__float2half_rn(f)); }

// 8x8 pixel tiles: a wave's 64 lanes cover a square, not a 64x1 strip (2-D coherent rays share
// BVH nodes).  Item j of a w x h frame -> pixel index, or -1 in the padding of a partial tile.
__device__ __forceinline__ int tile8_pixel(int j, int w, int h)
{
	const int tiles_x = (w + 7) >> 3;
	const int t = j >> 6, k = j & 63;
	const int x = ((t % tiles_x) << 3) + (k & 7), y = ((t / tiles_x) << 3) + (k >> 3);
	return (x < w && y < h) ? y * w + x : -1;
}
__host__ __device__ inline int tile8_count(int w, int h) { return ((w + 7) >> 3) * ((h + 7) >> 3) * 64; }

__global__ __launch_bounds__(TRACE_BLOCK) void k_primary_gbuffer(SceneView sv, float tanfov, int w, int h,
                                                                 float* __restrict__ nrm_rgba,
                                                                 float* __restrict__ pos_rgb, float* __restrict__ uv)
{
	extern __shared__ __attribute__((aligned(16))) int smem[];
	lds_int* stack = RT_LDS_STACK(smem) + threadIdx.x;
	grid_for_each(sv, tile8_count(w, h), [&](int64_t j) {
		const int i = tile8_pixel((int)j, w, h);
		if (i < 0) return;
		int x = i % w, y = i / w;
		float aspect = float(w) / h;
		float dirX = (float)(2 * ((x + 0.5) / w) - 1);
		float dirY = (float)(2 * ((y + 0.5) / h) - 1);
		dirX *= aspect;
		f3 D = mk3(dirX * tanfov, dirY * tanfov, -1.0f);
		f3 O = mk3(0, 0, 0);
		float alpha = 0.0f;
		float tc[2] = {0.0f, 0.0f};
		f3 P = mk3(0, 0, 0), N = mk3(0, 0, 1);
		if (sv.num_spheres > 0 && sv.num_tris == 0) {
			// nearest sphere along the TRUE ray (the G-buffer producer stands in for the rasterizer,
			// it does not inherit AccelSpheres::intersect's origin quirk)
			float bestd = __builtin_inff();
			f3 Dn = normalized3(D);
			for (int k = 0; k < sv.num_spheres; ++k) {
				f3 X, Nn;
				if (sphere_test_ref(sv.spheres[k], O, Dn, X, Nn)) {
					float d2 = dot3(X, X);
					if (d2 < bestd) { bestd = d2; P = X; N = Nn; alpha = (float)(sv.num_materials - sv.num_spheres + k + 1); }
				}
			}
		} else {
			Hit best;
			scene_trace<false>(sv, O, D, best, stack, TRACE_BLOCK);
			if (best.prim != RT_MISS) {
				HitInfo hi;
				resolve_tri(sv, best, hi, uv ? tc : nullptr);
				P = ld3(hi.pos); N = ld3(hi.nrm);
				alpha = (float)(hi.material + 1);
			}
		}
		f3 nn = normalized3(N);
		nrm_rgba[4 * (size_t)i + 0] = unorm16(nn.x * 0.5f + 0.5f);
		nrm_rgba[4 * (size_t)i + 1] = unorm16(nn.y * 0.5f + 0.5f);
		nrm_rgba[4 * (size_t)i + 2] = unorm16(nn.z * 0.5f + 0.5f);
		nrm_rgba[4 * (size_t)i + 3] = alpha;
		pos_rgb[3 * (size_t)i + 0] = half_round(P.x);
		pos_rgb[3 * (size_t)i + 1] = half_round(P.y);
		pos_rgb[3 * (size_t)i + 2] = half_round(P.z);
		if (uv) { uv[2 * (size_t)i] = tc[0]; uv[2 * (size_t)i + 1] = tc[1]; }
	});
}

// ---- Gen-1 shade() wavefront ---------------------------------------------------------------------
__device__ __forceinline__ bool tile_is_mine(int pixel, int w, int tile, int rank, int nranks)
{
	if (nranks <= 1) return true;
	int x = pixel % w, y = pixel / w;
	int tx = x / tile, ty = y / tile;
	int tiles_x = (w + tile - 1) / tile;
	(void)tiles_x;
	// round robin along a row, rotated by the row index to decorrelate columns
	return ((tx + ty) % nranks) == rank;
}

// Level-0 nodes are stored in 8x8-tile order (node j <-> pixel tile8_pixel(j)); every later level
// is a compaction of its parent level, so the whole tree of a frame inherits the 2-D locality.
__global__ __launch_bounds__(CBLOCK) void k_init_nodes(const float* __restrict__ pos_rgb,
                                                      const float* __restrict__ nrm_rgba, int w, int h, int n0, int tile,
                                                      int rank, int nranks, int num_materials,
                                                      NodeGeo* __restrict__ geo, float* __restrict__ inc,
                                                      int* __restrict__ active_count,
                                                      const float* __restrict__ uv_plane, float* __restrict__ node_uv)
{
	int j = blockIdx.x * CBLOCK + threadIdx.x;
	bool active = false;
	if (j < n0) {
		const int i = tile8_pixel(j, w, h);
		NodeGeo g;
		int mat = -2;  // -2: padding of a partial tile, or not this rank's texel: leave the pixel alone
		f3 P = mk3(0, 0, 0), N = mk3(0, 0, 0);
		if (i >= 0) {
			float a = nrm_rgba[4 * (size_t)i + 3];
			mat = (a > 0.0f) ? ((int)a - 1) : -1;
			if (mat >= num_materials) mat = -1;
			if (!tile_is_mine(i, w, tile, rank, nranks)) mat = -2;
			P = ld3(pos_rgb + 3 * (size_t)i);
			// decode as getRayTraceData does (c*2-1 in double, not renormalised)
			N.x = (float)((double)nrm_rgba[4 * (size_t)i + 0] * 2.0 - 1.0);
			N.y = (float)((double)nrm_rgba[4 * (size_t)i + 1] * 2.0 - 1.0);
			N.z = (float)((double)nrm_rgba[4 * (size_t)i + 2] * 2.0 - 1.0);
		}
		st3(g.V, P); st3(g.N, N);
		st3(inc + 3 * (size_t)j, mat >= 0 ? normalized3(P) : mk3(0, 0, 0));
		g.parent = i; g.material = mat;
		geo[j] = g;
		active = mat >= 0;
		if (node_uv) {  // textured scenes only
			const bool have = uv_plane && i >= 0;
			node_uv[2 * (size_t)j] = have ? uv_plane[2 * (size_t)i] : 0.0f;
			node_uv[2 * (size_t)j + 1] = have ? uv_plane[2 * (size_t)i + 1] : 0.0f;
		}
	}
	// one counter add per workgroup
	__shared__ int s_cnt;
	if (threadIdx.x == 0) s_cnt = 0;
	__syncthreads();
	uint64_t m = __ballot(active);
	if (m && (threadIdx.x & 63) == 0) atomicAdd(&s_cnt, (int)__popcll(m));
	__syncthreads();
	if (threadIdx.x == 0 && s_cnt) atomicAdd(active_count, s_cnt);
}

// Shadow rays: one lane per (light, node); lanes of a wave share the light, so the rays of a
// wave converge on one point.  occl[node] gets bit `light` when the ray is blocked.
// Level 0 in one pass (sync-free driver): k_init_nodes + the gate / reflection-ray part of
// k_light_spawn_reflect + cleared shade and shadow mask, so the G-buffer is read once and the node
// records are written once instead of written, re-read and re-written by three launches.
// Level-0 node j of a frame -> pixel.  Without a tile list: 8x8 blocks of the whole frame.  With
// one: the rank's own tiles one after the other, 8x8 blocks inside each tile (tile % 8 == 0).
// `slot` = where the pixel sits in the rank's packed chunk (tile after tile, row-major inside a tile).
__device__ __forceinline__ int level0_pixel(const FrameSet& fs, int jj, int w, int h, int tile, int& slot)
{
	slot = 0;
	if (!fs.tile_list) return tile8_pixel(jj, w, h);
	const int t2 = tile * tile, tiles_x = (w + tile - 1) / tile, sub_x = tile >> 3;
	const int kt = jj / t2, id = fs.tile_list[kt], r = jj % t2;
	const int sub = r >> 6, k = r & 63;
	const int lx = ((sub % sub_x) << 3) + (k & 7), ly = ((sub / sub_x) << 3) + (k >> 3);
	const int x = (id % tiles_x) * tile + lx, y = (id / tiles_x) * tile + ly;
	slot = kt * t2 + ly * tile + lx;
	return (x < w && y < h) ? y * w + x : -1;
}

// ---- the shared descent of a wave's shadow rays ---------------------------------------------------------
// The 64 shadow rays of a wave start on neighbouring texels (one 8x8 tile at level 0) and aim at one light.  Every ray
// starts ON a surface, hence inside every ancestor box of the leaves around its origin: measured on the 1080p frame
// (profiles/r3_variants/k_shadow.md), the first 17 of a wave's 37 traversal iterations visit the SAME node in all 64
// lanes -- the descent from the root to the neighbourhood of the origins, 64 times the same box tests.  That descent is
// made once per (8x8 tile of level 0, light) by ONE lane of a small kernel, against the group as a whole:
//   * the group's origins and direction vectors are bounded by boxes (k_init_spawn, group_box_store);
//   * a child box is classified against the group by the interval form of the slab test (every origin in the origin box,
//     every direction in the direction box -- a superset of the group's rays): missed by ALL rays, hit by ALL, or mixed;
//   * the walk (k_shadow_prefix) is the traversal every ray of the group would make, for as long as that is decidable:
//     step into a child all rays hit, keep the other child on the stack unless all rays miss it, pop when both are
//     missed, stop before stepping into a mixed box;
//   * the record = where the walk stopped (entry) + its stack.
// k_shadow then starts each ray of the group at `entry` with that stack.  A subtree is dropped only when it is missed by
// every ray of the group, so a ray visits at least what it would have visited by itself and its any-hit result is what it
// was (tests: frames bit-identical with tuning.shadow_prefix = 0 and > 0, and against the oracle).
struct GroupBox {  // 48 B; omin[0] > omax[0]: the group has no active node
	float omin[3], omax[3];  // origins O = V + 0.1 N
	float vmin[3], vmax[3];  // node positions V (direction = light - V)
};
constexpr int PREFIX_INTS = 32;             // record: [0] siblings kept, [1] entry reference, [2 ..] the siblings, bottom of the stack first
constexpr int PREFIX_MAX = PREFIX_INTS - 2;

// min / max over the wave through DPP (six vector instructions per value, no LDS traffic: __shfl_xor is a ds_bpermute
// each, and twelve reductions of them made k_init_spawn 14 us longer); the result is valid in lane 63.
// quad swaps and the two mirrors leave every lane of a row of 16 with the row's value (min / max are idempotent), then
// lane 15 of a row is broadcast into the next row (rows 1, 3) and lane 31 into rows 2, 3.
#define RT_DPP(v, ctrl, row_mask) __builtin_amdgcn_update_dpp(v, v, ctrl, row_mask, 0xf, false)
__device__ __forceinline__ float wave_min_to_lane63(float x)
{
	int v = __float_as_int(x);
	v = __float_as_int(fminf(__int_as_float(v), __int_as_float(RT_DPP(v, 0xB1, 0xf))));   // quad_perm [1,0,3,2]
	v = __float_as_int(fminf(__int_as_float(v), __int_as_float(RT_DPP(v, 0x4E, 0xf))));   // quad_perm [2,3,0,1]
	v = __float_as_int(fminf(__int_as_float(v), __int_as_float(RT_DPP(v, 0x141, 0xf))));  // row_half_mirror
	v = __float_as_int(fminf(__int_as_float(v), __int_as_float(RT_DPP(v, 0x140, 0xf))));  // row_mirror
	v = __float_as_int(fminf(__int_as_float(v), __int_as_float(RT_DPP(v, 0x142, 0xa))));  // row_bcast:15 into rows 1 and 3
	v = __float_as_int(fminf(__int_as_float(v), __int_as_float(RT_DPP(v, 0x143, 0xc))));  // row_bcast:31 into rows 2 and 3
	return __int_as_float(v);
}
__device__ __forceinline__ float wave_max_to_lane63(float x)
{
	int v = __float_as_int(x);
	v = __float_as_int(fmaxf(__int_as_float(v), __int_as_float(RT_DPP(v, 0xB1, 0xf))));
	v = __float_as_int(fmaxf(__int_as_float(v), __int_as_float(RT_DPP(v, 0x4E, 0xf))));
	v = __float_as_int(fmaxf(__int_as_float(v), __int_as_float(RT_DPP(v, 0x141, 0xf))));
	v = __float_as_int(fmaxf(__int_as_float(v), __int_as_float(RT_DPP(v, 0x140, 0xf))));
	v = __float_as_int(fmaxf(__int_as_float(v), __int_as_float(RT_DPP(v, 0x142, 0xa))));
	v = __float_as_int(fmaxf(__int_as_float(v), __int_as_float(RT_DPP(v, 0x143, 0xc))));
	return __int_as_float(v);
}
#undef RT_DPP

// Bounds of the wave's 64 nodes (inactive lanes contribute nothing); lane 63 of the wave stores them.
__device__ __forceinline__ void group_box_store(bool active, f3 V, f3 N, GroupBox* __restrict__ dst)
{
	const float inf = __builtin_inff();
	float lo[6] = {inf, inf, inf, inf, inf, inf}, hi[6] = {-inf, -inf, -inf, -inf, -inf, -inf};
	if (active) {
		const f3 O = V + N * 0.1f;  // as k_shadow forms it
		lo[0] = hi[0] = O.x; lo[1] = hi[1] = O.y; lo[2] = hi[2] = O.z;
		lo[3] = hi[3] = V.x; lo[4] = hi[4] = V.y; lo[5] = hi[5] = V.z;
	}
	for (int k = 0; k < 6; ++k) { lo[k] = wave_min_to_lane63(lo[k]); hi[k] = wave_max_to_lane63(hi[k]); }
	if ((threadIdx.x & 63) == 63) {
		GroupBox b;
		for (int k = 0; k < 3; ++k) { b.omin[k] = lo[k]; b.omax[k] = hi[k]; b.vmin[k] = lo[3 + k]; b.vmax[k] = hi[3 + k]; }
		*dst = b;
	}
}

__global__ __launch_bounds__(CBLOCK) void k_init_spawn(SceneView sv, FrameSet fs, int w, int h, int n0, int tile,
                                                      int rank, int nranks, int recurse, NodeGeo* __restrict__ geo,
                                                      float* __restrict__ inc, int* __restrict__ gate,
                                                      unsigned int* __restrict__ occl, float* __restrict__ q_org,
                                                      float* __restrict__ q_dir, int* __restrict__ q_parent,
                                                      int* __restrict__ q_count, int* __restrict__ active_count,
                                                      float* __restrict__ node_uv, GroupBox* __restrict__ group_boxes)
{
	int j = blockIdx.x * CBLOCK + threadIdx.x;
	bool active = false, emit = false;
	f3 P = mk3(0, 0, 0), R = mk3(0, 0, 0), Nrm = mk3(0, 0, 0);
	if (j < n0) {
		const int f = j / fs.per_frame;
		int slot;
		const int i = level0_pixel(fs, j - f * fs.per_frame, w, h, tile, slot);
		const float* __restrict__ pos_rgb = fs.pos[f];
		const float* __restrict__ nrm_rgba = fs.nrm[f];
		int mat = -2;  // -2: padding of a partial tile, or not this rank's texel: leave the pixel alone
		f3 N = mk3(0, 0, 0);
		if (i >= 0) {
			float a = nrm_rgba[4 * (size_t)i + 3];
			mat = (a > 0.0f) ? ((int)a - 1) : -1;
			if (mat >= sv.num_materials) mat = -1;
			if (!tile_is_mine(i, w, tile, rank, nranks)) mat = -2;
			P = ld3(pos_rgb + 3 * (size_t)i);
			// decode as getRayTraceData does (c*2-1 in double, not renormalised)
			N.x = (float)((double)nrm_rgba[4 * (size_t)i + 0] * 2.0 - 1.0);
			N.y = (float)((double)nrm_rgba[4 * (size_t)i + 1] * 2.0 - 1.0);
			N.z = (float)((double)nrm_rgba[4 * (size_t)i + 2] * 2.0 - 1.0);
		}
		active = mat >= 0;
		Nrm = N;
		const f3 I = active ? normalized3(P) : mk3(0, 0, 0);
		NodeGeo g;
		st3(g.V, P); st3(g.N, N);
		if (inc) st3(inc + 3 * (size_t)j, I);  // (the sync-free driver passes nullptr: level 0's incident direction is normalize(V), recomputed where it is needed)
		// where k_combine writes this node's pixel: frame f's image, or (packed) the rank's chunk of frame f
		g.parent = i < 0 ? -1 : (fs.packed ? f * fs.per_frame + slot : f * (w * h) + i);
		g.material = mat;
		geo[j] = g;
		gate[j] = active ? mat : -1;
		occl[j] = 0u;  // shadow mask + child flags; NodeShade stays unwritten until a child returns a colour
		if (node_uv) {  // textured scenes only
			const float* __restrict__ uvp = fs.uv[f];
			const bool have = uvp && i >= 0;
			node_uv[2 * (size_t)j] = have ? uvp[2 * (size_t)i] : 0.0f;
			node_uv[2 * (size_t)j + 1] = have ? uvp[2 * (size_t)i + 1] : 0.0f;
		}
		if (active && recurse && sv.materials[mat].reflectivity > 0.0f) {
			R = I - N * (2.0f * dot3(I, N));
			emit = true;
		}
	}
	if (group_boxes && (j & ~63) < n0) group_box_store(active, P, Nrm, group_boxes + (j >> 6));  // (the shared descent of the shadow rays)
	// two compactions share the workgroup's barriers: active count and reflection queue
	__shared__ int s_cnt;
	if (threadIdx.x == 0) s_cnt = 0;
	__syncthreads();
	uint64_t m = __ballot(active);
	if (m && (threadIdx.x & 63) == 0) atomicAdd(&s_cnt, (int)__popcll(m));
	__syncthreads();
	if (threadIdx.x == 0 && s_cnt) atomicAdd(active_count, s_cnt);
	int slot = block_compact_slot(emit, q_count);
	if (emit) {
		st3(q_org + 3 * (size_t)slot, P);
		st3(q_dir + 3 * (size_t)slot, R);
		q_parent[slot] = j;
	}
}

// Optionally the launch also carries a queue of closest-hit rays (the level's reflection or
// refraction rays): every thread first takes its share of that queue, then of the shadow rays.  On
// its own such a queue is a short, latency-bound launch (one generation of waves, as slow as its
// slowest ray); inside the shadow launch its stalls are covered by shadow-ray work.
struct FusedQueue {
	const float* org;      // nullptr: no queue
	const float* dir;
	const int* count_ptr;  // count on the device (nullptr: `capacity` is the count)
	int capacity;
	Hit* hits;
	unsigned long long* census;
};

// One lane per (light, group).  Record (light * ngroups + group), ngroups = ceil(n / 64): where k_shadow's item
// light * 64 * ngroups + node looks for it.
__global__ __launch_bounds__(BLOCK) void k_shadow_prefix(SceneView sv, const GroupBox* __restrict__ boxes, int n,
                                                         const int* __restrict__ n_ptr, int nlights, float shadow_t_max,
                                                         int max_push, int max_steps, int* __restrict__ prefix)
{
	__shared__ int s_stack[PREFIX_MAX][BLOCK];  // the group's stack while it is traced: [entry][lane], conflict-free
	n = dev_count(n, n_ptr);
	const int ngroups = (n + 63) >> 6;
	const int64_t id = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (id >= (int64_t)ngroups * nlights) return;
	const int light = (int)(id / ngroups), grp = (int)(id - (int64_t)light * ngroups);
	int* rec = prefix + id * PREFIX_INTS;
	int cur = (sv.num_tris == 1) ? ~0 : 0, cnt = 0;
	const GroupBox b = boxes[grp];
	if (!(b.omin[0] <= b.omax[0])) { rec[0] = 0; rec[1] = cur; return; }
	const float t_max = shadow_t_max > 0.0f ? shadow_t_max : sv.tp.t_max;
	// Per axis, mirrored so that every direction component is positive: origin in [o0, o1], direction in [d0, d1] with
	// d0 > 0 (float subtraction is monotone, so light - vmax <= light - V <= light - vmin holds for the rounded values the
	// rays use), r0 = 1 / d0 >= r1 = 1 / d1.  With the box's entry plane at A * q + Ce and its exit plane at A * q + Cx
	// (q = the plane's 16-bit index; the planes pushed outward by two grid cells and by the rounding of the decode), the
	// rays of the group reach the entry plane no earlier than min(e r0, e r1), e = A q + Ce - o1, and so on: each bound is
	// two FMAs of q with per-lane constants, widened by `err`, the rounding bound of such an FMA (cancellation included).
	// An axis on which the directions change sign (`open`) bounds nothing unless the origins lie inside the slab.
	struct AxisPre {
		float a0, a1;          // A r0, A r1
		float el0, el1;        // (Ce - o1) r0 - err, (Ce - o1) r1 - err : lower bound of t_entry
		float eh0, eh1;        // (Ce - o0) r0 + err, ...                : upper bound of t_entry
		float xl0, xl1;        // (Cx - o1) ...                          : lower bound of t_exit
		float xh0, xh1;        // (Cx - o0) ...                          : upper bound of t_exit
		int she, shx;          // which half-word of the plane pair is the entry / exit plane
	} ax[3];
	bool open[3];
	float blo_pad[3], bhi_pad[3];
	const float* Lp = sv.lights[light].position;
	for (int k = 0; k < 3; ++k) {
		const float pad = 2.0f * sv.qframe.scale[k] + 1e-6f * sv.qframe.reach[k];
		blo_pad[k] = sv.qframe.org[k] - pad; bhi_pad[k] = sv.qframe.org[k] + pad;
		float d0 = Lp[k] - b.vmax[k], d1 = Lp[k] - b.vmin[k];
		open[k] = !(d0 > 0.0f || d1 < 0.0f);
		const bool neg = d1 < 0.0f;
		float o0 = b.omin[k], o1 = b.omax[k];
		if (neg) { const float t = -d0; d0 = -d1; d1 = t; o0 = -b.omax[k]; o1 = -b.omin[k]; }
		const float r0 = 1.0f / d0, r1 = 1.0f / d1;
		const float A = neg ? -sv.qframe.scale[k] : sv.qframe.scale[k];
		const float Ce = (neg ? -sv.qframe.org[k] : sv.qframe.org[k]) - pad, Cx = (neg ? -sv.qframe.org[k] : sv.qframe.org[k]) + pad;
		const float err = 1e-6f * (sv.qframe.reach[k] + pad + fmaxf(fabsf(o0), fabsf(o1))) * r0;
		AxisPre& p = ax[k];
		p.a0 = A * r0; p.a1 = A * r1;
		p.el0 = (Ce - o1) * r0 - err; p.el1 = (Ce - o1) * r1 - err;
		p.eh0 = (Ce - o0) * r0 + err; p.eh1 = (Ce - o0) * r1 + err;
		p.xl0 = (Cx - o1) * r0 - err; p.xl1 = (Cx - o1) * r1 - err;
		p.xh0 = (Cx - o0) * r0 + err; p.xh1 = (Cx - o0) * r1 + err;
		p.she = neg ? 16 : 0; p.shx = neg ? 0 : 16;
	}
	// The group is traced like one ray (nearer child first, the other on the stack, pop when both are missed) for as long
	// as every decision is the same for all of its rays: a child box is MISSED by all (dropped -- the only decision results
	// depend on), HIT by all, or MIXED.  The walk steps into a child only when all rays hit it, may leave a MIXED child on
	// the stack beside such a step (the rays that miss it spend one visit there), and stops where it would have to step
	// into a MIXED box, at a leaf, or when the record is full.
	enum { MISSED = 0, MIXED = 1, ALL = 2 };
	uint32_t all_mask = 0;  // bit k: stack entry k is hit by all rays
	// The walk is one chain of dependent node fetches per lane, and the kernel is as long as its longest chain: both
	// children of a node are requested as soon as the node has arrived (the next node is one of them unless the walk
	// pops), so a step's fetch runs under the previous step's arithmetic.
	uint4 w0 = make_uint4(0, 0, 0, 0), w1 = w0, l0 = w0, l1 = w0, q0 = w0, q1 = w0;
	if (cur >= 0) { const uint4* np = (const uint4*)((const char*)sv.qnodes + (uint32_t)cur); w0 = np[0]; w1 = np[1]; }
	for (int step = 0; step < max_steps && cur >= 0; ++step) {
		const int ref[2] = {(int)w1.z, (int)w1.w};
		if (ref[0] >= 0) { const uint4* cp = (const uint4*)((const char*)sv.qnodes + (uint32_t)ref[0]); l0 = cp[0]; l1 = cp[1]; }
		if (ref[1] >= 0) { const uint4* cp = (const uint4*)((const char*)sv.qnodes + (uint32_t)ref[1]); q0 = cp[0]; q1 = cp[1]; }
		const uint32_t pl[2][3] = {{w0.x, w0.y, w0.z}, {w0.w, w1.x, w1.y}};
		int cls[2];
		float key[2];
		for (int c = 0; c < 2; ++c) {
			float tn_lb = 0.0f, tn_ub = 0.0f, tf_lb = t_max, tf_ub = t_max;
			for (int k = 0; k < 3; ++k) {
				if (open[k]) {
					// directions of both signs: inside the slab every ray has t_entry <= 0 <= t_exit on this axis; otherwise nothing is known
					const float blo = __builtin_fmaf((float)(pl[c][k] & 0xffffu), sv.qframe.scale[k], blo_pad[k]);
					const float bhi = __builtin_fmaf((float)(pl[c][k] >> 16), sv.qframe.scale[k], bhi_pad[k]);
					const bool inside = b.omin[k] >= blo && b.omax[k] <= bhi;
					if (!inside) tn_ub = __builtin_inff();
					tf_lb = inside ? fminf(tf_lb, 0.0f) : -__builtin_inff();
					continue;
				}
				const AxisPre& p = ax[k];
				const float qe = (float)((pl[c][k] >> p.she) & 0xffffu), qx = (float)((pl[c][k] >> p.shx) & 0xffffu);
				tn_lb = fmaxf(tn_lb, fminf(__builtin_fmaf(qe, p.a0, p.el0), __builtin_fmaf(qe, p.a1, p.el1)));
				tn_ub = fmaxf(tn_ub, fmaxf(__builtin_fmaf(qe, p.a0, p.eh0), __builtin_fmaf(qe, p.a1, p.eh1)));
				tf_lb = fminf(tf_lb, fminf(__builtin_fmaf(qx, p.a0, p.xl0), __builtin_fmaf(qx, p.a1, p.xl1)));
				tf_ub = fminf(tf_ub, fmaxf(__builtin_fmaf(qx, p.a0, p.xh0), __builtin_fmaf(qx, p.a1, p.xh1)));
			}
			const bool missed = tn_lb - 1e-5f * fabsf(tn_lb) > tf_ub + 1e-5f * fabsf(tf_ub);
			const bool all = tn_ub + 1e-5f * fabsf(tn_ub) <= tf_lb - 1e-5f * fabsf(tf_lb);
			cls[c] = missed ? MISSED : (all ? ALL : MIXED);
			key[c] = tn_lb;
		}
		if (cls[0] == MISSED && cls[1] == MISSED) {  // pop
			if (cnt == 0) { cur = RT_CUR_EXIT; break; }  // nothing left: no ray of the group is occluded
			--cnt;
			cur = s_stack[cnt][threadIdx.x];
			if (!((all_mask >> cnt) & 1u) || cur < 0) break;  // a MIXED subtree: the rays decide for themselves from here
			const uint4* np = (const uint4*)((const char*)sv.qnodes + (uint32_t)cur);
			w0 = np[0]; w1 = np[1];
			continue;
		}
		const int first = (cls[1] == ALL && (cls[0] != ALL || key[1] < key[0])) ? 1 : 0;
		if (cls[first] != ALL) break;                  // no child that all rays enter: this node is where they start
		const int other = first ^ 1;
		if (cls[other] != MISSED) {
			if (cnt >= max_push) break;
			s_stack[cnt][threadIdx.x] = ref[other];
			all_mask = (all_mask & ~(1u << cnt)) | ((cls[other] == ALL ? 1u : 0u) << cnt);
			++cnt;
		}
		cur = ref[first];
		w0 = first ? q0 : l0; w1 = first ? q1 : l1;
	}
	for (int k = 0; k < cnt; ++k) rec[2 + k] = s_stack[k][threadIdx.x];
	rec[0] = cnt; rec[1] = cur;
}

static_assert(TRACE_BLOCK == 64, "a shadow wave = one group of 64 consecutive nodes");
// Any-hit traversal that starts where the group's shared descent stopped.  `rv`: lane k holds word k of the group's
// record (loaded by all 64 lanes of the wave).
template <bool CENSUS>
__device__ __forceinline__ bool bvh_trace_any_from(const SceneView& sv, f3 O, f3 D, int rv, lds_int* stack, uint32_t& cnt_nodes,
                                                   uint32_t& cnt_tris)
{
	const RayPre rp = ray_pre(sv.qframe, O, D);
	lds_int* const lim = stack + sv.lds_depth * TRACE_BLOCK;
	stack_push(sv, stack, lim, TRACE_BLOCK, RT_CUR_EXIT);
	lds_int* top = stack + TRACE_BLOCK;
	const int cnt = __builtin_amdgcn_readlane(rv, 0);
	int cur = __builtin_amdgcn_readlane(rv, 1);
	for (int k = 0; k < cnt; ++k) {
		stack_push(sv, top, lim, TRACE_BLOCK, __builtin_amdgcn_readlane(rv, 2 + k));
		top += TRACE_BLOCK;
	}
	Hit best;
	best.t = sv.tp.t_max;
	while (cur < RT_CUR_POP) bvh_step<true, CENSUS>(sv, rp, O, D, best, cur, top, lim, TRACE_BLOCK, cnt_nodes, cnt_tris);
	return cur == RT_CUR_HIT;
}

// waves_per_eu(8): left alone the kernel takes 74 VGPRs = 6 waves/SIMD; held to 64 (two spills, both
// outside the traversal loops) it runs at the full 8 and the launch measured 0.351 instead of 0.360 ms.
template <bool CENSUS>
__global__ __launch_bounds__(TRACE_BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_shadow(SceneView sv, const NodeGeo* __restrict__ geo, int n,
                                                        const int* __restrict__ n_ptr, int light_lo, int light_hi,
                                                        float shadow_t_max, unsigned int* __restrict__ occl,
                                                        unsigned long long* census, FusedQueue fq,
                                                        const int* __restrict__ prefix)
{
	n = dev_count(n, n_ptr);
	extern __shared__ __attribute__((aligned(16))) int smem[];
	lds_int* stack = RT_LDS_STACK(smem) + threadIdx.x;
	if (fq.org) {
		const int nq = dev_count(fq.capacity, fq.count_ptr);
		uint32_t cn = 0, ct = 0, rays = 0;
		grid_for_each(sv, nq, [&](int64_t i) {
			f3 O = ld3(fq.org + 3 * (size_t)i), D = ld3(fq.dir + 3 * (size_t)i);
			Hit best;
			uint32_t a = 0, b = 0;
			if (sv.use_bvh) bvh_trace<false, CENSUS>(sv, O, D, best, stack, TRACE_BLOCK, a, b);
			else brute_trace<false>(sv, O, D, best);
			cn += a; ct += b; ++rays;
			fq.hits[i] = best;
		});
		if (CENSUS && fq.census) census_add(fq.census, cn, ct, rays);
	}
	if (shadow_t_max > 0.0f) sv.tp.t_max = shadow_t_max;
	uint32_t cn = 0, ct = 0, rays = 0;
	// with the shared descent (prefix != nullptr) a light's items are padded to whole groups of 64 nodes, so that a wave
	// never holds rays of two lights: item = light * npad + node
	const int npad = prefix ? (n + 63) & ~63 : n;
	grid_for_each(sv, (int64_t)npad * (light_hi - light_lo), [&](int64_t idx) {
		int light = light_lo + (int)(idx / npad), i = (int)(idx % npad);
		int rv = 0;
		if (prefix) {  // every lane of the wave is here (npad and the grid step are multiples of 64): lane k fetches word k
			const int64_t group = ((int64_t)light_lo * npad + idx) >> 6;
			rv = prefix[group * PREFIX_INTS + (threadIdx.x & (PREFIX_INTS - 1))];
			if (i >= n) return;
		}
		const NodeGeo g = geo[i];
		if (g.material < 0) return;
		++rays;
		f3 V = ld3(g.V), N = ld3(g.N);
		f3 L = ld3(sv.lights[light].position) - V;
		f3 O = V + N * 0.1f;  // bias = 0.1 (rt_raytracer.cpp:79,112)
		bool blocked;
		if (sv.num_spheres > 0 && sv.num_tris == 0) {
			blocked = false;
			f3 X, Nn;
			for (int k = 0; k < sv.num_spheres; ++k)
				if (sphere_test_ref(sv.spheres[k], L, L, X, Nn)) { blocked = true; break; }
		} else if (sv.use_bvh) {
			Hit best;
			uint32_t a = 0, b = 0;
			if (prefix) blocked = bvh_trace_any_from<CENSUS>(sv, O, L, rv, stack, a, b);
			else blocked = bvh_trace<true, CENSUS>(sv, O, L, best, stack, TRACE_BLOCK, a, b);
			cn += a; ct += b;
		} else {
			Hit best;
			blocked = brute_trace<true>(sv, O, L, best);
		}
		// `distance < L.norm()` with distance == 0 (rt_raytracer.cpp:115-116).  0 < sqrt(x) <=> 0 < x for the correctly rounded square
		// root (denormals included; NaN fails both), so the test is made on the squared length
		if (blocked && 0.0f < dot3(L, L)) atomicOr(&occl[i], 1u << light);
	});
	if (CENSUS && census) census_add(census, cn, ct, rays);
}

// light_color() without a photon map (rt_raytracer.cpp:98-123): lambert accumulated and clamped
// light by light, -0.4 per occluded light (Q8).
__device__ __forceinline__ float light_factor(const SceneView& sv, f3 V, f3 N, unsigned int oc)
{
	float lambert = 0.0f, shadow = 1.0f;
	for (int l = 0; l < sv.num_lights; ++l) {
		f3 L = ld3(sv.lights[l].position) - V;
		lambert += dot3(N, normalized3(L));
		lambert = fminf(fmaxf(lambert, 0.0f), 1.0f);
		if ((oc >> l) & 1u) shadow = (float)fmax((double)shadow - 0.4, 0.0);
	}
	return shadow * lambert;
}

// Per node: lambert / shadow -> F, and the reflection ray if this level may recurse.
// defer_light (sync-free driver): F is left to k_combine, which runs after all shadow launches of the
// level; only the gate and the reflection rays are produced here and NodeShade is not written.
__global__ __launch_bounds__(CBLOCK) void k_light_spawn_reflect(SceneView sv, const NodeGeo* __restrict__ geo,
                                                               const float* __restrict__ inc,
                                                               NodeShade* __restrict__ shade, int n,
                                                               const int* __restrict__ n_ptr,
                                                               const unsigned int* __restrict__ occl,
                                                               const float* __restrict__ photon_rgb, int do_shadows,
                                                               int defer_light, int recurse, int* __restrict__ gate,
                                                               float* __restrict__ q_org, float* __restrict__ q_dir,
                                                               int* __restrict__ q_parent, int* __restrict__ q_count)
{
	n = dev_count(n, n_ptr);
	int i = blockIdx.x * CBLOCK + threadIdx.x;
	bool emit = false;
	f3 V = mk3(0, 0, 0), R = mk3(0, 0, 0);
	if (i < n) {
		const NodeGeo g = geo[i];
		NodeShade s;  // written only when the light term is evaluated here (defer_light == 0)
		s.pad0 = s.pad1 = s.pad2 = 0;
		st3(s.refc, mk3(0, 0, 0)); st3(s.refrc, mk3(0, 0, 0));
		if (g.material >= 0) {
			V = ld3(g.V);
			f3 N = ld3(g.N);
			const MaterialRec m = sv.materials[g.material];
			if (photon_rgb) {
				st3(s.F, ld3(photon_rgb + 3 * (size_t)i));  // shadow == 1
			} else if (defer_light) {
				st3(s.F, mk3(0, 0, 0));
			} else {
				float f = light_factor(sv, V, N, (do_shadows && occl) ? occl[i] : 0u);
				st3(s.F, mk3(f, f, f));
			}
			gate[i] = g.material;
			if (recurse && m.reflectivity > 0.0f) {
				const f3 I = ld3(inc + 3 * (size_t)i);
				R = I - N * (2.0f * dot3(I, N));
				emit = true;
			}
		} else {
			st3(s.F, mk3(0, 0, 0));
			gate[i] = -1;
		}
		if (!defer_light) shade[i] = s;
	}
	// wave ballots + one atomic per workgroup
	int slot = block_compact_slot(emit, q_count);
	if (emit) {
		st3(q_org + 3 * (size_t)slot, V);
		st3(q_dir + 3 * (size_t)slot, R);
		q_parent[slot] = i;
	}
}

// Refraction spawn: gated by the (possibly overwritten, quirk Q9) material.
__global__ __launch_bounds__(CBLOCK) void k_spawn_refract(SceneView sv, const NodeGeo* __restrict__ geo,
                                                         const float* __restrict__ inc, int n,
                                                         const int* __restrict__ n_ptr, const int* __restrict__ gate, float* __restrict__ q_org,
                                                         float* __restrict__ q_dir, int* __restrict__ q_parent,
                                                         int* __restrict__ q_count)
{
	n = dev_count(n, n_ptr);
	int i = blockIdx.x * CBLOCK + threadIdx.x;
	bool emit = false;
	f3 V = mk3(0, 0, 0), T = mk3(0, 0, 0);
	if (i < n) {
		int gm = gate[i];
		if (gm >= 0) {
			const MaterialRec m = sv.materials[gm];
			if (m.alpha > 0.1f) {
				const NodeGeo g = geo[i];
				V = ld3(g.V);
				// inc == nullptr: level 0, whose incident direction is normalize(V) (the eye is the origin) -- the same arithmetic
				// as the stored copy, 12 bytes per node the level-0 pass no longer writes
				f3 N = ld3(g.N), I = inc ? ld3(inc + 3 * (size_t)i) : normalized3(V);
				float IoR = m.ior;
				float idn = dot3(I, N);
				T = div3(I - N * idn, IoR);
				float arg = 1.0f - ((1.0f - idn * idn)) / IoR * IoR;  // (x / IoR) * IoR, the reference's precedence
				float sq = rt_sqrtf(arg);
				T = T - N * sq;
				emit = true;
			}
		}
	}
	int slot = block_compact_slot(emit, q_count);
	if (emit) {
		st3(q_org + 3 * (size_t)slot, V);
		st3(q_dir + 3 * (size_t)slot, T);
		q_parent[slot] = i;
	}
}

// Sphere closest for the queue (AccelSpheres::intersect semantics: Ro = direction, first in list).
__global__ __launch_bounds__(BLOCK) void k_queue_sphere_hits(SceneView sv, const float* __restrict__ q_dir, int nq,
                                                             const int* __restrict__ n_ptr, HitInfo* __restrict__ out)
{
	nq = dev_count(nq, n_ptr);
	int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i >= nq) return;
	f3 Rd = ld3(q_dir + 3 * (size_t)i);
	HitInfo hi;
	hi.material = -1; hi.pad = 0;
	f3 X, N;
	for (int k = 0; k < sv.num_spheres; ++k) {
		if (sphere_test_ref(sv.spheres[k], Rd, Rd, X, N)) {
			st3(hi.pos, X); st3(hi.nrm, N);
			hi.material = sv.num_materials - sv.num_spheres + k;  // sphere materials sit at the table's end
			break;
		}
	}
	out[i] = hi;
}

// Turn the hits of a queue into child nodes of the next level.  `hi` holds resolved hits, or --
// when raw_hits != nullptr -- the hits are resolved here (position, shading normal, material) and
// `hi` is not read.  child_occl (optional): the child's shadow mask is cleared on creation.
__global__ __launch_bounds__(CBLOCK) void k_make_children(SceneView sv, const HitInfo* __restrict__ hi,
                                                         const Hit* __restrict__ raw_hits,
                                                         const float* __restrict__ q_dir,
                                                         const int* __restrict__ q_parent, int nq,
                                                         const int* __restrict__ n_ptr, int branch,
                                                         int* __restrict__ gate, NodeGeo* __restrict__ child_geo,
                                                         float* __restrict__ child_inc,
                                                         unsigned int* __restrict__ child_occl,
                                                         int* __restrict__ child_count, int child_capacity,
                                                         int* __restrict__ overflow, const Hit* __restrict__ tc_hits,
                                                         float* __restrict__ child_uv,
                                                         unsigned int* __restrict__ parent_flags,
                                                         int* __restrict__ parent_child_slot)
{
	nq = dev_count(nq, n_ptr);
	int i = blockIdx.x * CBLOCK + threadIdx.x;
	bool hit = false;
	HitInfo h;
	float tc[2] = {0.0f, 0.0f};
	if (i < nq) {
		if (raw_hits) resolve_tri(sv, raw_hits[i], h, child_uv ? tc : nullptr);
		else {
			h = hi[i];
			// textured scenes: IAccel::intersect's texcoord write-back from the queue's raw hits (spheres: none)
			if (child_uv && tc_hits && h.material >= 0) { HitInfo tmp; resolve_tri(sv, tc_hits[i], tmp, tc); }
		}
		hit = h.material >= 0;
		if (hit && branch == 0) gate[q_parent[i]] = h.material;  // Q9: reflection hit overwrites `material`
	}
	int slot = block_compact_slot(hit, child_count);
	if (hit && slot >= child_capacity) { *overflow = 1; hit = false; }  // the host re-renders with exact sizes
	if (hit) {
		NodeGeo g;
		for (int k = 0; k < 3; ++k) { g.V[k] = h.pos[k]; g.N[k] = h.nrm[k]; child_inc[3 * (size_t)slot + k] = q_dir[3 * (size_t)i + k]; }
		const int parent = q_parent[i];
		g.parent = parent | (int)((unsigned int)branch << 31);
		g.material = h.material;
		child_geo[slot] = g;
		if (child_occl) child_occl[slot] = 0u;
		if (parent_flags) atomicOr(&parent_flags[parent], branch == 0 ? CHILD_REFL_BIT : CHILD_REFR_BIT);
		// where the child sits in its level, for a parent that fetches the child's colour itself (k_combine, pull form);
		// valid exactly where the flag above is set, like the colour slots of this record
		// (an array of its own, two ints per node: the parents of consecutive queue entries ascend, so these writes share lines)
		if (parent_child_slot) parent_child_slot[2 * (size_t)parent + branch] = slot;
		if (child_uv) { child_uv[2 * (size_t)slot] = tc[0]; child_uv[2 * (size_t)slot + 1] = tc[1]; }
	}
}

// Bottom-up combine of one level: color = (shadow*light) * ((kd*diff + ref*refc) + alpha*refrc)
// (rt_raytracer.cpp:158).  Level 0 writes RGBA8 as renderScene does (rt_raytracer.cpp:175-185).
// The colour shade() returns for one node, given what its children returned.
__device__ __forceinline__ f3 node_colour(const SceneView& sv, const NodeGeo& g, f3 F, f3 refc, f3 refrc, const float* __restrict__ node_uv,
                                          size_t i, bool lit, unsigned int oc)
{
	const MaterialRec m = sv.materials[g.material];
	float ref = m.reflectivity, alpha = m.alpha;
	float kd = (float)(1.0 - (double)ref - (double)alpha);
	f3 diff = ld3(m.diffuse);
	if (node_uv) {  // textured scenes only: diff_color = texture->lookup(texcoord) (rt_raytracer.cpp:93-95)
		const TexRec t = sv.tex[g.material];
		if (t.w > 0) diff = tex_lookup_ref(t, sv.texels, node_uv[2 * i], node_uv[2 * i + 1]);
	}
	f3 mix = (diff * kd + refc * ref) + refrc * alpha;
	if (lit) {  // deferred light_color(): the level's shadow rays have landed by now
		float f = light_factor(sv, ld3(g.V), ld3(g.N), oc);
		F = mk3(f, f, f);
	}
	return mk3(F.x * mix.x, F.y * mix.y, F.z * mix.z);
}

// The children one level down, when that level is the LAST one (its nodes have no children and nobody else needs their
// colours): this level's combine fetches and shades them itself (`pull`), so the last level needs no launch of its own
// and no colour is written to and read back from NodeShade.  Same expressions on the same operands: same bits.
struct ChildLevel {
	const NodeGeo* geo;          // nullptr: children pushed their colours into shade[] (a launch per level)
	const unsigned int* occl;
	const float* uv;
	const int* slot;             // [node][branch] -> the child's index in its level (valid where the node's child flag is set)
};

__global__ __launch_bounds__(BLOCK) void k_combine(SceneView sv, const NodeGeo* __restrict__ geo,
                                                   const NodeShade* __restrict__ shade, int n,
                                                   const int* __restrict__ n_ptr, int level,
                                                   NodeShade* __restrict__ parent_shade, FrameSet fs, int frame_pixels,
                                                   float* __restrict__ rgb_float, int lit,
                                                   const unsigned int* __restrict__ occl,
                                                   const float* __restrict__ node_uv, ChildLevel pull)
{
	n = dev_count(n, n_ptr);
	int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n) return;
	const NodeGeo g = geo[i];
	if (g.material == -2) return;
	f3 col = mk3(0, 0, 0);
	if (g.material >= 0) {
		NodeShade s;
		const unsigned int oc = (lit && occl) ? occl[i] : 0u;
		if (lit) {  // sync-free driver: a child's colour exists only where the flag says so
			st3(s.F, mk3(0, 0, 0));
			if (pull.geo) {
				const f3 zero = mk3(0, 0, 0);
				f3 rc = zero, rr = zero;
				if (oc & CHILD_REFL_BIT) {
					const size_t c = (size_t)pull.slot[2 * (size_t)i];
					const NodeGeo cg = pull.geo[c];
					rc = node_colour(sv, cg, zero, zero, zero, pull.uv, c, true, pull.occl[c]);
				}
				if (oc & CHILD_REFR_BIT) {
					const size_t c = (size_t)pull.slot[2 * (size_t)i + 1];
					const NodeGeo cg = pull.geo[c];
					rr = node_colour(sv, cg, zero, zero, zero, pull.uv, c, true, pull.occl[c]);
				}
				st3(s.refc, rc); st3(s.refrc, rr);
			} else {
				st3(s.refc, (oc & CHILD_REFL_BIT) ? ld3(shade[i].refc) : mk3(0, 0, 0));
				st3(s.refrc, (oc & CHILD_REFR_BIT) ? ld3(shade[i].refrc) : mk3(0, 0, 0));
			}
		} else {
			s = shade[i];
		}
		col = node_colour(sv, g, ld3(s.F), ld3(s.refc), ld3(s.refrc), node_uv, (size_t)i, lit != 0, oc);
	}
	if (level > 0) {
		float* dst = node_branch(g) == 0 ? parent_shade[node_parent(g)].refc : parent_shade[node_parent(g)].refrc;
		st3(dst, col);
	} else {
		const int f = g.parent / frame_pixels, pixel = g.parent - f * frame_pixels;  // level 0: parent = frame * pixels + pixel
		uint8_t* px = fs.rgba[f] + 4 * (size_t)pixel;
		if (g.material >= 0) {
			px[0] = (uint8_t)min((int)col.x, 255);
			px[1] = (uint8_t)min((int)col.y, 255);
			px[2] = (uint8_t)min((int)col.z, 255);
			px[3] = 255;
		} else {
			px[0] = px[1] = px[2] = px[3] = 0;
		}
		if (rgb_float) st3(rgb_float + 3 * (size_t)pixel, col);
	}
}

__global__ __launch_bounds__(BLOCK) void k_node_points(const NodeGeo* __restrict__ geo, int n, float* __restrict__ pos,
                                                       float* __restrict__ nrm)
{
	int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n) return;
	for (int k = 0; k < 3; ++k) { pos[3 * (size_t)i + k] = geo[i].V[k]; nrm[3 * (size_t)i + k] = geo[i].N[k]; }
}

// ---- live shading (a20) ----------------------------------------------------------------------------
// mesh_fs.cpp:26-49: per light energy = color/255 * 50/(50+|L|^2); direct += energy*albedo*clamp(N.L)
// + energy*scolor*pow(clamp(N.H), shininess), E = normalize(-pos); color = refl*reflection +
// (1-refl)*direct.  Material and light colours / 255 (ras_rasterizer.cpp:184-188,232); at most
// LIGHT_COUNT = 8 lights (mesh_fs.cpp:5); albedo from live_albedo() (texture or material colour).
__device__ __forceinline__ f3 mesh_fs_color(const SceneView& sv, const MaterialRec& m, f3 albedo, f3 pos, f3 nrm, f3 reflection)
{
	f3 scol = div3(ld3(m.specular), 255.0f);
	f3 N = normalized3(nrm);
	f3 E = normalized3(neg3(pos));
	f3 direct = mk3(0, 0, 0);
	const int nl = sv.num_lights < 8 ? sv.num_lights : 8;
	for (int l = 0; l < nl; ++l) {
		f3 L = ld3(sv.lights[l].position) - pos;
		f3 energy = ld3(sv.lights[l].color) * (1.0f / 255);
		float len = norm3(L);
		energy = energy * (50.0f / (50.0f + len * len));
		L = normalized3(L);
		f3 H = normalized3(L + E);
		float lambert = fminf(fmaxf(dot3(N, L), 0.0f), 1.0f);
		float phong = powf(fminf(fmaxf(dot3(N, H), 0.0f), 1.0f), m.shininess);
		direct.x = energy.x * albedo.x * lambert + energy.x * scol.x * phong + direct.x;
		direct.y = energy.y * albedo.y * lambert + energy.y * scol.y * phong + direct.y;
		direct.z = energy.z * albedo.z * lambert + energy.z * scol.z * phong + direct.z;
	}
	const float refl = m.reflectivity;
	return mk3(refl * reflection.x + (1.0f - refl) * direct.x, refl * reflection.y + (1.0f - refl) * direct.y,
	           refl * reflection.z + (1.0f - refl) * direct.z);
}

__device__ __forceinline__ uint8_t unorm8(float c) { return (uint8_t)rintf(fminf(fmaxf(c, 0.0f), 1.0f) * 255.0f); }

// The raypass of Scene::draw (drawRayData, ras_rasterizer.cpp:619-669) without GL: every hit of the
// selected texels is shaded at the HIT point with the hit's material and written, as RGB8, to the
// texel the ray started from (the reference draws a point at Ray.origin, which projects back onto
// that texel).  Misses draw nothing: the texel keeps what an earlier phase left there.
__global__ __launch_bounds__(BLOCK) void k_live_raypass(SceneView sv, const Hit* __restrict__ hits,
                                                        const int* __restrict__ texel, int n, uint8_t* __restrict__ raypass)
{
	int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n) return;
	HitInfo hi;
	float tc[2] = {0.0f, 0.0f};
	resolve_tri(sv, hits[i], hi, sv.tex ? tc : nullptr);
	if (hi.material < 0) return;
	const MaterialRec m = sv.materials[hi.material];
	f3 c = mesh_fs_color(sv, m, live_albedo(sv, hi.material, m, tc[0], tc[1]), ld3(hi.pos), ld3(hi.nrm), mk3(0, 0, 0));
	uint8_t* px = raypass + 3 * (size_t)texel[i];
	px[0] = unorm8(c.x); px[1] = unorm8(c.y); px[2] = unorm8(c.z);
}

// The forward pass (drawMeshes, :237-281; mesh_fs.cpp:46-47) from the G-buffer: the visible surface
// shaded with its own material, reflection = the raypass texel (scale 1: texel centres, so the
// GL_LINEAR fetch returns the texel), background = the clear colour 0.3.
__global__ __launch_bounds__(BLOCK) void k_live_composite(SceneView sv, const float* __restrict__ pos_rgb,
                                                          const float* __restrict__ nrm_rgba, int n,
                                                          const uint8_t* __restrict__ raypass, uint8_t* __restrict__ rgba,
                                                          const float* __restrict__ uv)
{
	int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n) return;
	float a = nrm_rgba[4 * (size_t)i + 3];
	int mat = (a > 0.0f) ? ((int)a - 1) : -1;
	if (mat >= sv.num_materials) mat = -1;
	uint8_t* px = rgba + 4 * (size_t)i;
	if (mat < 0) { px[0] = px[1] = px[2] = unorm8(0.3f); px[3] = 255; return; }
	f3 N;
	N.x = (float)((double)nrm_rgba[4 * (size_t)i + 0] * 2.0 - 1.0);
	N.y = (float)((double)nrm_rgba[4 * (size_t)i + 1] * 2.0 - 1.0);
	N.z = (float)((double)nrm_rgba[4 * (size_t)i + 2] * 2.0 - 1.0);
	const uint8_t* rp = raypass + 3 * (size_t)i;
	f3 refl = mk3(rp[0] / 255.0f, rp[1] / 255.0f, rp[2] / 255.0f);
	const MaterialRec m = sv.materials[mat];
	const float tu = uv ? uv[2 * (size_t)i] : 0.0f, tv = uv ? uv[2 * (size_t)i + 1] : 0.0f;
	f3 c = mesh_fs_color(sv, m, live_albedo(sv, mat, m, tu, tv), ld3(pos_rgb + 3 * (size_t)i), N, refl);
	px[0] = unorm8(c.x); px[1] = unorm8(c.y); px[2] = unorm8(c.z); px[3] = 255;
}

// ---- the live frame with a prepass smaller than the frame (ras_rasterizer.cpp:39,167-168: 0.75) -------------
// proj_mat of projectionFromCamera (ras_rasterizer.cpp:140-153; n = 1, f = 100, the FRAME's aspect ratio) applied to a
// point as mesh_vs.cpp:17 applies it (transform_draw = 0: no model matrix), then GL's clip test and viewport
// transform onto a pw x ph target and the rasterisation of a size-1 point: the fragment that contains the window
// position.  Returns the texel, or -1 when the point is clipped.
struct LiveProj {
	float m00, m11, m22, m32;
	int pw, ph;
};
__device__ __forceinline__ int live_point_texel(const LiveProj& pr, f3 P)
{
	const float cx = pr.m00 * P.x, cy = pr.m11 * P.y, cz = pr.m22 * P.z + pr.m32, cw = -P.z;
	if (!(cx >= -cw && cx <= cw && cy >= -cw && cy <= cw && cz >= -cw && cz <= cw)) return -1;  // (also w <= 0 and NaN)
	const float xw = (cx / cw) * (0.5f * (float)pr.pw) + 0.5f * (float)pr.pw;
	const float yw = (cy / cw) * (0.5f * (float)pr.ph) + 0.5f * (float)pr.ph;
	const int ix = (int)floorf(xw), iy = (int)floorf(yw);
	if (ix < 0 || ix >= pr.pw || iy < 0 || iy >= pr.ph) return -1;
	return iy * pr.pw + ix;
}

// drawRayData's points (ras_rasterizer.cpp:619-669): every hit record is drawn at the projection of Ray.origin
// (mesh_vs.cpp:17, location 0) into the prepass-sized raypass target and shaded at the hit.  Several records can land on
// one texel (the origin went through FP16); GL keeps the last one drawn, here: an unspecified one of them.
__global__ __launch_bounds__(BLOCK) void k_live_raypass_proj(SceneView sv, const Hit* __restrict__ hits, const float* __restrict__ org,
                                                             int n, LiveProj pr, uint8_t* __restrict__ raypass)
{
	int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n) return;
	HitInfo hi;
	float tc[2] = {0.0f, 0.0f};
	resolve_tri(sv, hits[i], hi, sv.tex ? tc : nullptr);
	if (hi.material < 0) return;
	const int texel = live_point_texel(pr, ld3(org + 3 * (size_t)i));
	if (texel < 0) return;
	const MaterialRec m = sv.materials[hi.material];
	f3 c = mesh_fs_color(sv, m, live_albedo(sv, hi.material, m, tc[0], tc[1]), ld3(hi.pos), ld3(hi.nrm), mk3(0, 0, 0));
	uint8_t* px = raypass + 3 * (size_t)texel;
	px[0] = unorm8(c.x); px[1] = unorm8(c.y); px[2] = unorm8(c.z);
}

// texture2D(raypassBuffer, gl_FragCoord.st / frame_size) (mesh_fs.cpp:46) on the pw x ph RGB8 target with GL_LINEAR
// filters (ras_rasterizer.cpp:600-601) and the default GL_REPEAT wrap: bilinear between the four texels around
// (u pw - 0.5, v ph - 0.5), indices taken modulo the size.
__device__ __forceinline__ f3 raypass_linear(const uint8_t* __restrict__ raypass, int pw, int ph, float u, float v)
{
	const float x = u * (float)pw - 0.5f, y = v * (float)ph - 0.5f;
	const float fx = floorf(x), fy = floorf(y);
	const float a = x - fx, b = y - fy;
	int x0 = (int)fx % pw, y0 = (int)fy % ph;
	if (x0 < 0) x0 += pw;
	if (y0 < 0) y0 += ph;
	const int x1 = (x0 + 1) % pw, y1 = (y0 + 1) % ph;
	auto tx = [&](int xx, int yy) {
		const uint8_t* p = raypass + 3 * ((size_t)yy * pw + xx);
		return mk3(p[0] / 255.0f, p[1] / 255.0f, p[2] / 255.0f);
	};
	const f3 t00 = tx(x0, y0), t10 = tx(x1, y0), t01 = tx(x0, y1), t11 = tx(x1, y1);
	const f3 lo = t00 * (1.0f - a) + t10 * a, hi = t01 * (1.0f - a) + t11 * a;
	return lo * (1.0f - b) + hi * b;
}

__global__ __launch_bounds__(BLOCK) void k_live_composite_scaled(SceneView sv, const float* __restrict__ pos_rgb,
                                                                 const float* __restrict__ nrm_rgba, int w, int h,
                                                                 const uint8_t* __restrict__ raypass, int pw, int ph,
                                                                 uint8_t* __restrict__ rgba, const float* __restrict__ uv)
{
	int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i >= w * h) return;
	float a = nrm_rgba[4 * (size_t)i + 3];
	int mat = (a > 0.0f) ? ((int)a - 1) : -1;
	if (mat >= sv.num_materials) mat = -1;
	uint8_t* px = rgba + 4 * (size_t)i;
	if (mat < 0) { px[0] = px[1] = px[2] = unorm8(0.3f); px[3] = 255; return; }
	f3 N;
	N.x = (float)((double)nrm_rgba[4 * (size_t)i + 0] * 2.0 - 1.0);
	N.y = (float)((double)nrm_rgba[4 * (size_t)i + 1] * 2.0 - 1.0);
	N.z = (float)((double)nrm_rgba[4 * (size_t)i + 2] * 2.0 - 1.0);
	const int x = i % w, y = i / w;
	const f3 refl = raypass_linear(raypass, pw, ph, ((float)x + 0.5f) / (float)w, ((float)y + 0.5f) / (float)h);
	const MaterialRec m = sv.materials[mat];
	const float tu = uv ? uv[2 * (size_t)i] : 0.0f, tv = uv ? uv[2 * (size_t)i + 1] : 0.0f;
	f3 c = mesh_fs_color(sv, m, live_albedo(sv, mat, m, tu, tv), ld3(pos_rgb + 3 * (size_t)i), N, refl);
	px[0] = unorm8(c.x); px[1] = unorm8(c.y); px[2] = unorm8(c.z); px[3] = 255;
}

// ---- live shading of hit records ---------------------------------------------------------------------
// mesh_fs.cpp:26-49 at one hit record, as drawRayData draws it into the raypass target: the
// record's group gives the material; reflection term = 0.  One lane per record.
__global__ __launch_bounds__(BLOCK) void k_shade_results(SceneView sv, const rt_ray* __restrict__ rec, int n,
                                                         const rt_result_offset* __restrict__ groups, int ngroups,
                                                         float* __restrict__ rgb, uint8_t* __restrict__ rgb8)
{
	int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n) return;
	// group g holds records [end(g-1), end(g)): first g with end(g) > i
	int lo = 0, hi = ngroups - 1;
	while (lo < hi) {
		int mid = (lo + hi) >> 1;
		if (groups[mid].end > i) hi = mid; else lo = mid + 1;
	}
	const int mat = groups[lo].material;
	const MaterialRec m = sv.materials[mat];
	const float* tcp = (const float*)(rec + i) + 12;  // Ray::texcoord
	f3 c = mesh_fs_color(sv, m, live_albedo(sv, mat, m, tcp[0], tcp[1]), ld3(rec[i].position), ld3(rec[i].normal), mk3(0, 0, 0));
	st3(rgb + 3 * (size_t)i, c);
	if (rgb8) {  // GL UNORM8 conversion of the RGB8 raypass target
		rgb8[3 * (size_t)i + 0] = unorm8(c.x);
		rgb8[3 * (size_t)i + 1] = unorm8(c.y);
		rgb8[3 * (size_t)i + 2] = unorm8(c.z);
	}
}

// ---- packed tile exchange (multi-GPU tile gather) ------------------------------------------------
// One lane per packed pixel: slot p belongs to tile tile_list[p / tile^2]; pixels of an edge tile
// outside the frame are padding (zero when packing, skipped when unpacking).
template <bool PACK>
__global__ __launch_bounds__(BLOCK) void k_tiles_pack(uint32_t* __restrict__ frame, int w, int h, int tile,
                                                      const int* __restrict__ tile_list, int ntiles,
                                                      uint32_t* __restrict__ packed)
{
	const int64_t p = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	const int t2 = tile * tile;
	if (p >= (int64_t)ntiles * t2) return;
	const int tiles_x = (w + tile - 1) / tile;
	const int id = tile_list[p / t2], k = (int)(p % t2);
	const int x = (id % tiles_x) * tile + k % tile, y = (id / tiles_x) * tile + k / tile;
	const bool inside = x < w && y < h;
	if (PACK) packed[p] = inside ? frame[(size_t)y * w + x] : 0u;
	else if (inside) frame[(size_t)y * w + x] = packed[p];
}

// All ranks' chunks of one received step into one frame: chunk r holds rank r's tiles.
struct ChunkLists {
	int offset[RT_MAX_BATCH + 1];  // rank r's tile ids are all_tiles[offset[r] .. offset[r + 1])
};
// (rank_stride: pixels from one rank's chunk to the next; larger than chunk_pixels when every rank sent several frames' chunks)
__global__ __launch_bounds__(BLOCK) void k_unpack_chunks(const uint32_t* __restrict__ chunks, int64_t chunk_pixels, int64_t rank_stride,
                                                         int nranks, ChunkLists cl, const int* __restrict__ all_tiles,
                                                         int w, int h, int tile, uint32_t* __restrict__ frame)
{
	const int64_t p = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	if (p >= chunk_pixels * nranks) return;
	const int r = (int)(p / chunk_pixels);
	const int64_t local = p - (int64_t)r * chunk_pixels;
	const int t2 = tile * tile, kt = (int)(local / t2);
	if (kt >= cl.offset[r + 1] - cl.offset[r]) return;  // padding at the end of a short chunk
	const int tiles_x = (w + tile - 1) / tile;
	const int id = all_tiles[cl.offset[r] + kt], k = (int)(local % t2);
	const int x = (id % tiles_x) * tile + k % tile, y = (id / tiles_x) * tile + k / tile;
	if (x < w && y < h) frame[(size_t)y * w + x] = chunks[(int64_t)r * rank_stride + local];
}

}  // namespace

// ================================================================================================
int rt_unpack_chunks(rt_hip_ctx* ctx, const uint8_t* d_chunks, int64_t chunk_bytes, int w, int h, int tile, int nranks,
                     uint8_t* d_frame, hipStream_t st, int64_t rank_stride_bytes)
{
	if (!st) st = ctx->stream;
	if (rank_stride_bytes <= 0) rank_stride_bytes = chunk_bytes;
	const int key[4] = {w, h, tile, nranks};
	const int tiles_x = (w + tile - 1) / tile, tiles_y = (h + tile - 1) / tile;
	if (memcmp(key, ctx->all_tiles_key, sizeof(key)) != 0 || !ctx->d_all_tiles.p) {
		std::vector<int> ids;
		for (int r = 0; r < nranks; ++r) {
			ctx->all_tiles_offset[r] = (int)ids.size();
			for (int ty = 0; ty < tiles_y; ++ty)
				for (int tx = 0; tx < tiles_x; ++tx)
					if ((tx + ty) % nranks == r) ids.push_back(ty * tiles_x + tx);
		}
		ctx->all_tiles_offset[nranks] = (int)ids.size();
		RT_CHECK(ctx->d_all_tiles.reserve(std::max<size_t>(ids.size(), 1) * sizeof(int)));
		RT_HIP_CHECK(hipMemcpy(ctx->d_all_tiles.p, ids.data(), ids.size() * sizeof(int), hipMemcpyHostToDevice));
		memcpy(ctx->all_tiles_key, key, sizeof(key));
	}
	ChunkLists cl;
	int most = 0;
	for (int r = 0; r <= nranks; ++r) cl.offset[r] = ctx->all_tiles_offset[r];
	for (int r = 0; r < nranks; ++r) most = std::max(most, cl.offset[r + 1] - cl.offset[r]);
	const int64_t chunk_pixels = chunk_bytes / 4;
	if (chunk_pixels < (int64_t)most * tile * tile) {
		rt_set_error("chunk of %lld bytes cannot hold %d tiles of %d x %d RGBA8 pixels", (long long)chunk_bytes, most, tile, tile);
		return RT_HIP_ERR_ARG;
	}
	const int64_t n = chunk_pixels * nranks;
	hipLaunchKernelGGL(k_unpack_chunks, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st,
	                   (const uint32_t*)d_chunks, chunk_pixels, rank_stride_bytes / 4, nranks, cl, ctx->d_all_tiles.as<int>(), w, h, tile,
	                   (uint32_t*)d_frame);
	RT_HIP_CHECK(hipGetLastError());
	return RT_HIP_OK;
}

int rt_launch_tiles_pack(rt_hip_ctx* ctx, bool pack, uint8_t* d_frame, int w, int h, int tile, const int* d_tile_list,
                         int ntiles, uint8_t* d_packed)
{
	const int64_t n = (int64_t)ntiles * tile * tile;
	if (n <= 0) return RT_HIP_OK;
	if (pack)
		hipLaunchKernelGGL(k_tiles_pack<true>, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, ctx->stream,
		                   (uint32_t*)d_frame, w, h, tile, d_tile_list, ntiles, (uint32_t*)d_packed);
	else
		hipLaunchKernelGGL(k_tiles_pack<false>, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, ctx->stream,
		                   (uint32_t*)d_frame, w, h, tile, d_tile_list, ntiles, (uint32_t*)d_packed);
	RT_HIP_CHECK(hipGetLastError());
	return RT_HIP_OK;
}

// Scene::draw (data/scene.cpp:40-56) from the G-buffer on: select -> processRays' ray-gen and
// closest hit -> raypass -> composite.  The hits are not grouped by material: that grouping only
// exists to batch GL material binds (ras_rasterizer.cpp:653-662).
int rt_live_frame(rt_hip_ctx* ctx, const float* d_pos_rgb, const float* d_nrm_rgba, int w, int h, int interlace, int phase,
                  uint8_t* d_raypass, uint8_t* d_rgba, int* out_rays)
{
	const int n = w * h;
	*out_rays = 0;
	RT_CHECK(ctx->d_in_a.reserve((size_t)n * 12));
	RT_CHECK(ctx->d_in_b.reserve((size_t)n * 12));
	RT_CHECK(ctx->d_ray_parent.reserve((size_t)n * 4));
	int count = 0;
	RT_CHECK(rt_gbuffer_select(ctx, d_nrm_rgba, d_pos_rgb, n, interlace, phase, ctx->d_in_a.as<float>(), ctx->d_in_b.as<float>(),
	                           ctx->d_ray_parent.as<int>(), &count));
	SceneView sv = rt_scene_view(ctx);
	if (count > 0) {
		RT_CHECK(ctx->d_org.reserve((size_t)count * 12));
		RT_CHECK(ctx->d_dir.reserve((size_t)count * 12));
		RT_CHECK(ctx->d_hits.reserve((size_t)count * sizeof(Hit)));
		RT_CHECK(rt_launch_reflect_raygen(ctx, ctx->d_in_a.as<float>(), ctx->d_in_b.as<float>(), count, ctx->d_org.as<float>(),
		                                  ctx->d_dir.as<float>()));
		RT_CHECK(rt_launch_closest(ctx, ctx->d_org.as<float>(), ctx->d_dir.as<float>(), count, ctx->d_hits.as<Hit>()));
		hipLaunchKernelGGL(k_live_raypass, dim3(nblocks(count, BLOCK)), dim3(BLOCK), 0, ctx->stream, sv, ctx->d_hits.as<Hit>(),
		                   ctx->d_ray_parent.as<int>(), count, d_raypass);
	}
	hipLaunchKernelGGL(k_live_composite, dim3(nblocks(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, sv, d_pos_rgb, d_nrm_rgba, n,
	                   d_raypass, d_rgba, ctx->frame_uv_count > 0 ? ctx->frame_uv[0] : (const float*)nullptr);
	RT_HIP_CHECK(hipGetLastError());
	RT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
	RT_CHECK(rt_collect_timing(ctx));
	*out_rays = count;
	return RT_HIP_OK;
}

// The same with the reference's two sizes (ras_rasterizer.cpp:39,167-173): rays from a pw x ph prepass G-buffer, their
// hits splatted into the pw x ph raypass image at the projection of the ray origin, the w x h frame composed from the
// full-size G-buffer with a bilinear fetch of the raypass image.
int rt_live_frame_scaled(rt_hip_ctx* ctx, const float* d_pre_pos, const float* d_pre_nrm, int pw, int ph, const float* d_pos_rgb,
                         const float* d_nrm_rgba, int w, int h, float fov, int interlace, int phase, uint8_t* d_raypass,
                         uint8_t* d_rgba, int* out_rays)
{
	const int n = pw * ph;
	*out_rays = 0;
	RT_CHECK(ctx->d_in_a.reserve((size_t)n * 12));
	RT_CHECK(ctx->d_in_b.reserve((size_t)n * 12));
	int count = 0;
	RT_CHECK(rt_gbuffer_select(ctx, d_pre_nrm, d_pre_pos, n, interlace, phase, ctx->d_in_a.as<float>(), ctx->d_in_b.as<float>(),
	                           nullptr, &count));
	SceneView sv = rt_scene_view(ctx);
	if (count > 0) {
		// projectionFromCamera (ras_rasterizer.cpp:140-153)
		const float nn = 1.0f, ff = 100.0f;
		const float scale = (float)(tan((double)fov) * nn);
		const float r = ((float)w) / h * scale, t = scale;
		LiveProj pr;
		pr.m00 = nn / r; pr.m11 = nn / t; pr.m22 = -(ff + nn) / (ff - nn); pr.m32 = -2 * ff * nn / (ff - nn);
		pr.pw = pw; pr.ph = ph;
		RT_CHECK(ctx->d_org.reserve((size_t)count * 12));
		RT_CHECK(ctx->d_dir.reserve((size_t)count * 12));
		RT_CHECK(ctx->d_hits.reserve((size_t)count * sizeof(Hit)));
		RT_CHECK(rt_launch_reflect_raygen(ctx, ctx->d_in_a.as<float>(), ctx->d_in_b.as<float>(), count, ctx->d_org.as<float>(),
		                                  ctx->d_dir.as<float>()));
		RT_CHECK(rt_launch_closest(ctx, ctx->d_org.as<float>(), ctx->d_dir.as<float>(), count, ctx->d_hits.as<Hit>()));
		hipLaunchKernelGGL(k_live_raypass_proj, dim3(nblocks(count, BLOCK)), dim3(BLOCK), 0, ctx->stream, sv, ctx->d_hits.as<Hit>(),
		                   (const float*)ctx->d_org.as<float>(), count, pr, d_raypass);
	}
	hipLaunchKernelGGL(k_live_composite_scaled, dim3(nblocks((int64_t)w * h, BLOCK)), dim3(BLOCK), 0, ctx->stream, sv, d_pos_rgb, d_nrm_rgba,
	                   w, h, (const uint8_t*)d_raypass, pw, ph, d_rgba, ctx->frame_uv_count > 0 ? ctx->frame_uv[0] : (const float*)nullptr);
	RT_HIP_CHECK(hipGetLastError());
	RT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
	RT_CHECK(rt_collect_timing(ctx));
	*out_rays = count;
	return RT_HIP_OK;
}

int rt_launch_shade_results(rt_hip_ctx* ctx, const rt_ray* d_rec, int n, const rt_result_offset* d_groups, int ngroups,
                            float* d_rgb, uint8_t* d_rgb8)
{
	if (n <= 0) return RT_HIP_OK;
	hipLaunchKernelGGL(k_shade_results, dim3(nblocks(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, rt_scene_view(ctx), d_rec, n,
	                   d_groups, ngroups, d_rgb, d_rgb8);
	RT_HIP_CHECK(hipGetLastError());
	return RT_HIP_OK;
}

int rt_launch_split_rays(rt_hip_ctx* ctx, const rt_ray* d_rays, int n, float* d_org, float* d_dir)
{
	if (n <= 0) return RT_HIP_OK;
	hipLaunchKernelGGL(k_split_rays, dim3(nblocks(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, d_rays, n, d_org, d_dir);
	RT_HIP_CHECK(hipGetLastError());
	return RT_HIP_OK;
}

int rt_launch_reflect_raygen(rt_hip_ctx* ctx, const float* d_pos, const float* d_nrm, int n, float* d_org,
                             float* d_dir)
{
	if (n <= 0) return RT_HIP_OK;
	hipLaunchKernelGGL(k_reflect_raygen, dim3(nblocks(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, d_pos, d_nrm, n, d_org, d_dir);
	RT_HIP_CHECK(hipGetLastError());
	return RT_HIP_OK;
}

int rt_launch_fill_records(rt_hip_ctx* ctx, const float* d_org, const float* d_dir, const Hit* d_hits, int n,
                           rt_ray* d_rays_inout, uint8_t* d_hit_flag, uint32_t* d_material)
{
	(void)d_org; (void)d_dir;
	if (n <= 0) return RT_HIP_OK;
	hipLaunchKernelGGL(k_fill_records, dim3(nblocks(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, rt_scene_view(ctx), d_hits, n,
	                   d_rays_inout, d_hit_flag, d_material);
	RT_HIP_CHECK(hipGetLastError());
	return RT_HIP_OK;
}

int rt_fill_sphere_records(rt_hip_ctx* ctx, const int* d_index, const float* d_pos, const float* d_nrm, int n,
                           rt_ray* d_rays, uint8_t* d_flag, uint32_t* d_material)
{
	if (n <= 0) return RT_HIP_OK;
	hipLaunchKernelGGL(k_fill_sphere_records, dim3(nblocks(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, d_index, d_pos, d_nrm,
	                   n, d_rays, d_flag, d_material);
	RT_HIP_CHECK(hipGetLastError());
	return RT_HIP_OK;
}

// Stable group-by-material of the hits (misses dropped): one 256-way multisplit per key byte.
int rt_group_by_material(rt_hip_ctx* ctx, const float* d_org, const float* d_dir, const Hit* d_hits, int n,
                         rt_ray* d_records, rt_result_offset* d_offsets, int* out_results, int* out_groups)
{
	*out_results = 0;
	*out_groups = 0;
	if (n <= 0) return RT_HIP_OK;
	hipStream_t st = ctx->stream;
	SceneView sv = rt_scene_view(ctx);
	const int nmat = (int)ctx->materials.size();
	if (nmat >= 0xFFFF) { rt_set_error("more than 65534 materials"); return RT_HIP_ERR_LIMIT; }
	RT_CHECK(ctx->d_keys.reserve((size_t)n * 4));
	RT_CHECK(ctx->d_rank.reserve((size_t)n * 4));
	RT_CHECK(ctx->d_in_a.reserve((size_t)n * 4));
	RT_CHECK(ctx->d_in_b.reserve((size_t)n * 4));
	RT_CHECK(ctx->d_counts.reserve(2 * 256 * 4));
	uint32_t* digit = ctx->d_keys.as<uint32_t>();
	uint32_t* dest = ctx->d_rank.as<uint32_t>();
	uint32_t* perm_a = ctx->d_in_a.as<uint32_t>();
	uint32_t* perm_b = ctx->d_in_b.as<uint32_t>();
	uint32_t* totals = ctx->d_counts.as<uint32_t>();
	const int g = nblocks(n, BLOCK);
	const bool two_pass = nmat >= 255;
	// pass 1: low byte (misses carry 0xFF / 0xFFFF and sort last)
	hipLaunchKernelGGL(k_material_digit, dim3(g), dim3(BLOCK), 0, st, sv, d_hits, n, 0, digit);
	RT_CHECK(rt_multisplit256(ctx, digit, n, dest, totals));
	uint32_t* perm = perm_a;
	if (two_pass) {
		hipLaunchKernelGGL(k_iota, dim3(g), dim3(BLOCK), 0, st, perm_b, n);
		hipLaunchKernelGGL(k_apply_dest, dim3(g), dim3(BLOCK), 0, st, perm_b, dest, n, perm_a);
	} else if (d_records) {
		// one pass: dest[i] is ray i's final slot; the records go out before the host has seen any count
		hipLaunchKernelGGL(k_emit_records_scatter, dim3(g), dim3(BLOCK), 0, st, sv, d_org, d_dir, d_hits, dest, n, d_records);
	}
	std::vector<uint32_t> h_lo(256), h_hi(256);
	RT_HIP_CHECK(hipMemcpyAsync(h_lo.data(), totals, 256 * 4, hipMemcpyDeviceToHost, st));
	if (two_pass) {
		hipLaunchKernelGGL(k_material_digit_perm, dim3(g), dim3(BLOCK), 0, st, sv, d_hits, perm_a, n, 8, digit);
		RT_CHECK(rt_multisplit256(ctx, digit, n, dest, totals + 256));
		hipLaunchKernelGGL(k_apply_dest, dim3(g), dim3(BLOCK), 0, st, perm_a, dest, n, perm_b);
		perm = perm_b;
		RT_HIP_CHECK(hipMemcpyAsync(h_hi.data(), totals + 256, 256 * 4, hipMemcpyDeviceToHost, st));
	}
	RT_HIP_CHECK(hipGetLastError());
	RT_HIP_CHECK(hipStreamSynchronize(st));
	int misses = two_pass ? (int)h_hi[255] : (int)h_lo[255];
	// With two passes the high-byte bucket 255 holds only misses as long as nmat < 0xFF00.
	int nres = n - misses;
	*out_results = nres;
	if (nres > 0 && two_pass) {
		hipLaunchKernelGGL(k_emit_records, dim3(nblocks(nres, BLOCK)), dim3(BLOCK), 0, st, sv, d_org, d_dir, d_hits, perm,
		                   nres, d_records);
		RT_HIP_CHECK(hipGetLastError());
	}
	// Cumulative end offsets for the materials that have hits (std::map holds only those).
	std::vector<rt_result_offset> off;
	if (!two_pass) {
		int run = 0;
		for (int m = 0; m < 255 && m < nmat; ++m)
			if (h_lo[m]) { run += (int)h_lo[m]; rt_result_offset o; o.material = (uint32_t)m; o.end = run; off.push_back(o); }
	} else {
		// counts per full key need the sorted keys; recompute from the permutation on the host side
		// would cost a download -- instead count materials with a tiny second histogram on device:
		// (rare path: >= 255 materials) download the per-ray material ids.
		std::vector<Hit> hh((size_t)n);
		RT_HIP_CHECK(hipMemcpy(hh.data(), d_hits, (size_t)n * sizeof(Hit), hipMemcpyDeviceToHost));
		std::vector<TriPos> tp((size_t)ctx->num_tris);
		RT_HIP_CHECK(hipMemcpy(tp.data(), ctx->d_tri_pos.p, tp.size() * sizeof(TriPos), hipMemcpyDeviceToHost));
		std::vector<int> cnt((size_t)nmat, 0);
		for (int i = 0; i < n; ++i)
			if (hh[i].prim != RT_MISS) {
				int m;
				memcpy(&m, &tp[hh[i].prim].p0.w, 4);
				if (m >= 0 && m < nmat) ++cnt[m];
			}
		int run = 0;
		for (int m = 0; m < nmat; ++m)
			if (cnt[m]) { run += cnt[m]; rt_result_offset o; o.material = (uint32_t)m; o.end = run; off.push_back(o); }
	}
	*out_groups = (int)off.size();
	if (!off.empty() && d_offsets)
		RT_HIP_CHECK(hipMemcpyAsync(d_offsets, off.data(), off.size() * sizeof(rt_result_offset), hipMemcpyHostToDevice, st));
	RT_HIP_CHECK(hipStreamSynchronize(st));
	return RT_HIP_OK;
}

int rt_gbuffer_select(rt_hip_ctx* ctx, const float* d_nrm_rgba, const float* d_pos_rgb, int n, int interlace,
                      int phase, float* d_pos_out, float* d_nrm_out, int* d_index_out, int* out_count)
{
	*out_count = 0;
	if (n <= 0) return RT_HIP_OK;
	hipStream_t st = ctx->stream;
	RT_CHECK(ctx->d_keys.reserve((size_t)n * 4));
	RT_CHECK(ctx->d_rank.reserve((size_t)n * 4));
	RT_CHECK(ctx->d_counts.reserve(2 * 256 * 4));
	uint32_t* digit = ctx->d_keys.as<uint32_t>();
	uint32_t* dest = ctx->d_rank.as<uint32_t>();
	uint32_t* totals = ctx->d_counts.as<uint32_t>();
	const int g = nblocks(n, BLOCK);
	hipLaunchKernelGGL(k_select_digit, dim3(g), dim3(BLOCK), 0, st, d_nrm_rgba, n, interlace, phase, digit);
	RT_CHECK(rt_multisplit256(ctx, digit, n, dest, totals));
	hipLaunchKernelGGL(k_select_scatter, dim3(g), dim3(BLOCK), 0, st, d_nrm_rgba, d_pos_rgb, digit, dest, n, d_pos_out,
	                   d_nrm_out, d_index_out);
	RT_HIP_CHECK(hipGetLastError());
	uint32_t cnt = 0;
	RT_HIP_CHECK(hipMemcpyAsync(&cnt, totals, 4, hipMemcpyDeviceToHost, st));
	RT_HIP_CHECK(hipStreamSynchronize(st));
	*out_count = (int)cnt;
	return RT_HIP_OK;
}

int rt_render_gbuffer(rt_hip_ctx* ctx, float fov, int w, int h, float* d_nrm_rgba, float* d_pos_rgb, float* d_uv)
{
	SceneView sv = rt_scene_view(ctx);
	float tanfov = tanf(fov);  // Camera::setFOV (data/camera.cpp:41-44)
	size_t lds = rt_trace_lds(sv, TRACE_BLOCK);
	hipLaunchKernelGGL(k_primary_gbuffer, dim3(rt_trace_blocks(ctx, tile8_count(w, h), TRACE_BLOCK)), dim3(TRACE_BLOCK), lds, ctx->stream,
	                   sv, tanfov, w, h, d_nrm_rgba, d_pos_rgb, d_uv);
	RT_HIP_CHECK(hipGetLastError());
	return RT_HIP_OK;
}

// Resolve the hits of a ray queue into HitInfo (triangles via the trace kernels, spheres inline).
// d_count != nullptr: nq is the queue's capacity and the count stays on the device (the caller
// accounts for the rays in the stats once it has read the counts back).
static int trace_queue(rt_hip_ctx* ctx, const SceneView& sv, const float* q_org, const float* q_dir, int nq,
                       HitInfo* d_hi, const int* d_count = nullptr)
{
	if (nq <= 0) return RT_HIP_OK;
	if (sv.num_spheres > 0 && sv.num_tris == 0) {
		hipLaunchKernelGGL(k_queue_sphere_hits, dim3(nblocks(nq, BLOCK)), dim3(BLOCK), 0, ctx->stream, sv, q_dir, nq, d_count, d_hi);
		RT_HIP_CHECK(hipGetLastError());
		if (!d_count) ctx->stats.rays_closest += nq;
		return RT_HIP_OK;
	}
	RT_CHECK(ctx->d_hits.reserve((size_t)nq * sizeof(Hit)));
	RT_CHECK(rt_launch_closest(ctx, q_org, q_dir, nq, ctx->d_hits.as<Hit>(), d_count));
	hipLaunchKernelGGL(k_resolve, dim3(nblocks(nq, BLOCK)), dim3(BLOCK), 0, ctx->stream, sv, ctx->d_hits.as<Hit>(), nq, d_count, d_hi);
	RT_HIP_CHECK(hipGetLastError());
	return RT_HIP_OK;
}

// ---- the frame without host round trips -----------------------------------------------------------
// Every level's buffers are sized by what the level can hold at most (2 children per node, capped
// at FRAME_LEVEL_CAP), the kernels are launched over that bound and read the actual counts on the
// device, and the host reads all counts once, after the last kernel.  If a level would have needed
// more than its capacity, *overflowed is set and the caller renders the frame again on the
// synchronous path, which sizes every level exactly.
constexpr int64_t FRAME_LEVEL_CAP = (int64_t)1 << 23;      // nodes per level
constexpr int64_t FRAME_TOTAL_CAP = (int64_t)3 << 23;      // nodes per frame (about 2.5 GB of level arrays)

// The rank's own tiles as a device list (cached per frame geometry); used when the tile size lets a
// tile be cut into 8x8 blocks.
static int owned_tile_list(rt_hip_ctx* ctx, int w, int h, int tile, int rank, int nranks, const int** d_list, int* count)
{
	const int key[5] = {w, h, tile, rank, nranks};
	if (memcmp(key, ctx->tile_key, sizeof(key)) != 0 || !ctx->d_tile_list.p) {
		const int tiles_x = (w + tile - 1) / tile, tiles_y = (h + tile - 1) / tile;
		std::vector<int> ids;
		for (int ty = 0; ty < tiles_y; ++ty)
			for (int tx = 0; tx < tiles_x; ++tx)
				if ((tx + ty) % nranks == rank) ids.push_back(ty * tiles_x + tx);
		RT_CHECK(ctx->d_tile_list.reserve(std::max<size_t>(ids.size(), 1) * sizeof(int)));
		if (!ids.empty()) RT_HIP_CHECK(hipMemcpy(ctx->d_tile_list.p, ids.data(), ids.size() * sizeof(int), hipMemcpyHostToDevice));
		memcpy(ctx->tile_key, key, sizeof(key));
		ctx->tile_count = (int)ids.size();
	}
	*d_list = ctx->d_tile_list.as<int>();
	*count = ctx->tile_count;
	return RT_HIP_OK;
}

// Lane `lane` of the frame pipeline: its stream (created on first use; lane 0 = the context's) and its level arrays.
FrameLane rt_frame_lane(rt_hip_ctx* ctx, int lane)
{
	FrameLane l;
	l.index = lane;
	if (lane == 0) {
		l.stream = ctx->stream;
		l.nodes_a = &ctx->d_nodes_a; l.nodes_b = &ctx->d_nodes_b; l.gate = &ctx->d_gate; l.node_inc = &ctx->d_node_inc; l.node_uv = &ctx->d_node_uv;
		l.rays_q = &ctx->d_rays_q; l.ray_parent = &ctx->d_ray_parent; l.records = &ctx->d_records; l.hits = &ctx->d_hits; l.occ = &ctx->d_occ;
		l.child_slot = &ctx->d_child_slot; l.fcount = &ctx->d_fcount; l.group_boxes = &ctx->d_group_boxes; l.prefix = &ctx->d_prefix;
		l.fcount_clean = &ctx->fcount_clean; l.fcount_clean_n = &ctx->fcount_clean_n;
		return l;
	}
	if (!ctx->lane_stream[lane] && hipStreamCreateWithFlags(&ctx->lane_stream[lane], hipStreamNonBlocking) != hipSuccess) ctx->lane_stream[lane] = nullptr;
	FrameBufs& b = ctx->lane_bufs[lane];
	l.stream = ctx->lane_stream[lane] ? ctx->lane_stream[lane] : ctx->stream;  // (no stream to be had: the lane shares the context's; still correct)
	l.nodes_a = &b.nodes_a; l.nodes_b = &b.nodes_b; l.gate = &b.gate; l.node_inc = &b.node_inc; l.node_uv = &b.node_uv;
	l.rays_q = &b.rays_q; l.ray_parent = &b.ray_parent; l.records = &b.records; l.hits = &b.hits; l.occ = &b.occ;
	l.child_slot = &b.child_slot; l.fcount = &b.fcount; l.group_boxes = &b.group_boxes; l.prefix = &b.prefix;
	l.fcount_clean = &ctx->lane_fcount_clean[lane]; l.fcount_clean_n = &ctx->lane_fcount_clean_n[lane];
	return l;
}

// Queue every kernel of the frame and the read-back of its counters into pf->h_fc; nothing is waited for.
static int frame_async_enqueue(rt_hip_ctx* ctx, FrameSet fs, int w, int h, int tile, int rank, int nranks, PendingFrame* pf,
                               bool* declined, int lane_index = 0)
{
	*declined = true;
	const FrameLane ln = rt_frame_lane(ctx, lane_index);
	hipStream_t st = ln.stream;
	pf->stream = st;
	SceneView sv = rt_scene_view(ctx);
	// the HBM part of the traversal stacks: a set of columns per lane (frames of different lanes trace at the same time)
	if (sv.spill && sv.stack_depth > sv.lds_depth) sv.spill += (size_t)lane_index * (size_t)(sv.stack_depth - sv.lds_depth) * sv.spill_stride;
	const int bounces = ctx->cfg.bounces;
	const int packed = fs.packed;
	fs.tile_list = nullptr; fs.ntiles = 0; fs.packed = 0;
	fs.per_frame = tile8_count(w, h);
	if ((nranks > 1 || packed) && tile % 8 == 0) {   // (a one-rank communicator takes the same path: it owns every tile)
		RT_CHECK(owned_tile_list(ctx, w, h, tile, rank, nranks, &fs.tile_list, &fs.ntiles));
		fs.per_frame = fs.ntiles * tile * tile;
		fs.packed = packed;
	}
	if (packed && !fs.packed) { rt_set_error("packed tile output needs a tile size that is a multiple of 8"); return RT_HIP_ERR_ARG; }
	if ((int64_t)fs.per_frame * fs.nframes > FRAME_LEVEL_CAP) return RT_HIP_OK;
	const int n0 = fs.per_frame * fs.nframes;
	pf->n0 = n0; pf->bounces = bounces; pf->nframes = fs.nframes; pf->num_lights = sv.num_lights; pf->rays = 0;
	pf->trace_mark = ctx->trace_events_used;
	if (n0 == 0) { *declined = false; pf->complete = true; pf->can_overflow = false; return RT_HIP_OK; }  // this rank owns no tile
	std::vector<int64_t> cap(bounces), off(bounces + 1, 0);
	for (int L = 0; L < bounces; ++L) {
		cap[L] = L == 0 ? n0 : std::min<int64_t>(2 * cap[L - 1], FRAME_LEVEL_CAP);
		off[L + 1] = off[L] + cap[L];
	}
	if (n0 > FRAME_LEVEL_CAP || off[bounces] > FRAME_TOTAL_CAP) return RT_HIP_OK;  // too big to size by bounds
	*declined = false;
	pf->can_overflow = false;  // a level whose bound is the cap rather than twice its parent's can come out larger than its arrays
	for (int L = 1; L < bounces; ++L) pf->can_overflow = pf->can_overflow || cap[L] < 2 * cap[L - 1];
	const int64_t maxcap = *std::max_element(cap.begin(), cap.end());
	RT_CHECK(ln.nodes_a->reserve((size_t)off[bounces] * sizeof(NodeGeo)));
	RT_CHECK(ln.nodes_b->reserve((size_t)off[bounces] * sizeof(NodeShade)));
	RT_CHECK(ln.gate->reserve((size_t)off[bounces] * sizeof(int)));
	RT_CHECK(ln.node_inc->reserve((size_t)off[bounces] * 12));  // incident directions, 3 floats per node
	float* const inc_base = ln.node_inc->as<float>();
	if (sv.tex) RT_CHECK(ln.node_uv->reserve((size_t)off[bounces] * 8));  // textured scenes: a texcoord per node
	float* const uv_base = sv.tex ? ln.node_uv->as<float>() : nullptr;
	auto level_uv = [&](int L) -> float* { return uv_base ? uv_base + 2 * (size_t)off[L] : nullptr; };
	RT_CHECK(ln.rays_q->reserve((size_t)maxcap * 24));
	RT_CHECK(ln.ray_parent->reserve((size_t)maxcap * 4));
	RT_CHECK(ln.records->reserve((size_t)maxcap * sizeof(HitInfo)));
	RT_CHECK(ln.hits->reserve((size_t)maxcap * sizeof(Hit)));
	// frame counters: [0] active level-0 nodes, [1] overflow flag,
	// [8 + 8L + 0] nodes of level L, [.. + 1] reflection rays of level L, [.. + 2] refraction rays
	const int nfc = 8 + 8 * (bounces + 1);
	RT_CHECK(ln.fcount->reserve((size_t)nfc * 4));  // this driver's own: nothing else writes to it
	int* fc = ln.fcount->as<int>();
	float* q_org = ln.rays_q->as<float>();
	float* q_dir = q_org + (size_t)maxcap * 3;
	int* q_parent = ln.ray_parent->as<int>();
	HitInfo* d_hi = ln.records->as<HitInfo>();
	const bool shadows = !ctx->cfg.use_photon_map && ctx->cfg.do_shadows && sv.num_lights > 0;

	// Order of a level: spawn the reflection rays; launch A = reflection queue + the shadow rays of
	// the first half of the lights; children; spawn the refraction rays (gated by the reflection
	// hits, Q9); launch B = refraction queue + the shadow rays of the other lights; children.  The
	// light term needs all shadow rays of the level, so it is evaluated in k_combine at the end.
	const bool fuse = shadows && !(sv.num_spheres > 0 && sv.num_tris == 0);
	RT_CHECK(ln.occ->reserve((size_t)off[bounces] * 4));  // shadow masks and child flags
	RT_CHECK(ln.child_slot->reserve((size_t)off[bounces] * 8));  // [node][branch] -> child index (k_combine's pull form)
	// The shared descent of every (group of 64 nodes, light) of a level (k_shadow_prefix): tuning.shadow_prefix = the most
	// siblings a record may carry (they start on the rays' LDS stacks), 0 = every ray descends from the root by itself.
	const int max_push = std::min(ctx->tuning.shadow_prefix, PREFIX_MAX);
	// The walks are one kernel whose length is that of its longest walk (~1 us per node: a chain of dependent fetches),
	// whatever the size of the frame: it pays once the level's shadow rays fill the machine several times over.  Below
	// that (an 1/4 or 1/8 share of a 1080p frame) the shadow launches are as long as their slowest wave and the walks cost
	// more than they save: 0.430 -> 0.454 ms for a quarter of the 1080p frame, 0.592 -> 0.574 ms for a half.
	const bool use_prefix = shadows && !(sv.num_spheres > 0 && sv.num_tris == 0) && sv.use_bvh && sv.num_tris > 0 && max_push > 0 &&
	                        (int64_t)n0 * sv.num_lights >= (ctx->tuning.shadow_prefix_min_rays < 0 ? (int64_t)6 * ctx->resident_threads : (int64_t)ctx->tuning.shadow_prefix_min_rays);
	// Level 0 only: its groups are the frame's 8x8 tiles.  Deeper levels are compactions of their parents, 64 consecutive
	// nodes of them lie too far apart for a shared descent (measured: the level-1 launch 235 -> 232 us for 35 us of walks).
	const int64_t groups0 = ((int64_t)n0 + 63) / 64;
	if (use_prefix) {
		RT_CHECK(ln.group_boxes->reserve((size_t)groups0 * sizeof(GroupBox)));
		RT_CHECK(ln.prefix->reserve((size_t)groups0 * sv.num_lights * PREFIX_INTS * sizeof(int)));
	}
	auto level_prefix = [&](int L) -> int* { return (use_prefix && L == 0) ? ln.prefix->as<int>() : nullptr; };
	auto prefix_launch = [&](int L, int n) -> int {
		if (!use_prefix || L != 0) return RT_HIP_OK;
		// (the groups' boxes: k_init_spawn)
		hipEvent_t t0, t1;  // (profile mode: its time is booked with the shadow launches it serves)
		RT_CHECK(rt_next_trace_events(ctx, 3, &t0, &t1));
		hipExtLaunchKernelGGL(k_shadow_prefix, dim3(nblocks(groups0 * sv.num_lights, BLOCK)), dim3(BLOCK), 0, st, t0, t1, 0u, sv,
		                      (const GroupBox*)ln.group_boxes->as<GroupBox>(), n, (const int*)nullptr, sv.num_lights, ctx->cfg.shadow_t_max,
		                      max_push, ctx->tuning.shadow_prefix_steps, level_prefix(0));
		RT_HIP_CHECK(hipGetLastError());
		return RT_HIP_OK;
	};
	auto shadow_launch = [&](const NodeGeo* geo, int n, const int* n_ptr, int lo, int hi, unsigned int* occl,
	                         const int* q_count, const int* prefix) -> int {
		FusedQueue fq{};
		unsigned long long* cen = rt_census_ptr(ctx, 1);
		if (q_count) {
			fq.org = q_org; fq.dir = q_dir; fq.count_ptr = q_count; fq.capacity = n;
			fq.hits = ln.hits->as<Hit>(); fq.census = rt_census_ptr(ctx, 0);
		}
		if (hi <= lo && !q_count) return RT_HIP_OK;
		const size_t lds = rt_trace_lds(sv, TRACE_BLOCK);
		const int64_t npad = prefix ? ((int64_t)n + 63) / 64 * 64 : n;
		const int blocks = rt_trace_blocks(ctx, std::max<int64_t>(npad * (hi - lo), n), TRACE_BLOCK);
		// timed (profile mode) through the dispatch itself: no event markers between the frame's kernels
		hipEvent_t t0, t1;
		RT_CHECK(rt_next_trace_events(ctx, 1, &t0, &t1));
		if (cen)
			hipExtLaunchKernelGGL(k_shadow<true>, dim3(blocks), dim3(TRACE_BLOCK), (uint32_t)lds, st, t0, t1, 0u, sv, geo, n, n_ptr, lo, hi,
			                      ctx->cfg.shadow_t_max, occl, cen, fq, prefix);
		else
			hipExtLaunchKernelGGL(k_shadow<false>, dim3(blocks), dim3(TRACE_BLOCK), (uint32_t)lds, st, t0, t1, 0u, sv, geo, n, n_ptr, lo, hi,
			                      ctx->cfg.shadow_t_max, occl, (unsigned long long*)nullptr, fq, prefix);
		RT_HIP_CHECK(hipGetLastError());
		return RT_HIP_OK;
	};
	pf->shadows = shadows;
	pf->complete = false;
	RT_HIP_CHECK(hipEventRecord(pf->ev0, st));
	// The counters are zeroed at the END of a frame, behind its read-back, for the frame that follows: the
	// first thing a frame queues is then its first kernel, not a fill and the host time after it.
	if ((*ln.fcount_clean) != fc || (*ln.fcount_clean_n) < nfc) RT_HIP_CHECK(hipMemsetAsync(fc, 0, (size_t)nfc * 4, st));
	(*ln.fcount_clean) = nullptr;
	for (int L = 0; L < bounces; ++L) {
		const int n = (int)cap[L];
		int* lc = fc + 8 + 8 * L;                      // this level's counters
		const int* n_ptr = L == 0 ? nullptr : lc;      // level 0's count is known: n0
		const bool recurse = (L + 1 < bounces);
		NodeGeo* geo = ln.nodes_a->as<NodeGeo>() + off[L];
		NodeShade* shade = ln.nodes_b->as<NodeShade>() + off[L];
		int* gate = ln.gate->as<int>() + off[L];
		unsigned int* occl = ln.occ->as<unsigned int>() + off[L];
		if (L == 0) {
			// nodes, gate, cleared shade / shadow mask and the reflection queue of level 0 in one pass
			hipLaunchKernelGGL(k_init_spawn, dim3(nblocks(n0, CBLOCK)), dim3(CBLOCK), 0, st, sv, fs, w, h, n0,
			                   tile, rank, nranks, recurse ? 1 : 0, geo, (float*)nullptr, gate, occl, q_org, q_dir, q_parent, lc + 1, fc + 0, level_uv(0),
			                   use_prefix ? ln.group_boxes->as<GroupBox>() : (GroupBox*)nullptr);
		} else if (recurse) {
			// deeper levels: k_make_children has cleared the shadow masks of the nodes it created.  The last
			// level needs nothing here: no rays to spawn, the light term is k_combine's, NodeShade stays unwritten.
			hipLaunchKernelGGL(k_light_spawn_reflect, dim3(nblocks(n, CBLOCK)), dim3(CBLOCK), 0, st, sv, geo,
			                   (const float*)(inc_base + 3 * (size_t)off[L]), shade, n, n_ptr,
			                   (const unsigned int*)nullptr, (const float*)nullptr, ctx->cfg.do_shadows, 1, recurse ? 1 : 0, gate,
			                   q_org, q_dir, q_parent, lc + 1);
		}
		RT_HIP_CHECK(hipGetLastError());
		const int nl = sv.num_lights, nl_a = (recurse && fuse) ? nl / 2 : nl;
		RT_CHECK(prefix_launch(L, n));
		const int* prefix = level_prefix(L);
		if (!recurse) {
			if (shadows) RT_CHECK(shadow_launch(geo, n, n_ptr, 0, nl, occl, nullptr, prefix));
			break;
		}
		NodeGeo* child_geo = ln.nodes_a->as<NodeGeo>() + off[L + 1];
		unsigned int* child_occl = ln.occ->as<unsigned int>() + off[L + 1];
		int* child_count = lc + 8;                     // = level L+1's node count
		const int child_cap = (int)cap[L + 1];
		const Hit* raw = nullptr;                      // fused: children are made straight from the raw hits
		if (fuse) {
			RT_CHECK(shadow_launch(geo, n, n_ptr, 0, nl_a, occl, lc + 1, prefix));
			raw = ln.hits->as<Hit>();
		} else {
			if (shadows) RT_CHECK(shadow_launch(geo, n, n_ptr, 0, nl, occl, nullptr, prefix));
			RT_CHECK(trace_queue(ctx, sv, q_org, q_dir, n, d_hi, lc + 1));
		}
		// texcoords of the children: from the queue's raw hits (trace_queue leaves them in d_hits for triangles)
		const Hit* tc_hits = (sv.num_spheres > 0 && sv.num_tris == 0) ? nullptr : ln.hits->as<Hit>();
		hipLaunchKernelGGL(k_make_children, dim3(nblocks(n, CBLOCK)), dim3(CBLOCK), 0, st, sv, d_hi, raw, q_dir, q_parent, n,
		                   lc + 1, 0, gate, child_geo, inc_base + 3 * (size_t)off[L + 1], child_occl, child_count, child_cap, fc + 1, tc_hits,
		                   level_uv(L + 1), occl, ln.child_slot->as<int>() + 2 * off[L]);
		// refraction, gated after the reflection hits are known (Q9)
		hipLaunchKernelGGL(k_spawn_refract, dim3(nblocks(n, CBLOCK)), dim3(CBLOCK), 0, st, sv, geo,
		                   L == 0 ? (const float*)nullptr : (const float*)(inc_base + 3 * (size_t)off[L]), n, n_ptr, gate, q_org, q_dir, q_parent, lc + 2);
		RT_HIP_CHECK(hipGetLastError());
		if (fuse) RT_CHECK(shadow_launch(geo, n, n_ptr, nl_a, nl, occl, lc + 2, prefix));
		else RT_CHECK(trace_queue(ctx, sv, q_org, q_dir, n, d_hi, lc + 2));
		hipLaunchKernelGGL(k_make_children, dim3(nblocks(n, CBLOCK)), dim3(CBLOCK), 0, st, sv, d_hi, raw, q_dir, q_parent, n,
		                   lc + 2, 1, gate, child_geo, inc_base + 3 * (size_t)off[L + 1], child_occl, child_count, child_cap, fc + 1, tc_hits,
		                   level_uv(L + 1), occl, ln.child_slot->as<int>() + 2 * off[L]);
		RT_HIP_CHECK(hipGetLastError());
	}
	// The last level is folded into the combine of the level above it (pull form: a launch, a 5 us gap and a round trip of
	// the children's colours through NodeShade less); deeper frames chain the remaining levels bottom-up as before.
	for (int L = bounces >= 2 ? bounces - 2 : 0; L >= 0; --L) {
		const int* n_ptr = L == 0 ? nullptr : fc + 8 + 8 * L;
		NodeGeo* geo = ln.nodes_a->as<NodeGeo>() + off[L];
		NodeShade* shade = ln.nodes_b->as<NodeShade>() + off[L];
		NodeShade* pshade = L > 0 ? ln.nodes_b->as<NodeShade>() + off[L - 1] : nullptr;
		const unsigned int* occl = ln.occ->as<unsigned int>() + off[L];
		ChildLevel pull = {nullptr, nullptr, nullptr, nullptr};
		if (bounces >= 2 && L == bounces - 2) {
			pull.slot = ln.child_slot->as<int>() + 2 * off[L];
			pull.geo = ln.nodes_a->as<NodeGeo>() + off[L + 1];
			pull.occl = ln.occ->as<unsigned int>() + off[L + 1];
			pull.uv = level_uv(L + 1);
		}
		hipLaunchKernelGGL(k_combine, dim3(nblocks((int)cap[L], BLOCK)), dim3(BLOCK), 0, st, sv, geo, shade, (int)cap[L], n_ptr, L,
		                   pshade, fs, fs.packed ? fs.per_frame : w * h, (float*)nullptr, 1, occl, level_uv(L), pull);
		RT_HIP_CHECK(hipGetLastError());
	}
	RT_HIP_CHECK(hipMemcpyAsync(pf->h_fc, fc, (size_t)nfc * 4, hipMemcpyDeviceToHost, st));  // 512 B pinned per slot; nfc <= 80 ints
	RT_HIP_CHECK(hipEventRecord(pf->ev1, st));
	hipLaunchKernelGGL(k_zero_ints, dim3(1), dim3(128), 0, st, fc, nfc);  // behind the event: for the next frame
	RT_HIP_CHECK(hipGetLastError());
	(*ln.fcount_clean) = fc; (*ln.fcount_clean_n) = nfc;
	pf->trace_mark = ctx->trace_events_used;
	return RT_HIP_OK;
}

// Wait for the frame's counters and account for it.  *overflowed: a level needed more than its bound, the frame has to be
// rendered again with host-sized levels.
static int frame_async_finish(rt_hip_ctx* ctx, PendingFrame* pf, bool* overflowed)
{
	*overflowed = false;
	if (pf->complete) { ctx->last_frame_rays = pf->rays; return RT_HIP_OK; }
	RT_HIP_CHECK(hipEventSynchronize(pf->ev1));
	const int* h_fc = pf->h_fc;
	if (h_fc[1]) {
		*overflowed = true;
		// the frame is rendered again; this attempt's launch timings are folded in as they are (rare path)
		RT_CHECK(rt_collect_timing(ctx, pf->trace_mark));
		return RT_HIP_OK;
	}
	int64_t rays_any = 0, rays_closest = 0;
	int levels = 0;
	for (int L = 0; L < pf->bounces; ++L) {
		const int nL = L == 0 ? pf->n0 : h_fc[8 + 8 * L];
		if (nL == 0) break;
		levels = L + 1;
		if (pf->shadows) rays_any += (int64_t)(L == 0 ? h_fc[0] : nL) * pf->num_lights;
		rays_closest += (int64_t)h_fc[8 + 8 * L + 1] + h_fc[8 + 8 * L + 2];
	}
	ctx->stats.rays_any += rays_any;
	ctx->stats.rays_closest += rays_closest;
	float ms = 0;
	RT_HIP_CHECK(hipEventElapsedTime(&ms, pf->ev0, pf->ev1));
	ctx->stats.frame_ms += ms;
	ctx->stats.frames += pf->nframes;
	pf->rays = rays_any + rays_closest;
	ctx->last_frame_rays = pf->rays;
	ctx->last_count = levels;
	RT_CHECK(rt_collect_timing(ctx, pf->trace_mark));
	return RT_HIP_OK;
}

static int render_frame_async(rt_hip_ctx* ctx, FrameSet fs, int w, int h, int tile, int rank, int nranks,
                              bool* overflowed, bool* declined)
{
	*overflowed = false;
	if (ctx->pending_count > 0) { rt_set_error("%d frame(s) are enqueued: rt_hip_frame_finish() them first", ctx->pending_count); return RT_HIP_ERR_STATE; }
	PendingFrame* pf = &ctx->pending[0];
	RT_CHECK(frame_async_enqueue(ctx, fs, w, h, tile, rank, nranks, pf, declined));
	if (*declined) return RT_HIP_OK;
	// The hook runs exactly once per rt_render_frames call (it may hold a collective, which every rank must enter the same
	// number of times): here, ahead of the host's wait, when no level of this frame can overflow its bound; otherwise after
	// the wait, or after the frame has been redone.
	const bool early = ctx->after_frame_queued && !pf->can_overflow;
	if (early) RT_CHECK(ctx->after_frame_queued());
	RT_CHECK(frame_async_finish(ctx, pf, overflowed));
	if (ctx->after_frame_queued && !early && !*overflowed) RT_CHECK(ctx->after_frame_queued());
	return RT_HIP_OK;
}


static int render_frame_sync(rt_hip_ctx* ctx, const float* d_pos_rgb, const float* d_nrm_rgba, const float* d_uv, int w, int h,
                             int tile, int rank, int nranks, uint8_t* d_rgba);

int rt_render_frames(rt_hip_ctx* ctx, const float* const* d_pos_rgb, const float* const* d_nrm_rgba, int nframes, int w,
                     int h, int tile, int rank, int nranks, uint8_t* const* d_rgba, int packed)
{
	ctx->last_frame_rays = 0;
	if (w <= 0 || h <= 0 || nframes <= 0 || ctx->cfg.bounces <= 0) return ctx->after_frame_queued ? ctx->after_frame_queued() : RT_HIP_OK;
	if (nframes > RT_MAX_BATCH) { rt_set_error("more than %d frames in a batch", RT_MAX_BATCH); return RT_HIP_ERR_LIMIT; }
	if ((int64_t)((w + 7) & ~7) * ((h + 7) & ~7) > (int64_t)1 << 26) { rt_set_error("frame larger than 2^26 texels"); return RT_HIP_ERR_LIMIT; }
	if (ctx->cfg.bounces > 8) { rt_set_error("bounces > 8 not supported"); return RT_HIP_ERR_LIMIT; }
	if ((int)ctx->lights.size() > 32) { rt_set_error("more than 32 lights"); return RT_HIP_ERR_LIMIT; }
	// The photon gather takes its point count on the host, so photon-mapped frames stay on the
	// synchronous path.  tuning.frame_sync forces it (tests compare the two).
	const int force_sync = ctx->tuning.frame_sync;
	// (more lights than the mask word has room for beside the child flags: the synchronous driver as well)
	if (!ctx->cfg.use_photon_map && !force_sync && (int)ctx->lights.size() <= FRAME_ASYNC_MAX_LIGHTS) {
		FrameSet fs;
		memset(&fs, 0, sizeof(fs));
		fs.nframes = nframes;
		fs.packed = packed;
		for (int f = 0; f < nframes; ++f) {
			fs.pos[f] = d_pos_rgb[f]; fs.nrm[f] = d_nrm_rgba[f]; fs.rgba[f] = d_rgba[f];
			fs.uv[f] = f < ctx->frame_uv_count ? ctx->frame_uv[f] : nullptr;
		}
		bool overflowed = false, declined = false;
		RT_CHECK(render_frame_async(ctx, fs, w, h, tile, rank, nranks, &overflowed, &declined));
		if (!overflowed && !declined) return RT_HIP_OK;
		// exceptional, and an order of magnitude slower for split frames: say so once per context
		if (!ctx->warned_fallback) {
			ctx->warned_fallback = true;
			fprintf(stderr, "librt_hip: frame batch (%d x %dx%d, rank %d/%d) %s; rendering frame by frame with host-sized levels\n",
			        nframes, w, h, rank, nranks, overflowed ? "overflowed a level's bound" : "is too large to size by bounds");
		}
	}
	// frame by frame, every level sized on the host (full-size image, then packed if asked for)
	int64_t rays = 0;
	for (int f = 0; f < nframes; ++f) {
		uint8_t* target = d_rgba[f];
		if (packed) {
			RT_CHECK(ctx->d_offsets.reserve((size_t)w * h * 4));
			target = ctx->d_offsets.as<uint8_t>();
		}
		RT_CHECK(render_frame_sync(ctx, d_pos_rgb[f], d_nrm_rgba[f], f < ctx->frame_uv_count ? ctx->frame_uv[f] : nullptr, w, h, tile,
		                           rank, nranks, target));
		rays += ctx->last_frame_rays;
		if (packed) {
			const int* list; int nt;
			RT_CHECK(owned_tile_list(ctx, w, h, tile, rank, nranks, &list, &nt));
			RT_CHECK(rt_launch_tiles_pack(ctx, true, target, w, h, tile, list, nt, d_rgba[f]));
			RT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
		}
	}
	ctx->last_frame_rays = rays;
	if (ctx->after_frame_queued) RT_CHECK(ctx->after_frame_queued());
	return RT_HIP_OK;
}

int rt_render_frame(rt_hip_ctx* ctx, const float* d_pos_rgb, const float* d_nrm_rgba, int w, int h, int tile,
                    int rank, int nranks, uint8_t* d_rgba)
{
	return rt_render_frames(ctx, &d_pos_rgb, &d_nrm_rgba, 1, w, h, tile, rank, nranks, &d_rgba);
}

static int render_frame_sync(rt_hip_ctx* ctx, const float* d_pos_rgb, const float* d_nrm_rgba, const float* d_uv, int w, int h,
                             int tile, int rank, int nranks, uint8_t* d_rgba)
{
	hipStream_t st = ctx->stream;
	SceneView sv = rt_scene_view(ctx);
	const int bounces = ctx->cfg.bounces;
	const int npix = tile8_count(w, h);  // level-0 nodes: the frame padded to whole 8x8 tiles
	FrameSet fs1;                        // k_combine's output description: this one frame
	memset(&fs1, 0, sizeof(fs1));
	fs1.nframes = 1; fs1.per_frame = npix; fs1.rgba[0] = d_rgba;
	const bool spheres = sv.num_spheres > 0 && sv.num_tris == 0;

	// level arrays: level L can hold at most 2^L * npix nodes, but only what was spawned is used;
	// capacity is grown on demand after each level's count is known.
	std::vector<int> count(bounces + 1, 0);
	std::vector<size_t> geo_off(bounces + 1, 0);
	count[0] = npix;
	// reserve level 0 now, deeper levels as they appear (one DevBuf, offsets in nodes)
	auto ensure_nodes = [&](size_t nodes) -> int {
		RT_CHECK(ctx->d_nodes_a.reserve(nodes * sizeof(NodeGeo)));
		RT_CHECK(ctx->d_nodes_b.reserve(nodes * sizeof(NodeShade)));
		RT_CHECK(ctx->d_gate.reserve(nodes * sizeof(int)));
		RT_CHECK(ctx->d_node_inc.reserve(nodes * 12));
		if (sv.tex) RT_CHECK(ctx->d_node_uv.reserve(nodes * 8));  // textured scenes: a texcoord per node
		return RT_HIP_OK;
	};
	// (re)fetched at every use: reserve() may move the buffer
	auto level_uv = [&](size_t node_off) -> float* { return sv.tex ? ctx->d_node_uv.as<float>() + 2 * node_off : nullptr; };
	auto level_inc = [&](size_t node_off) -> float* { return ctx->d_node_inc.as<float>() + 3 * node_off; };
	// worst case for the next level is 2 children per node; reserve level 0 + level 1 bound up front
	RT_CHECK(ensure_nodes((size_t)npix * (bounces > 1 ? 3 : 1)));
	RT_CHECK(ctx->d_qcount.reserve(64));
	int* qcount = ctx->d_qcount.as<int>();  // [0] ray queue count, [1] child count
	RT_CHECK(ctx->d_occ.reserve((size_t)npix * 4 * 2));

	RT_HIP_CHECK(hipEventRecord(ctx->ev0, st));
	RT_HIP_CHECK(hipMemsetAsync(qcount, 0, 64, st));
	hipLaunchKernelGGL(k_init_nodes, dim3(nblocks(npix, CBLOCK)), dim3(CBLOCK), 0, st, d_pos_rgb, d_nrm_rgba, w, h, npix, tile, rank,
	                   nranks, sv.num_materials, ctx->d_nodes_a.as<NodeGeo>(), level_inc(0), qcount + 8, d_uv, level_uv(0));
	RT_HIP_CHECK(hipGetLastError());
	int active0 = 0;
	RT_HIP_CHECK(hipMemcpyAsync(&active0, qcount + 8, 4, hipMemcpyDeviceToHost, st));
	RT_HIP_CHECK(hipStreamSynchronize(st));

	int64_t rays = 0;
	int levels = 0;
	for (int L = 0; L < bounces; ++L) {
		const int n = count[L];
		if (n == 0) break;
		levels = L + 1;
		const bool recurse = (L + 1 < bounces);
		// (re)fetch base pointers: reserve() may have moved the buffers
		NodeGeo* geo = ctx->d_nodes_a.as<NodeGeo>() + geo_off[L];
		NodeShade* shade = ctx->d_nodes_b.as<NodeShade>() + geo_off[L];
		int* gate = ctx->d_gate.as<int>() + geo_off[L];

		const float* photon_rgb = nullptr;
		unsigned int* occl = nullptr;
		if (ctx->cfg.use_photon_map) {
			RT_CHECK(ctx->d_org.reserve((size_t)n * 12));
			RT_CHECK(ctx->d_dir.reserve((size_t)n * 12));
			RT_CHECK(ctx->d_accum.reserve((size_t)n * 12));
			hipLaunchKernelGGL(k_node_points, dim3(nblocks(n, BLOCK)), dim3(BLOCK), 0, st, geo, n, ctx->d_org.as<float>(),
			                   ctx->d_dir.as<float>());
			RT_CHECK(rt_photon_gather(ctx, ctx->d_org.as<float>(), ctx->d_dir.as<float>(), n, ctx->cfg.photon_estimate,
			                          ctx->cfg.photon_radius, ctx->d_accum.as<float>(), L == 0));  // level 0: 8 x 8 tiles of the frame
			photon_rgb = ctx->d_accum.as<float>();
		} else if (ctx->cfg.do_shadows && sv.num_lights > 0) {
			RT_CHECK(ctx->d_occ.reserve((size_t)n * 4));
			occl = ctx->d_occ.as<unsigned int>();
			RT_HIP_CHECK(hipMemsetAsync(occl, 0, (size_t)n * 4, st));
			int64_t total = (int64_t)n * sv.num_lights;
			size_t lds = rt_trace_lds(sv, TRACE_BLOCK);
			RT_CHECK(rt_begin_trace_timing(ctx));
			unsigned long long* cen = rt_census_ptr(ctx, 1);
			if (cen)
				hipLaunchKernelGGL(k_shadow<true>, dim3(rt_trace_blocks(ctx, total, TRACE_BLOCK)), dim3(TRACE_BLOCK), lds, st, sv, geo, n,
				                   (const int*)nullptr, 0, sv.num_lights, ctx->cfg.shadow_t_max, occl, cen, FusedQueue{}, (const int*)nullptr);
			else
				hipLaunchKernelGGL(k_shadow<false>, dim3(rt_trace_blocks(ctx, total, TRACE_BLOCK)), dim3(TRACE_BLOCK), lds, st, sv, geo, n,
				                   (const int*)nullptr, 0, sv.num_lights, ctx->cfg.shadow_t_max, occl, (unsigned long long*)nullptr,
				                   FusedQueue{}, (const int*)nullptr);
			RT_HIP_CHECK(hipGetLastError());
			RT_CHECK(rt_end_trace_timing(ctx, 1));
			int64_t issued = (int64_t)(L == 0 ? active0 : n) * sv.num_lights;  // only active nodes trace
			rays += issued;
			ctx->stats.rays_any += issued;
		}

		// queue buffers sized for one ray per node
		RT_CHECK(ctx->d_rays_q.reserve((size_t)n * 24));
		RT_CHECK(ctx->d_ray_parent.reserve((size_t)n * 4));
		RT_CHECK(ctx->d_records.reserve((size_t)n * sizeof(HitInfo)));
		float* q_org = ctx->d_rays_q.as<float>();
		float* q_dir = q_org + (size_t)n * 3;
		int* q_parent = ctx->d_ray_parent.as<int>();
		HitInfo* d_hi = ctx->d_records.as<HitInfo>();

		RT_HIP_CHECK(hipMemsetAsync(qcount, 0, 16, st));
		hipLaunchKernelGGL(k_light_spawn_reflect, dim3(nblocks(n, CBLOCK)), dim3(CBLOCK), 0, st, sv, geo,
		                   (const float*)level_inc(geo_off[L]), shade, n, (const int*)nullptr, occl, photon_rgb, ctx->cfg.do_shadows, 0, recurse ? 1 : 0, gate, q_org, q_dir, q_parent, qcount);
		RT_HIP_CHECK(hipGetLastError());
		if (!recurse) break;

		// children of this level are appended after it
		geo_off[L + 1] = geo_off[L] + (size_t)n;
		int h_counts[4] = {0, 0, 0, 0};
		RT_HIP_CHECK(hipMemcpyAsync(h_counts, qcount, 16, hipMemcpyDeviceToHost, st));
		RT_HIP_CHECK(hipStreamSynchronize(st));
		int nq = h_counts[0];
		RT_CHECK(ensure_nodes(geo_off[L + 1] + (size_t)2 * n));  // both branches can at most add n each
		geo = ctx->d_nodes_a.as<NodeGeo>() + geo_off[L];
		gate = ctx->d_gate.as<int>() + geo_off[L];
		NodeGeo* child_geo = ctx->d_nodes_a.as<NodeGeo>() + geo_off[L + 1];
		const Hit* tc_hits = spheres ? nullptr : ctx->d_hits.as<Hit>();  // trace_queue leaves the raw hits there
		if (nq > 0) {
			RT_CHECK(trace_queue(ctx, sv, q_org, q_dir, nq, d_hi));
			rays += nq;
			tc_hits = spheres ? nullptr : ctx->d_hits.as<Hit>();
			hipLaunchKernelGGL(k_make_children, dim3(nblocks(nq, CBLOCK)), dim3(CBLOCK), 0, st, sv, d_hi, (const Hit*)nullptr, q_dir,
			                   q_parent, nq, (const int*)nullptr, 0, gate, child_geo, level_inc(geo_off[L + 1]), (unsigned int*)nullptr, qcount + 1, 0x7fffffff,
			                   qcount + 2, tc_hits, level_uv(geo_off[L + 1]), (unsigned int*)nullptr, (int*)nullptr);
			RT_HIP_CHECK(hipGetLastError());
		}
		// refraction, gated after the reflection hits are known (Q9)
		RT_HIP_CHECK(hipMemsetAsync(qcount, 0, 4, st));
		hipLaunchKernelGGL(k_spawn_refract, dim3(nblocks(n, CBLOCK)), dim3(CBLOCK), 0, st, sv, geo,
		                   (const float*)level_inc(geo_off[L]), n, (const int*)nullptr, gate, q_org, q_dir, q_parent, qcount);
		RT_HIP_CHECK(hipGetLastError());
		RT_HIP_CHECK(hipMemcpyAsync(h_counts, qcount, 16, hipMemcpyDeviceToHost, st));
		RT_HIP_CHECK(hipStreamSynchronize(st));
		nq = h_counts[0];
		if (nq > 0) {
			RT_CHECK(trace_queue(ctx, sv, q_org, q_dir, nq, d_hi));
			rays += nq;
			tc_hits = spheres ? nullptr : ctx->d_hits.as<Hit>();
			hipLaunchKernelGGL(k_make_children, dim3(nblocks(nq, CBLOCK)), dim3(CBLOCK), 0, st, sv, d_hi, (const Hit*)nullptr, q_dir,
			                   q_parent, nq, (const int*)nullptr, 1, gate, child_geo, level_inc(geo_off[L + 1]), (unsigned int*)nullptr, qcount + 1, 0x7fffffff,
			                   qcount + 2, tc_hits, level_uv(geo_off[L + 1]), (unsigned int*)nullptr, (int*)nullptr);
			RT_HIP_CHECK(hipGetLastError());
			RT_HIP_CHECK(hipMemcpyAsync(h_counts, qcount, 16, hipMemcpyDeviceToHost, st));
			RT_HIP_CHECK(hipStreamSynchronize(st));
		}
		count[L + 1] = h_counts[1];
	}

	// shadow-ray count: every active node of every shaded level issues num_lights rays
	// bottom-up combine
	for (int L = levels - 1; L >= 0; --L) {
		const int n = count[L];
		NodeGeo* geo = ctx->d_nodes_a.as<NodeGeo>() + geo_off[L];
		NodeShade* shade = ctx->d_nodes_b.as<NodeShade>() + geo_off[L];
		NodeShade* pshade = L > 0 ? ctx->d_nodes_b.as<NodeShade>() + geo_off[L - 1] : nullptr;
		hipLaunchKernelGGL(k_combine, dim3(nblocks(n, BLOCK)), dim3(BLOCK), 0, st, sv, geo, shade, n, (const int*)nullptr, L,
		                   pshade, fs1, w * h, (float*)nullptr, 0, (const unsigned int*)nullptr, level_uv(geo_off[L]), ChildLevel{nullptr, nullptr, nullptr, nullptr});
		RT_HIP_CHECK(hipGetLastError());
	}
	RT_HIP_CHECK(hipEventRecord(ctx->ev1, st));
	RT_HIP_CHECK(hipEventSynchronize(ctx->ev1));
	float ms = 0;
	RT_HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
	ctx->stats.frame_ms += ms;
	ctx->stats.frames += 1;
	ctx->last_frame_rays = rays;
	// stash per-level counts for the census
	ctx->last_count = levels;
	RT_CHECK(rt_collect_timing(ctx));
	return RT_HIP_OK;
}

// ---- the frame pipeline ------------------------------------------------------------------------------
// rt_hip_frame_enqueue_device queues a whole frame and returns without waiting for it; rt_hip_frame_finish waits for the
// OLDEST enqueued frame and accounts for it.  With one frame enqueued ahead, the host's per-frame work (reading the
// counters, returning to the caller, queueing the next dozen launches: ~55 us) runs under the previous frame's kernels.
int rt_launch_decode_g16(rt_hip_ctx* ctx, const uint16_t* d_nrm16, const uint16_t* d_pos16, int n, float* d_nrm_rgba, float* d_pos_rgb)
{
	if (n <= 0) return RT_HIP_OK;
	hipLaunchKernelGGL(k_decode_g16, dim3(nblocks(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, (const ushort4*)d_nrm16, d_pos16, n,
	                   (float4*)d_nrm_rgba, d_pos_rgb);
	RT_HIP_CHECK(hipGetLastError());
	return RT_HIP_OK;
}

// The ring slot the next enqueued frame takes.  At most max(2, tuning.frame_streams) frames are unfinished at a time.
int rt_pipeline_slot(rt_hip_ctx* ctx, int* slot)
{
	const int limit = std::max(2, ctx->tuning.frame_streams);
	if (ctx->pending_count >= limit) { rt_set_error("%d frames are already enqueued: rt_hip_frame_finish() one first", ctx->pending_count); return RT_HIP_ERR_STATE; }
	*slot = (ctx->pending_head + ctx->pending_count) % RT_MAX_PENDING;
	return RT_HIP_OK;
}

// `nframes` frames of one size as ONE pending entry of the pipeline: a single pass of launches shades all of them (the level
// arrays hold the frames' nodes one after the other), rt_frame_finish completes and accounts for the whole batch.
int rt_frames_enqueue(rt_hip_ctx* ctx, const float* const* d_pos_rgb, const float* const* d_nrm_rgba, int nframes, int w, int h,
                      uint8_t* const* d_rgba, int tile, int rank, int nranks, int packed)
{
	int slot = 0;
	RT_CHECK(rt_pipeline_slot(ctx, &slot));
	if (nframes < 1 || nframes > RT_MAX_BATCH) { rt_set_error("1..%d frames in a batch", RT_MAX_BATCH); return RT_HIP_ERR_LIMIT; }
	if (ctx->cfg.bounces <= 0 || ctx->cfg.bounces > 8) { rt_set_error("bounces must be 1..8"); return RT_HIP_ERR_LIMIT; }
	if ((int)ctx->lights.size() > 32) { rt_set_error("more than 32 lights"); return RT_HIP_ERR_LIMIT; }
	if ((int64_t)((w + 7) & ~7) * ((h + 7) & ~7) > (int64_t)1 << 26) { rt_set_error("frame larger than 2^26 texels"); return RT_HIP_ERR_LIMIT; }
	PendingFrame* pf = &ctx->pending[slot];
	FrameSet fs;
	memset(&fs, 0, sizeof(fs));
	fs.nframes = nframes;
	fs.packed = packed;
	for (int f = 0; f < nframes; ++f) {
		fs.pos[f] = d_pos_rgb[f]; fs.nrm[f] = d_nrm_rgba[f]; fs.rgba[f] = d_rgba[f];
		fs.uv[f] = f < ctx->frame_uv_count ? ctx->frame_uv[f] : nullptr;
	}
	pf->fs = fs; pf->w = w; pf->h = h; pf->tile = tile; pf->rank = rank; pf->nranks = nranks; pf->packed = packed;
	pf->comm = false;
	pf->stream = ctx->stream;
	bool declined = true;
	const bool can_async = !ctx->cfg.use_photon_map && !ctx->tuning.frame_sync && (int)ctx->lights.size() <= FRAME_ASYNC_MAX_LIGHTS;
	if (can_async) {
		// Lanes beyond the first serve the usual frame only (shadow rays on triangles, the closest-hit queues fused into the
		// shadow launches): the other forms queue trace launches of their own on the context's stream.
		const bool spheres = ctx->cfg.geometry == RT_HIP_SPHERES;
		const int lanes = (ctx->cfg.do_shadows && !ctx->lights.empty() && !spheres) ? ctx->tuning.frame_streams : 1;
		const int lane = (int)(ctx->frame_seq % lanes);
		if (lane != 0) {
			// whatever the caller has queued on the context's stream (the G-buffer's producer, an upload's event) comes first
			const FrameLane ln = rt_frame_lane(ctx, lane);
			if (ln.stream != ctx->stream) {
				if (!ctx->ev_order) RT_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_order, hipEventDisableTiming));
				RT_HIP_CHECK(hipEventRecord(ctx->ev_order, ctx->stream));
				RT_HIP_CHECK(hipStreamWaitEvent(ln.stream, ctx->ev_order, 0));
			}
		}
		RT_CHECK(frame_async_enqueue(ctx, fs, w, h, tile, rank, nranks, pf, &declined, lane));
	}
	if (declined) {
		// photon-mapped frames, more than 30 lights, frames too large to size by bounds: rendered here and now
		if (ctx->pending_count > 0) { rt_set_error("this frame needs the synchronous driver: finish the enqueued frame first"); return RT_HIP_ERR_STATE; }
		if (packed) { rt_set_error("a packed frame that needs the synchronous driver cannot be enqueued"); return RT_HIP_ERR_STATE; }
		int64_t rays = 0;
		for (int f = 0; f < nframes; ++f) {
			RT_CHECK(render_frame_sync(ctx, fs.pos[f], fs.nrm[f], fs.uv[f], w, h, tile, rank, nranks, fs.rgba[f]));
			rays += ctx->last_frame_rays;
		}
		pf->complete = true;
		pf->rays = rays;
		pf->stream = ctx->stream;
	}
	pf->used = true;
	++ctx->pending_count;
	++ctx->frame_seq;
	return RT_HIP_OK;
}

int rt_frame_enqueue(rt_hip_ctx* ctx, const float* d_pos_rgb, const float* d_nrm_rgba, int w, int h, uint8_t* d_rgba, int tile, int rank,
                     int nranks, int packed)
{
	return rt_frames_enqueue(ctx, &d_pos_rgb, &d_nrm_rgba, 1, w, h, &d_rgba, tile, rank, nranks, packed);
}

int rt_frame_finish(rt_hip_ctx* ctx, int64_t* out_rays)
{
	if (out_rays) *out_rays = 0;
	if (ctx->pending_count == 0) { rt_set_error("no frame is enqueued"); return RT_HIP_ERR_STATE; }
	PendingFrame* pf = &ctx->pending[ctx->pending_head];
	bool overflowed = false;
	RT_CHECK(frame_async_finish(ctx, pf, &overflowed));
	ctx->frame_redone = overflowed;
	if (overflowed) {
		// exceptional: the frame again, on the driver that sizes every level on the host (behind whatever is still queued)
		int64_t rays = 0;
		for (int f = 0; f < pf->fs.nframes; ++f) {
			RT_CHECK(render_frame_sync(ctx, pf->fs.pos[f], pf->fs.nrm[f], pf->fs.uv[f], pf->w, pf->h, pf->tile, pf->rank, pf->nranks, pf->fs.rgba[f]));
			rays += ctx->last_frame_rays;
		}
		pf->rays = rays;
	}
	if (out_rays) *out_rays = pf->rays;
	ctx->last_frame_rays = pf->rays;
	pf->used = false;
	ctx->pending_head = (ctx->pending_head + 1) % RT_MAX_PENDING;
	--ctx->pending_count;
	return RT_HIP_OK;
}
