// raytracer_amd/csrc/k_trace.hip -- intersection kernels: brute force (geometry through the
// scalar/LDS path), LBVH traversal with a per-lane LDS stack, spheres.  FP32 VALU only; MFMA is
// not used because nothing here is a dense contraction.
//
// Results restate, bit for bit, the reference's closest-hit / any-hit arithmetic:
//   closest  rt_raytracer.cpp:315-370 (naiveTrace) / rt_accel_array.cpp:75-137
//   occlude  rt_accel_array.cpp:45-73
//   spheres  rt_accel_spheres.cpp:38-104
#include "rt_ctx.h"
#include "rt_trace_dev.h"

namespace {

constexpr int BLOCK = 256;
constexpr int BVH_BLOCK = 64;

// ---- brute force ------------------------------------------------------------------------------
// One ray per lane; the triangle index is wave-uniform, so the 48-byte records come in through
// scalar loads (constant across the wave) and feed the VALU as SGPR operands.
template <int MODE>  // 0 closest, 1 any (intersect tests), 2 AccelArray::occlude
__global__ __launch_bounds__(BLOCK) void k_brute(const TriIsect* __restrict__ tris, int num_tris,
                                                 const float* __restrict__ org, const float* __restrict__ dir,
                                                 int n, const int* __restrict__ n_ptr, TraceParams tp,
                                                 Hit* __restrict__ hits, uint8_t* __restrict__ occ)
{
	if (n_ptr) { int m = *n_ptr; n = m < n ? m : n; }
	int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n) return;
	f3 O = ld3(org + 3 * (size_t)i), D = ld3(dir + 3 * (size_t)i);
	Hit best;
	best.prim = RT_MISS; best.t = tp.t_max; best.u = 0; best.v = 0;
	bool any = false;
	for (int k = 0; k < num_tris; ++k) {
		float4 a = tris[k].a, b = tris[k].b, c = tris[k].c;
		f3 v0 = mk3(a.x, a.y, a.z), e1 = mk3(b.x, b.y, b.z), e2 = mk3(c.x, c.y, c.z);
		if (MODE == 0) {
			mt_closest(O, D, v0, e1, e2, __float_as_uint(a.w), tp, best);
		} else if (MODE == 1) {
			if (mt_any(O, D, v0, e1, e2, tp)) { any = true; break; }
		} else {
			if (mt_occlude_ref(O, D, v0, e1, e2, tp.eps_det)) { any = true; break; }
		}
	}
	if (MODE == 0) {
		if (best.prim == RT_MISS) best.t = __builtin_inff();
		hits[i] = best;
	} else {
		occ[i] = any ? 1 : 0;
	}
}

// ---- LBVH traversal ---------------------------------------------------------------------------
template <bool ANY, bool CENSUS>
__global__ __launch_bounds__(BVH_BLOCK) void k_bvh(SceneView sv, const float* __restrict__ org,
                                                   const float* __restrict__ dir, int n, const int* __restrict__ n_ptr,
                                                   Hit* __restrict__ hits,
                                                   uint8_t* __restrict__ occ, uint32_t* __restrict__ nv,
                                                   uint32_t* __restrict__ nt, unsigned long long* census)
{
	// Capped grid (see grid_for_each); a thread's overflow-stack column in HBM is its global thread id.
	extern __shared__ __attribute__((aligned(16))) int smem[];
	lds_int* stack = RT_LDS_STACK(smem) + threadIdx.x;
	if (n_ptr) { int m = *n_ptr; n = m < n ? m : n; }  // count on the device, n = capacity (sync-free frame path)
	uint32_t cn = 0, ct = 0, rays = 0;
	grid_for_each(sv, n, [&](int64_t i) {
		f3 O = ld3(org + 3 * (size_t)i), D = ld3(dir + 3 * (size_t)i);
		Hit best;
		uint32_t a = 0, b = 0;
		bool any = bvh_trace<ANY, CENSUS>(sv, O, D, best, stack, BVH_BLOCK, a, b);
		if (CENSUS && nv) { nv[i] = a; nt[i] = b; }
		cn += a; ct += b; ++rays;
		if (ANY) { if (occ) occ[i] = any ? 1 : 0; }
		else if (hits) hits[i] = best;
	});
	if (CENSUS && census) census_add(census, cn, ct, rays);
}

// ---- spheres ----------------------------------------------------------------------------------
// AccelSpheres::intersect (rt_accel_spheres.cpp:58-104): Ro = direction (reference quirk Q5), the
// FIRST sphere in list order that passes wins (Q6).
__global__ __launch_bounds__(BLOCK) void k_sphere_intersect(const SphereRec* __restrict__ sph, int ns,
                                                            const float* __restrict__ dir, int n,
                                                            int* __restrict__ index, float* __restrict__ pos,
                                                            float* __restrict__ nrm)
{
	int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n) return;
	f3 Rd = ld3(dir + 3 * (size_t)i);
	int found = -1;
	f3 X = mk3(0, 0, 0), N = mk3(0, 0, 0);
	for (int k = 0; k < ns; ++k) {
		if (sphere_test_ref(sph[k], Rd, Rd, X, N)) { found = k; break; }
	}
	index[i] = found;
	if (found >= 0) { st3(pos + 3 * (size_t)i, X); st3(nrm + 3 * (size_t)i, N); }
}

// AccelSpheres::occlude (rt_accel_spheres.cpp:38-56): discriminant of the infinite line.
__global__ __launch_bounds__(BLOCK) void k_sphere_occlude(const SphereRec* __restrict__ sph, int ns,
                                                          const float* __restrict__ org,
                                                          const float* __restrict__ dir, int n,
                                                          uint8_t* __restrict__ occ)
{
	int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i >= n) return;
	f3 Ro = ld3(org + 3 * (size_t)i), Rd = ld3(dir + 3 * (size_t)i);
	bool any = false;
	for (int k = 0; k < ns; ++k) {
		if (sphere_occlude_ref(sph[k], Ro, Rd)) { any = true; break; }
	}
	occ[i] = any ? 1 : 0;
}

inline int nblocks(int n, int b) { return (n + b - 1) / b; }

}  // namespace

int rt_launch_closest(rt_hip_ctx* ctx, const float* d_org, const float* d_dir, int n, Hit* d_hits, const int* d_count)
{
	if (n <= 0) return RT_HIP_OK;
	SceneView sv = rt_scene_view(ctx);
	RT_CHECK(rt_begin_trace_timing(ctx));
	unsigned long long* cen = rt_census_ptr(ctx, 0);
	if (sv.use_bvh && cen) {
		size_t lds = rt_trace_lds(sv, BVH_BLOCK);
		hipLaunchKernelGGL((k_bvh<false, true>), dim3(rt_trace_blocks(ctx, n, BVH_BLOCK)), dim3(BVH_BLOCK), lds, ctx->stream, sv,
		                   d_org, d_dir, n, d_count, d_hits, (uint8_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, cen);
	} else if (sv.use_bvh) {
		size_t lds = rt_trace_lds(sv, BVH_BLOCK);
		hipLaunchKernelGGL((k_bvh<false, false>), dim3(rt_trace_blocks(ctx, n, BVH_BLOCK)), dim3(BVH_BLOCK), lds, ctx->stream, sv,
		                   d_org, d_dir, n, d_count, d_hits, (uint8_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
		                   (unsigned long long*)nullptr);
	} else {
		hipLaunchKernelGGL((k_brute<0>), dim3(nblocks(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, sv.tris, sv.num_tris,
		                   d_org, d_dir, n, d_count, sv.tp, d_hits, (uint8_t*)nullptr);
	}
	RT_HIP_CHECK(hipGetLastError());
	RT_CHECK(rt_end_trace_timing(ctx, 0));
	if (!d_count) ctx->stats.rays_closest += n;  // with a device-side count the caller accounts for the rays
	return RT_HIP_OK;
}

int rt_launch_any(rt_hip_ctx* ctx, const float* d_org, const float* d_dir, int n, uint8_t* d_occ)
{
	const int* d_count = nullptr;
	if (n <= 0) return RT_HIP_OK;
	SceneView sv = rt_scene_view(ctx);
	RT_CHECK(rt_begin_trace_timing(ctx));
	unsigned long long* cen = rt_census_ptr(ctx, 1);
	if (sv.use_bvh && cen) {
		size_t lds = rt_trace_lds(sv, BVH_BLOCK);
		hipLaunchKernelGGL((k_bvh<true, true>), dim3(rt_trace_blocks(ctx, n, BVH_BLOCK)), dim3(BVH_BLOCK), lds, ctx->stream, sv,
		                   d_org, d_dir, n, d_count, (Hit*)nullptr, d_occ, (uint32_t*)nullptr, (uint32_t*)nullptr, cen);
	} else if (sv.use_bvh) {
		size_t lds = rt_trace_lds(sv, BVH_BLOCK);
		hipLaunchKernelGGL((k_bvh<true, false>), dim3(rt_trace_blocks(ctx, n, BVH_BLOCK)), dim3(BVH_BLOCK), lds, ctx->stream, sv,
		                   d_org, d_dir, n, d_count, (Hit*)nullptr, d_occ, (uint32_t*)nullptr, (uint32_t*)nullptr,
		                   (unsigned long long*)nullptr);
	} else {
		hipLaunchKernelGGL((k_brute<1>), dim3(nblocks(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, sv.tris, sv.num_tris,
		                   d_org, d_dir, n, (const int*)nullptr, sv.tp, (Hit*)nullptr, d_occ);
	}
	RT_HIP_CHECK(hipGetLastError());
	RT_CHECK(rt_end_trace_timing(ctx, 1));
	ctx->stats.rays_any += n;
	return RT_HIP_OK;
}

// AccelArray::occlude has no t test, so no BVH pruning applies: every triangle whose plane the
// LINE crosses inside the triangle occludes.  It is evaluated exhaustively (API-parity path, not
// a hot one: the shading loop uses rt_launch_any).
int rt_launch_occlude_ref(rt_hip_ctx* ctx, const float* d_org, const float* d_dir, int n, uint8_t* d_occ)
{
	if (n <= 0) return RT_HIP_OK;
	SceneView sv = rt_scene_view(ctx);
	hipLaunchKernelGGL((k_brute<2>), dim3(nblocks(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, sv.tris, sv.num_tris,
	                   d_org, d_dir, n, (const int*)nullptr, sv.tp, (Hit*)nullptr, d_occ);
	RT_HIP_CHECK(hipGetLastError());
	return RT_HIP_OK;
}

int rt_launch_census(rt_hip_ctx* ctx, const float* d_org, const float* d_dir, int n, int any_hit, uint32_t* d_nv,
                     uint32_t* d_nt)
{
	if (n <= 0) return RT_HIP_OK;
	SceneView sv = rt_scene_view(ctx);
	if (!sv.use_bvh) {
		rt_set_error("census needs the LBVH (accel = brute force)");
		return RT_HIP_ERR_STATE;
	}
	const int* d_count = nullptr;
	size_t lds = rt_trace_lds(sv, BVH_BLOCK);
	if (any_hit)
		hipLaunchKernelGGL((k_bvh<true, true>), dim3(rt_trace_blocks(ctx, n, BVH_BLOCK)), dim3(BVH_BLOCK), lds, ctx->stream, sv,
		                   d_org, d_dir, n, d_count, (Hit*)nullptr, (uint8_t*)nullptr, d_nv, d_nt, (unsigned long long*)nullptr);
	else
		hipLaunchKernelGGL((k_bvh<false, true>), dim3(rt_trace_blocks(ctx, n, BVH_BLOCK)), dim3(BVH_BLOCK), lds, ctx->stream, sv,
		                   d_org, d_dir, n, d_count, (Hit*)nullptr, (uint8_t*)nullptr, d_nv, d_nt, (unsigned long long*)nullptr);
	RT_HIP_CHECK(hipGetLastError());
	return RT_HIP_OK;
}

int rt_launch_sphere_intersect(rt_hip_ctx* ctx, const float* d_org, const float* d_dir, int n, int* d_index,
                               float* d_pos, float* d_nrm)
{
	(void)d_org;  // the reference replaces the origin by the direction (rt_accel_spheres.cpp:68)
	if (n <= 0) return RT_HIP_OK;
	hipLaunchKernelGGL(k_sphere_intersect, dim3(nblocks(n, BLOCK)), dim3(BLOCK), 0, ctx->stream,
	                   ctx->d_spheres.as<SphereRec>(), (int)ctx->spheres.size(), d_dir, n, d_index, d_pos, d_nrm);
	RT_HIP_CHECK(hipGetLastError());
	return RT_HIP_OK;
}

int rt_launch_sphere_occlude(rt_hip_ctx* ctx, const float* d_org, const float* d_dir, int n, uint8_t* d_occ)
{
	if (n <= 0) return RT_HIP_OK;
	hipLaunchKernelGGL(k_sphere_occlude, dim3(nblocks(n, BLOCK)), dim3(BLOCK), 0, ctx->stream,
	                   ctx->d_spheres.as<SphereRec>(), (int)ctx->spheres.size(), d_org, d_dir, n, d_occ);
	RT_HIP_CHECK(hipGetLastError());
	return RT_HIP_OK;
}
