// raytracer_amd/csrc/rt_trace_dev.h -- the inlined traversal core (used by the batch trace
// kernels, the shading kernels and the photon emitter) and the sphere tests.
#pragma once

#include "rt_ctx.h"

// Traverse the LBVH for one ray.  `stack` points at this lane's column of an LDS array laid out
// [lds_depth][block_threads] (so that every ds_read/ds_write of a wave touches 64 consecutive dwords
// -> conflict free), `stride` = threads per block.  Only the first sv.lds_depth entries of a lane's
// stack live in LDS (4 KB per wave at the default depth 16, so the full 32 waves/CU stay resident); deeper
// entries -- rare -- go to this thread's column (spill_column) of the HBM array sv.spill[..][sv.spill_stride].
//   ANY = false: closest hit into `best` (prim = RT_MISS / t = +inf on a miss); returns hit?
//   ANY = true : first accepted hit terminates; returns occluded?
// The LDS part is addressed through an address_space(3) pointer so that push/pop are ds_write /
// ds_read, never flat accesses.
typedef __attribute__((address_space(3))) int lds_int;
#define RT_LDS_STACK(smem) ((lds_int*)(smem))

// This thread's column of the overflow array, formed where it is needed (the rare path) instead of being carried
// through the loop as a pointer: held to 64 VGPRs, k_shadow otherwise spills that pointer to scratch at entry.
__device__ __forceinline__ size_t spill_column(int stride) { return (size_t)blockIdx.x * (size_t)stride + threadIdx.x; }

// The stack pointer is the LDS address of the next free entry (`top`, advancing by `stride` ints), `lim` the first
// address past the lane's LDS part: push and pop are one LDS instruction at an address that is already in a register.
// The overflow part is reached through a WAVE-uniform test (no lane of the wave is that deep: the common case by far),
// so that the usual push / pop has no second level of exec handling.
__device__ __forceinline__ void stack_push(const SceneView& sv, lds_int* top, lds_int* lim, int stride, int v)
{
	if (__builtin_amdgcn_ballot_w64(top >= lim) == 0) { *top = v; return; }
	if (top < lim) *top = v;
	else sv.spill[(size_t)((int)(top - lim) / stride) * sv.spill_stride + spill_column(stride)] = v;
}
__device__ __forceinline__ int stack_pop(const SceneView& sv, const lds_int* top, const lds_int* lim, int stride)
{
	if (__builtin_amdgcn_ballot_w64(top >= lim) == 0) return *top;
	int v;
	if (top < lim) v = *top;
	else v = sv.spill[(size_t)((int)(top - lim) / stride) * sv.spill_stride + spill_column(stride)];
	return v;
}

// ---- launch shape of the trace kernels ----------------------------------------------------------
// Grid-stride loops over a grid capped at tuning.trace_oversub (4) x the device's resident-thread capacity.
// The cap bounds the spill stack (one column per launched thread); the oversubscription leaves the
// balancing of uneven rays to the hardware dispatcher.  Measured on MI355X (atrium-262k, shadow
// rays): 1x 0.765 ms, 2x 0.657 ms, 4x 0.586 ms, uncapped 0.594 ms; a software queue (one returning
// atomic per wave per 64 rays) was atomic-bound at 1.08 ms.
// Above the cap the grid is sized so that every thread makes the same number of rounds (rt_trace_blocks).
// XCD-aware block order: workgroups are dealt round robin to the 8 XCDs (block b runs on XCD
// b % 8), each with an L2 of its own.  Renumbering b -> (b % 8) * ceil(nb / 8) + b / 8 hands every
// XCD one contiguous eighth of each grid-sized slice of the items, so that its L2 holds the part of
// the BVH under one region of the frame rather than a thin sample of all of it.
__device__ __forceinline__ int64_t xcd_block_index(int chunk)
{
	// groups of 8 * chunk consecutive block numbers; inside a group XCD x takes numbers
	// [x * chunk, (x + 1) * chunk).  A bijection on [0, nb) when nb is a multiple of 8 * chunk;
	// chunk == 0 keeps the hardware order.
	const int64_t b = blockIdx.x;
	if (chunk <= 0) return b;
	const int64_t q = b >> 3, x = b & 7;
	return (q / chunk) * (8 * chunk) + x * chunk + (q % chunk);
}

template <typename F>
__device__ __forceinline__ void grid_for_each(const SceneView& sv, int64_t total, F&& body)
{
	const int64_t gsize = (int64_t)gridDim.x * blockDim.x;
	for (int64_t i = xcd_block_index(sv.xcd_chunk) * blockDim.x + threadIdx.x; i < total; i += gsize) body(i);
}

// Control state of a traversal lives in `cur` alone: a node (>= 0, the byte offset of its QNode), a leaf (< 0, ~byte offset
// of its TriIsect) or one of three sentinels.  A lane that hits (any-hit) or runs out of stack leaves through the loop
// condition `cur < RT_CUR_POP`; "pop" is a value of cur, not a flag.  Flags (`found`, `pop`) and a `break` in the body make
// the compiler carry exit masks through every divergent region and merge them each iteration: that was a third of the
// loop's scalar instructions (0.323 -> 0.290 ms per k_shadow launch on the 1080p frame without them).  The bottom stack
// entry is EXIT, so a pop needs no empty-stack test.
#define RT_CUR_EXIT 0x7fffffff
#define RT_CUR_HIT 0x7ffffffe
#define RT_CUR_POP 0x7ffffffd

// One iteration for a lane with cur < RT_CUR_POP: it visits either an internal node or a leaf ("if-if").  Measured
// against "while-while" (inner loop over internal nodes, leaves batched per wave) on MI355X: 0.563 ms vs 0.681 ms for the
// shadow rays of atrium-262k -- on 64-wide waves the lanes parked at a leaf cost more than the masked triangle code does
// (round 3, leaves run only when >= 4 / 8 / 16 lanes wait at one: 0.275 / 0.303 / 0.322 against 0.252 ms).
template <bool ANY, bool CENSUS>
__device__ __forceinline__ void bvh_step(const SceneView& sv, const RayPre& rp, f3 O, f3 D, Hit& best, int& cur, lds_int*& top,
                                         lds_int* lim, int stride, uint32_t& cnt_nodes, uint32_t& cnt_tris)
{
	if (cur >= 0) {
		const uint4* np = (const uint4*)((const char*)sv.qnodes + (uint32_t)cur);
		const uint4 w0 = np[0], w1 = np[1];
		const int2 ch = make_int2((int)w1.z, (int)w1.w);
		if (CENSUS) ++cnt_nodes;
		float tl, tr;
		bool hl, hr;
		slab2(rp, w0, w1, ANY ? sv.tp.t_max : best.t, hl, hr, tl, tr);
		// right child first iff it is hit and (the left one is not, or the right one is entered earlier): one compare
		// of the entry distances with +inf standing for "not hit" -- no branch, no mask arithmetic
		const float inf = __builtin_inff();
		const bool right_first = (hr ? tr : inf) < (hl ? tl : inf);
		if (hl && hr) {
			stack_push(sv, top, lim, stride, right_first ? ch.x : ch.y);
			top += stride;
		}
		cur = (hl || hr) ? (right_first ? ch.y : ch.x) : RT_CUR_POP;
	} else {
		const TriIsect* tp = (const TriIsect*)((const char*)sv.tris + (uint32_t)(~cur));
		float4 a = tp->a, b = tp->b, c = tp->c;
		// keep the three loads together: left alone, the compiler sinks the load of `a` below
		// the det test and the leaf visit pays two memory round trips instead of one
		asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w));
		if (CENSUS) ++cnt_tris;
		f3 v0 = mk3(a.x, a.y, a.z), e1 = mk3(b.x, b.y, b.z), e2 = mk3(c.x, c.y, c.z);
		if (ANY) {
			cur = mt_any(O, D, v0, e1, e2, sv.tp) ? RT_CUR_HIT : RT_CUR_POP;
		} else {
			mt_closest(O, D, v0, e1, e2, __float_as_uint(a.w), sv.tp, best);
			cur = RT_CUR_POP;
		}
	}
	if (cur == RT_CUR_POP) { top -= stride; cur = stack_pop(sv, top, lim, stride); }
}

template <bool ANY, bool CENSUS>
__device__ __forceinline__ bool bvh_trace(const SceneView& sv, f3 O, f3 D, Hit& best, lds_int* stack, int stride,
                                          uint32_t& cnt_nodes, uint32_t& cnt_tris)
{
	best.prim = RT_MISS;
	best.t = sv.tp.t_max;
	best.u = 0.0f;
	best.v = 0.0f;
	bool found = false;
	if (sv.num_tris > 0) {
		const RayPre rp = ray_pre(sv.qframe, O, D);
		// num_tris == 1: no internal node; the root "node" is leaf slot 0.
		int cur = (sv.num_tris == 1) ? ~0 : 0;
		lds_int* const lim = stack + sv.lds_depth * stride;
		stack_push(sv, stack, lim, stride, RT_CUR_EXIT);
		lds_int* top = stack + stride;
		while (cur < RT_CUR_POP) bvh_step<ANY, CENSUS>(sv, rp, O, D, best, cur, top, lim, stride, cnt_nodes, cnt_tris);
		found = ANY ? (cur == RT_CUR_HIT) : (best.prim != RT_MISS);
	}
	if (!ANY && best.prim == RT_MISS) best.t = __builtin_inff();
	return found;
}

// Census mode: wave-reduce the per-ray counters, one 64-bit atomic triple per wave.
__device__ __forceinline__ void census_add(unsigned long long* census, uint32_t nodes, uint32_t tris, uint32_t rays)
{
	uint32_t a = nodes, b = tris, c = rays;
	for (int off = 32; off >= 1; off >>= 1) {
		a += __shfl_xor(a, off);
		b += __shfl_xor(b, off);
		c += __shfl_xor(c, off);
	}
	if ((threadIdx.x & 63) == 0) {
		atomicAdd(&census[0], (unsigned long long)a);
		atomicAdd(&census[1], (unsigned long long)b);
		atomicAdd(&census[2], (unsigned long long)c);
	}
}

// Brute-force loop over sv.tris with the same contract (used when the scene has no BVH).
template <bool ANY>
__device__ __forceinline__ bool brute_trace(const SceneView& sv, f3 O, f3 D, Hit& best)
{
	best.prim = RT_MISS;
	best.t = sv.tp.t_max;
	best.u = 0.0f;
	best.v = 0.0f;
	bool found = false;
	for (int k = 0; k < sv.num_tris; ++k) {
		float4 a = sv.tris[k].a, b = sv.tris[k].b, c = sv.tris[k].c;
		f3 v0 = mk3(a.x, a.y, a.z), e1 = mk3(b.x, b.y, b.z), e2 = mk3(c.x, c.y, c.z);
		if (ANY) {
			if (mt_any(O, D, v0, e1, e2, sv.tp)) { found = true; break; }
		} else {
			if (mt_closest(O, D, v0, e1, e2, __float_as_uint(a.w), sv.tp, best)) found = true;
		}
	}
	if (!ANY && best.prim == RT_MISS) best.t = __builtin_inff();
	return found;
}

template <bool ANY>
__device__ __forceinline__ bool scene_trace(const SceneView& sv, f3 O, f3 D, Hit& best, lds_int* stack, int stride)
{
	uint32_t a = 0, b = 0;
	if (sv.use_bvh) return bvh_trace<ANY, false>(sv, O, D, best, stack, stride, a, b);
	return brute_trace<ANY>(sv, O, D, best);
}

// ---- spheres (rt_accel_spheres.cpp) -----------------------------------------------------------
// One sphere of AccelSpheres::intersect's loop body (:66-100).  Ro is passed explicitly; the
// reference passes the DIRECTION as Ro (:68) and callers reproduce that.
__device__ __forceinline__ bool sphere_test_ref(const SphereRec& s, f3 Ro, f3 Rd, f3& X, f3& N)
{
	f3 c = mk3(s.cx, s.cy, s.cz);
	f3 oc = Ro - c;
	float b = dot3(Rd * 2.0f, oc);
	float cc = dot3(oc, oc) - s.r * s.r;
	float disc = (b * b - 4.0f * cc);
	if (disc < 0.0f) return false;
	disc = rt_sqrtf(disc);
	float t0 = (-b - disc) * 0.5f;
	float t1 = (-b + disc) * 0.5f;
	if (t0 < 0.0f && t1 > 0.0f) return false;  // inside the sphere
	t0 = fminf(t0, t1);
	if (t0 <= 0.0f) return false;
	X = Ro + Rd * t0;
	float distance = dot3(X, X);
	if (!(distance < 1000000.0f)) return false;
	N = div3(X - c, s.r);
	return true;
}

__device__ __forceinline__ bool sphere_occlude_ref(const SphereRec& s, f3 Ro, f3 Rd)
{
	f3 c = mk3(s.cx, s.cy, s.cz);
	f3 oc = Ro - c;
	float b = dot3(Rd * 2.0f, oc);
	float cc = dot3(oc, oc) - s.r * s.r;
	return (b * b - 4.0f * cc) > 0.0f;
}
