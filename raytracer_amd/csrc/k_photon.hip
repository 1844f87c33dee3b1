// raytracer_amd/csrc/k_photon.hip -- the photon map on the GPU (raytracer/rt_photon_map.cpp).
//
//   emission  PhotonMap::generate + emit_photon (rt_photon_map.cpp:86-124, :32-69): one lane per
//             photon INDEX of the global sequence lights x photons_per_light, the reference's
//             sampling arithmetic draw for draw, but the draws come from a counter RNG keyed by
//             (seed, photon index, draw number) instead of the serial libc stream (:94), so any
//             shard of indices yields the same photons on any number of GPUs.  Stored photons are
//             compacted stably, i.e. they leave in index order.
//   grid      uniform grid over the photon bounds, cell >= radius; photons counting-sorted by cell
//             with the wave64 radix sort of k_sort.hip (stable: index order inside a cell).
//   gather    PhotonMap::radiance_estimate (:149-208): candidates l2 < r^2, max_dist2 over ALL
//             candidates, the `count` nearest (k-th smallest l2 found by bisection on the float bit
//             pattern), Gaussian weights, the `#else` normalisation 1/(2 pi max_dist2) that the
//             FILTER_GUASS typo selects, NaN for an empty neighbourhood.
#include <math.h>

#include "rt_ctx.h"
#include "rt_trace_dev.h"

namespace {

constexpr int BLOCK = 128;
inline int nblocks(int64_t n, int b) { return (int)((n + b - 1) / b); }

// ---- counter RNG (identical to oracle/rt_oracle.cpp: struct Rng, mode 1) -------------------------
__device__ __forceinline__ uint32_t lowbias32(uint32_t x)
{
	x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
	return x;
}
struct Rng {
	uint32_t seed_lo, seed_hi, index, counter;
	__device__ __forceinline__ int next()  // 31-bit draw, the range of libc rand()
	{
		uint32_t h = lowbias32(index ^ seed_lo);
		h = lowbias32(h + 0x9E3779B9U * (counter + 1u) + seed_hi);
		h = lowbias32(h ^ (counter * 0x85EBCA6BU));
		++counter;
		return (int)(h >> 1);
	}
};

struct PhotonRec {  // 9 floats, the layout of the C ABI
	float p[3], e[3], d[3];
};

__global__ __launch_bounds__(BLOCK) void k_photon_emit(SceneView sv, int per_light, int64_t first, int64_t count,
                                                       uint32_t seed_lo, uint32_t seed_hi, float max_dist0,
                                                       float absorb, PhotonRec* __restrict__ tmp,
                                                       uint32_t* __restrict__ miss_digit)
{
	extern __shared__ __attribute__((aligned(16))) int smem[];
	const int64_t gtid = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
	const int64_t gsize = (int64_t)gridDim.x * BLOCK;
	for (int64_t k = gtid; k < count; k += gsize) {
	int64_t idx = first + k;
	int light = (int)(idx / per_light);
	Rng rng;
	rng.seed_lo = seed_lo; rng.seed_hi = seed_hi; rng.index = (uint32_t)idx; rng.counter = 0;
	const float rmax_2 = 1073741824.0f;   // float(RAND_MAX / 2.0)   rt_photon_map.cpp:20,93
	const float rand_max = 2147483648.0f; // (float)RAND_MAX
	f3 dir;
	// generate(): dir in [-1,1]^2 x [-1,0], accepted when |dir|^2 <= 1, NOT normalised (:99-104)
	for (;;) {
		dir.x = (float)rng.next() / rmax_2 - 1.0f;
		dir.y = (float)rng.next() / rmax_2 - 1.0f;
		dir.z = (float)(-rng.next()) / rand_max;
		if (!(dot3(dir, dir) > 1.0f)) break;
	}
	f3 color = ld3(sv.lights[light].color);
	f3 energy = div3(color * 5.0f, (float)per_light);  // 5*light->color/count (:97)
	f3 origin = ld3(sv.lights[light].position);
	float max_dist = max_dist0;
	bool stored = false;
	PhotonRec out;
	// emit_photon(): `pass++` hands the OLD value down, so every level runs with pass == 0 (:66)
	for (int depth = 0; depth < (1 << 20); ++depth) {
		Hit best;
		scene_trace<false>(sv, origin, dir, best, RT_LDS_STACK(smem) + threadIdx.x, BLOCK);
		if (best.prim == RT_MISS) break;
		const TriPos tp = sv.tri_pos[best.prim];
		const TriAttr ta = sv.tri_attr[best.prim];
		f3 pos = bary3(best.u, best.v, mk3(tp.p0.x, tp.p0.y, tp.p0.z), mk3(tp.p1.x, tp.p1.y, tp.p1.z), mk3(tp.p2.x, tp.p2.y, tp.p2.z));
		f3 nrm = normalized3(bary3(best.u, best.v, ld3(ta.n0), ld3(ta.n1), ld3(ta.n2)));
		float dist = norm3(pos - origin);
		if (dist > max_dist) break;
		if ((float)rng.next() / rand_max < absorb) {
			st3(out.p, pos); st3(out.e, energy); st3(out.d, dir);
			stored = true;
			break;
		}
		origin = pos;
		f3 nd = mk3(1.0f, 1.0f, 1.0f);
		int guard = 0;
		while ((dot3(nd, nd) > 1.0f || dot3(nd, nrm) < 0.0f) && guard < (1 << 22)) {
			nd.x = (float)rng.next() / rmax_2 - 1.0f;
			nd.y = (float)rng.next() / rmax_2 - 1.0f;
			nd.z = (float)rng.next() / rmax_2 - 1.0f;
			++guard;
		}
		dir = nd;
		int mat = __float_as_int(tp.p0.w);
		energy = ld3(sv.materials[mat].diffuse) * (norm3(energy) / 255.0f);  // (:63)
		max_dist = max_dist - dist;
	}
	miss_digit[k] = stored ? 0u : 1u;
	if (stored) tmp[k] = out;
	}
}

__global__ __launch_bounds__(256) void k_photon_compact(const PhotonRec* __restrict__ tmp,
                                                        const uint32_t* __restrict__ digit,
                                                        const uint32_t* __restrict__ dest, int64_t n,
                                                        PhotonRec* __restrict__ out)
{
	int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (k < n && digit[k] == 0u) out[dest[k]] = tmp[k];
}

// ---- grid -------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t f2ord(float f)
{
	uint32_t u = __float_as_uint(f);
	return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(256) void k_photon_bounds(const PhotonRec* __restrict__ ph, int64_t n,
                                                       uint32_t* __restrict__ bounds_ord)
{
	__shared__ uint32_t sb[6];
	if (threadIdx.x < 3) sb[threadIdx.x] = 0xFFFFFFFFu;
	else if (threadIdx.x < 6) sb[threadIdx.x] = 0u;
	__syncthreads();
	int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) {
		for (int k = 0; k < 3; ++k) {
			uint32_t o = f2ord(ph[i].p[k]);
			atomicMin(&sb[k], o);
			atomicMax(&sb[3 + k], o);
		}
	}
	__syncthreads();
	if (threadIdx.x < 3) atomicMin(&bounds_ord[threadIdx.x], sb[threadIdx.x]);
	else if (threadIdx.x < 6) atomicMax(&bounds_ord[threadIdx.x], sb[threadIdx.x]);
}

struct GridDesc {
	float mn[3];
	float inv_cell;
	int dim[3];
	float cell;
};

__device__ __forceinline__ int cell_coord(float p, float mn, float inv_cell, int dim)
{
	int c = (int)floorf((p - mn) * inv_cell);
	return min(max(c, 0), dim - 1);
}

__global__ __launch_bounds__(256) void k_photon_cell_keys(const PhotonRec* __restrict__ ph, int64_t n, GridDesc gd,
                                                          uint64_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
	int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	int cx = cell_coord(ph[i].p[0], gd.mn[0], gd.inv_cell, gd.dim[0]);
	int cy = cell_coord(ph[i].p[1], gd.mn[1], gd.inv_cell, gd.dim[1]);
	int cz = cell_coord(ph[i].p[2], gd.mn[2], gd.inv_cell, gd.dim[2]);
	keys[i] = (uint64_t)((cz * gd.dim[1] + cy) * gd.dim[0] + cx);
	vals[i] = (uint32_t)i;
}

// sorted[s] = photons[order[s]] split into a 16-byte position record and a 32-byte payload
__global__ __launch_bounds__(256) void k_photon_permute(const PhotonRec* __restrict__ ph,
                                                        const uint32_t* __restrict__ order,
                                                        const uint64_t* __restrict__ keys, int64_t n, int ncells,
                                                        float4* __restrict__ pos, float4* __restrict__ payload)
{
	int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
	(void)keys; (void)ncells;
	if (s >= n) return;
	const PhotonRec r = ph[order[s]];
	pos[s] = make_float4(r.p[0], r.p[1], r.p[2], __uint_as_float(order[s]));
	payload[2 * s] = make_float4(r.e[0], r.e[1], r.e[2], r.d[0]);
	payload[2 * s + 1] = make_float4(r.d[1], r.d[2], 0.0f, 0.0f);
}

// cell_start[c] = first sorted slot whose cell id is >= c (lower bound on the sorted keys), for c in [0, ncells]:
// one thread per cell, so a sparse grid costs log2(n) reads per cell instead of one thread walking a long gap.
__global__ __launch_bounds__(256) void k_photon_cell_start(const uint64_t* __restrict__ keys, int64_t n, int ncells,
                                                           uint32_t* __restrict__ cell_start)
{
	int c = blockIdx.x * 256 + threadIdx.x;
	if (c > ncells) return;
	int64_t lo = 0, hi = n;
	while (lo < hi) {
		const int64_t mid = (lo + hi) >> 1;
		if (keys[mid] < (uint64_t)c) lo = mid + 1; else hi = mid;
	}
	cell_start[c] = (uint32_t)lo;
}

// ---- gather -------------------------------------------------------------------------------------
struct GatherConst {
	float alpha, beta, e, gauss_denom, two_pi;
};

__device__ __forceinline__ float bound_sqrtf(float x) { return __builtin_amdgcn_sqrtf(x); }   // for conservative bounds only

// The x-cells of grid row (y, z) that can hold a photon within rho of P: the row's distance from P in the
// yz-plane bounds the chord of the sphere along x.  false: the whole row is out of reach.
__device__ __forceinline__ bool row_chord(const GridDesc& gd, f3 P, float rho, int y, int z, int& x0, int& x1)
{
	const float ylo = gd.mn[1] + (float)y * gd.cell, zlo = gd.mn[2] + (float)z * gd.cell;
	// distance from P to the row's [lo, lo + cell] interval per axis (0 inside); the border rows of the grid
	// also hold what cell_coord clamped into them, so they count as reaching to infinity
	float dy = fmaxf(fmaxf(ylo - P.y, P.y - (ylo + gd.cell)), 0.0f);
	float dz = fmaxf(fmaxf(zlo - P.z, P.z - (zlo + gd.cell)), 0.0f);
	if ((y == 0 && P.y < ylo) || (y == gd.dim[1] - 1 && P.y > ylo + gd.cell)) dy = 0.0f;
	if ((z == 0 && P.z < zlo) || (z == gd.dim[2] - 1 && P.z > zlo + gd.cell)) dz = 0.0f;
	// never over-estimate a gap: the row edges computed here and cell_coord's floor() may disagree by a few ulps of
	// the grid's extent, far below a ten-thousandth of a cell
	const float slack = gd.cell * 1e-4f;
	dy = fmaxf(dy * 0.9999f - slack, 0.0f); dz = fmaxf(dz * 0.9999f - slack, 0.0f);
	const float h2 = rho * rho - (dy * dy + dz * dz);
	if (h2 < 0.0f) return false;
	// bound_sqrtf: v_sqrt_f32 (1 ulp) instead of the correctly rounded sequence; the margins below are 1e-4 relative
	const float half = bound_sqrtf(h2) * 1.0001f + slack;
	x0 = cell_coord(P.x - half, gd.mn[0], gd.inv_cell, gd.dim[0]);
	x1 = cell_coord(P.x + half, gd.mn[0], gd.inv_cell, gd.dim[0]);
	return true;
}

// Visit every photon whose grid cell can hold a photon within rho of P: the rows (y, z) of the cube of
// half-width rho, each cut to the sphere's chord.  Cells along x are contiguous in the sorted array: one
// span per row.
template <typename F>
__device__ __forceinline__ void for_each_neighbour(const GridDesc& gd, const uint32_t* __restrict__ cell_start,
                                                   const float4* __restrict__ pos, f3 P, float rho, F f)
{
	rho = rho * 1.0001f + 1e-30f;  // the cell of a photon at distance <= rho must not fall outside by rounding
	int y0 = cell_coord(P.y - rho, gd.mn[1], gd.inv_cell, gd.dim[1]), y1 = cell_coord(P.y + rho, gd.mn[1], gd.inv_cell, gd.dim[1]);
	int z0 = cell_coord(P.z - rho, gd.mn[2], gd.inv_cell, gd.dim[2]), z1 = cell_coord(P.z + rho, gd.mn[2], gd.inv_cell, gd.dim[2]);
	for (int z = z0; z <= z1; ++z)
		for (int y = y0; y <= y1; ++y) {
			int x0, x1;
			if (!row_chord(gd, P, rho, y, z, x0, x1)) continue;
			uint32_t row = (uint32_t)((z * gd.dim[1] + y) * gd.dim[0]);
			uint32_t s0 = cell_start[row + x0], s1 = cell_start[row + x1 + 1];
			// eight photons in flight per lane: the loads are independent of the loop body, and a
			// lane walking its span one dependent 16-byte load at a time is latency-bound
			uint32_t s = s0;
			for (; s + 8 <= s1; s += 8) {
				float4 q[8];
#pragma unroll
				for (int k = 0; k < 8; ++k) q[k] = pos[s + k];
#pragma unroll
				for (int k = 0; k < 8; ++k) {
					f3 d = mk3(q[k].x, q[k].y, q[k].z) - P;
					f(s + k, dot3(d, d));
				}
			}
			for (; s < s1; ++s) {
				float4 q = pos[s];
				f3 d = mk3(q.x, q.y, q.z) - P;
				f(s, dot3(d, d));
			}
		}
}

// Photons in the cells for_each_neighbour would visit (an upper bound of the photons within rho).
__device__ __forceinline__ uint32_t cube_population(const GridDesc& gd, const uint32_t* __restrict__ cell_start, f3 P, float rho)
{
	rho = rho * 1.0001f + 1e-30f;
	int y0 = cell_coord(P.y - rho, gd.mn[1], gd.inv_cell, gd.dim[1]), y1 = cell_coord(P.y + rho, gd.mn[1], gd.inv_cell, gd.dim[1]);
	int z0 = cell_coord(P.z - rho, gd.mn[2], gd.inv_cell, gd.dim[2]), z1 = cell_coord(P.z + rho, gd.mn[2], gd.inv_cell, gd.dim[2]);
	uint32_t total = 0;
	for (int z = z0; z <= z1; ++z)
		for (int y = y0; y <= y1; ++y) {
			int x0, x1;
			if (!row_chord(gd, P, rho, y, z, x0, x1)) continue;
			uint32_t row = (uint32_t)((z * gd.dim[1] + y) * gd.dim[0]);
			total += cell_start[row + x1 + 1] - cell_start[row + x0];
		}
	return total;
}

// row / ny by multiplication: umulhi(row, floor((2^32 - 1) / ny) + 1) == row / ny for every row < 2^16 and ny <= 4096
// (checked exhaustively); a larger cube of rows takes the division.
__device__ __forceinline__ uint32_t row_div_magic(int ny) { return 0xFFFFFFFFu / (uint32_t)ny + 1u; }
__device__ __forceinline__ int row_div(int row, int ny, uint32_t magic, bool small)
{
	return small ? (int)__umulhi((uint32_t)row, magic) : row / ny;
}

__device__ __forceinline__ uint32_t wave_sum_u(uint32_t v)
{
	for (int off = 32; off >= 1; off >>= 1) v += (uint32_t)__shfl_xor((int)v, off);
	return v;
}

// The wave visits every sorted slot of 64 runs [r0, r1), one run held per lane (empty: r1 <= r0).  A point's
// neighbourhood is many short runs (one per grid row) and a wave that reads them one after the other waits one memory
// round trip per run -- that, not the bytes, bounded the gather (~140 us per point).  So: runs of 256 slots and more
// are read one at a time with coalesced 1-KB loads, four in flight; all shorter runs are FLATTENED -- a prefix sum of
// their lengths across the lanes, then element e of the concatenation belongs to the run found by a 6-step binary
// search over the lanes' prefix sums (shuffles) -- and read 256 elements per round, whatever run they belong to.
// visit(q, slot, active) runs in wave-uniform control flow.
#define RT_GATHER_LONG 256u
// Inclusive scans across the 64 lanes on the DPP network (no LDS traffic): within rows of 16 by row_shr 1, 2, 4, 8, then
// row 0's total into row 1 and row 2's into row 3 (row_bcast:15), then the lower half's total into the upper (row_bcast:31).
#define RT_DPP_SCAN(v, OP)                                                                                      \
	do {                                                                                                        \
		v = OP(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), 0x111, 0xf, 0xf, false));                  \
		v = OP(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), 0x112, 0xf, 0xf, false));                  \
		v = OP(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), 0x114, 0xf, 0xf, false));                  \
		v = OP(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), 0x118, 0xf, 0xf, false));                  \
		v = OP(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), 0x142, 0xa, 0xf, false));                  \
		v = OP(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), 0x143, 0xc, 0xf, false));                  \
	} while (0)
__device__ __forceinline__ uint32_t dpp_op_add(uint32_t a, uint32_t b) { return a + b; }
__device__ __forceinline__ uint32_t dpp_op_max(uint32_t a, uint32_t b) { return max(a, b); }
__device__ __forceinline__ uint32_t wave_scan_add_u32(uint32_t v) { RT_DPP_SCAN(v, dpp_op_add); return v; }
__device__ __forceinline__ uint32_t wave_scan_max_u32(uint32_t v) { RT_DPP_SCAN(v, dpp_op_max); return v; }

// `mark`: 64 dwords of LDS owned by the calling wave.
template <bool MARKS, typename V>
__device__ __forceinline__ void wave_visit_runs(uint32_t* mark, const float4* __restrict__ pos, uint32_t r0, uint32_t r1, V visit)
{
	const int lane = lane_id();
	uint32_t len = r1 > r0 ? r1 - r0 : 0u;
	uint64_t longm = __ballot(len >= RT_GATHER_LONG);
	while (longm) {
		const int src = __ffsll((unsigned long long)longm) - 1;
		longm &= longm - 1;
		const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)r0, src), b = (uint32_t)__builtin_amdgcn_readlane((int)r1, src);
		uint32_t s = a;
		for (; s + 256 <= b; s += 256) {
			const float4 qa = pos[s + lane], qb = pos[s + 64 + lane], qc = pos[s + 128 + lane], qd = pos[s + 192 + lane];
			visit(qa, s + lane, true); visit(qb, s + 64 + lane, true); visit(qc, s + 128 + lane, true); visit(qd, s + 192 + lane, true);
		}
		for (; s < b; s += 64) {
			// padding lanes re-read the run's last slot: a load under a branch is waited for at once, and the
			// loads of this loop would queue up behind each other's memory round trips
			const bool act = s + lane < b;
			const float4 q = pos[act ? s + lane : b - 1];
			visit(q, s + lane, act);
		}
	}
	if (len >= RT_GATHER_LONG) len = 0u;
	const uint32_t end = wave_scan_add_u32(len);   // inclusive prefix sum over the lanes
	const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)end, 63);
	const uint32_t base = r0 - (end - len);   // slot of flattened element e inside this lane's run = base + e
	// Each lane takes FOUR consecutive elements of the concatenation per round.  Which run an element belongs to: every
	// non-empty run that starts inside the round's 256 elements writes its lane number + 1 into a byte table at its start
	// (starts are distinct), and an element's run is the maximum of the marks at or before it -- the lane's own four bytes,
	// a max-scan across the lanes, and the last run of the rounds before.  (A 6-step binary search over the prefix sums by
	// shuffles, plus a shuffle per further element, cost 4 x the instructions.)
	const uint64_t nonempty = __ballot(len > 0u);
	const uint32_t first_slot = nonempty ? (uint32_t)__builtin_amdgcn_readlane((int)r0, __ffsll((unsigned long long)nonempty) - 1) : 0u;
	const uint32_t start = end - len;
	uint32_t carry = 0u;   // mark of the last run that started before this round
	if constexpr (!MARKS) {
		// the older form, kept for the short sweeps of the selection, which measured faster with it: one binary search over
		// the lanes' prefix sums per lane and round, then one shuffle per further element (same run, or the next non-empty one)
		for (uint32_t e0 = 0; e0 < total; e0 += 256) {
			float4 q[4];
			uint32_t slot[4];
			bool act[4];
			const uint32_t e = e0 + 4u * lane;
			int j = 0;   // the first lane whose prefix sum exceeds e (never an empty run)
#pragma unroll
			for (int step = 32; step >= 1; step >>= 1)
				if (__shfl(end, min(j + step - 1, 63)) <= e) j += step;
			j = min(j, 63);
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const uint32_t ek = e + k;
				act[k] = ek < total;
				if (k > 0) {
					const uint32_t t = __shfl(end, j);
					if (t <= ek) {
						const uint64_t rest = j >= 63 ? 0ull : (nonempty >> (j + 1));
						j = rest ? j + __ffsll((unsigned long long)rest) : 63;
					}
				}
				slot[k] = __shfl(base, j) + ek;
				q[k] = pos[act[k] ? slot[k] : first_slot];
			}
#pragma unroll
			for (int k = 0; k < 4; ++k) visit(q[k], slot[k], act[k]);
		}
		return;
	}
	for (uint32_t e0 = 0; e0 < total; e0 += 256) {
		float4 q[4];
		uint32_t slot[4];
		bool act[4];
		const uint32_t e = e0 + 4u * lane;
		mark[lane] = 0u;
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
		if (len > 0u && start - e0 < 256u) ((uint8_t*)mark)[start - e0] = (uint8_t)(lane + 1);
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
		const uint32_t w = mark[lane];
		uint32_t own[4];
		own[0] = w & 0xffu;
		own[1] = max(own[0], (w >> 8) & 0xffu);
		own[2] = max(own[1], (w >> 16) & 0xffu);
		own[3] = max(own[2], w >> 24);
		const uint32_t inc = wave_scan_max_u32(own[3]);
		// the lanes before this one: inc shifted up by a lane (wave_shr:1), the carry into lane 0
		const uint32_t before = max(carry, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x138, 0xf, 0xf, false));
		carry = max(carry, (uint32_t)__builtin_amdgcn_readlane((int)inc, 63));
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const uint32_t ek = e + k;
			act[k] = ek < total;
			const uint32_t o = max(own[k], before);   // >= 1 for every element below `total`
			slot[k] = __shfl(base, (int)((o - 1u) & 63u)) + ek;
			// unconditional (see above): the four loads are in flight together; padding lanes read the run's first slot
			q[k] = pos[act[k] ? slot[k] : first_slot];
		}
#pragma unroll
		for (int k = 0; k < 4; ++k) visit(q[k], slot[k], act[k]);
	}
}

// The walk of for_each_neighbour with a whole wave on one point (P and rho are wave-uniform): the grid rows of the cube
// are dealt to the lanes, 64 at a time, each lane looks up its row's run, and the wave visits the 64 runs together.
// f(slot, l2, active) runs in uniform control flow, `active` false in the padding.
#define RT_GATHER_MARKS_SEL 0
template <bool MARKS = (RT_GATHER_MARKS_SEL != 0), typename F>
__device__ __forceinline__ void for_each_neighbour_wave(uint32_t* mark, const GridDesc& gd, const uint32_t* __restrict__ cell_start,
                                                        const float4* __restrict__ pos, f3 P, float rho, F f)
{
	rho = rho * 1.0001f + 1e-30f;
	const int y0 = cell_coord(P.y - rho, gd.mn[1], gd.inv_cell, gd.dim[1]), y1 = cell_coord(P.y + rho, gd.mn[1], gd.inv_cell, gd.dim[1]);
	const int z0 = cell_coord(P.z - rho, gd.mn[2], gd.inv_cell, gd.dim[2]), z1 = cell_coord(P.z + rho, gd.mn[2], gd.inv_cell, gd.dim[2]);
	const int ny = y1 - y0 + 1, nrows = ny * (z1 - z0 + 1);
	const int lane = lane_id();
	const uint32_t ny_magic = row_div_magic(ny);
	const bool rows_small = nrows < 65536 && ny <= 4096 && ny > 1;
	for (int rbase = 0; rbase < nrows; rbase += 64) {
		const int row = rbase + lane;
		uint32_t s0 = 0, s1 = 0;
		if (row < nrows) {
			const int zr = row_div(row, ny, ny_magic, rows_small);
			const int y = y0 + (row - zr * ny), z = z0 + zr;
			int x0, x1;
			if (row_chord(gd, P, rho, y, z, x0, x1)) {
				const uint32_t rb = (uint32_t)((z * gd.dim[1] + y) * gd.dim[0]);
				s0 = cell_start[rb + x0]; s1 = cell_start[rb + x1 + 1];
			}
		}
		wave_visit_runs<MARKS>(mark, pos, s0, s1, [&](const float4 q, uint32_t slot, bool act) {
			const f3 d = mk3(q.x, q.y, q.z) - P;
			f(slot, dot3(d, d), act);
		});
	}
}

__device__ __forceinline__ int wave_sum_i(int v)
{
	for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
	return v;
}
__device__ __forceinline__ float wave_sum_f(float v)
{
	for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
	return v;
}
__device__ __forceinline__ float wave_max_f(float v)
{
	for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
	return v;
}

// Pass 1 of the wave-per-point gather on the fine grid: the number of photons within the radius and max_dist2 over ALL
// of them (rt_photon_map.cpp:159-169) without reading all of them.  Per grid row (y, z) the cells along x split into
// an INTERIOR run -- cells whose farthest corner lies inside the sphere: every photon in them is a candidate, so they
// are counted from the cell table and not read -- and at most two SHELL runs at the ends of the chord, which are read.
// The farthest candidate is almost always a shell photon; only if an interior run could hold a farther one (its far
// corner is farther than the best shell candidate, e.g. an empty shell) is that run read too, in a second sweep.
// Rows are dealt to the lanes for the table work, the wave then reads the 64 rows' shell runs together.
template <bool SECOND>
__device__ __forceinline__ void wave_shell_sweep(uint32_t* mark, const GridDesc& gd, const uint32_t* __restrict__ cs, const float4* __restrict__ pos,
                                                 f3 P, float radius, float r2, float mx_in, int& nc_l, float& mx_l, float& bound_l)
{
	const float rho = radius * 1.0001f + 1e-30f;
	const int y0 = cell_coord(P.y - rho, gd.mn[1], gd.inv_cell, gd.dim[1]), y1 = cell_coord(P.y + rho, gd.mn[1], gd.inv_cell, gd.dim[1]);
	const int z0 = cell_coord(P.z - rho, gd.mn[2], gd.inv_cell, gd.dim[2]), z1 = cell_coord(P.z + rho, gd.mn[2], gd.inv_cell, gd.dim[2]);
	const int ny = y1 - y0 + 1, nrows = ny * (z1 - z0 + 1);
	const int lane = lane_id();
	const uint32_t ny_magic = row_div_magic(ny);
	const bool rows_small = nrows < 65536 && ny <= 4096 && ny > 1;
	// a photon sits in the cell floor((p - mn) / cell) computed in float: its coordinate may be off the cell's nominal
	// box by a few ulps of the grid extent, far below a thousandth of a cell
	const float slack = gd.cell * 1e-3f;
	// 0.999: the interior runs' far corners then stay below 0.998 r^2, under the farthest shell candidate of any
	// point with a few hundred photons around it, so the second sweep stays the exception
	const float rin = radius * 0.999f;
	for (int base = 0; base < nrows; base += 64) {
		const int row = base + lane;
		uint32_t a0 = 0, a1 = 0, b0 = 0, b1 = 0;  // the runs this lane's row contributes to the wave's reading
		if (row < nrows) {
			const int zr = row_div(row, ny, ny_magic, rows_small);
			const int y = y0 + (row - zr * ny), z = z0 + zr;
			int x0, x1;
			if (row_chord(gd, P, rho, y, z, x0, x1)) {
				const uint32_t rb = (uint32_t)((z * gd.dim[1] + y) * gd.dim[0]);
				const uint32_t s0 = cs[rb + x0], s1 = cs[rb + x1 + 1];
				if (!SECOND) { a0 = s0; a1 = s1; }
				// the border cells of the grid also hold what cell_coord clamped into them: never interior
				const bool border = y == 0 || y == gd.dim[1] - 1 || z == 0 || z == gd.dim[2] - 1;
				if (s1 > s0 && !border) {
					const float ylo = gd.mn[1] + (float)y * gd.cell, zlo = gd.mn[2] + (float)z * gd.cell;
					const float dyf = fmaxf(fabsf(ylo - P.y), fabsf(ylo + gd.cell - P.y)) + slack;
					const float dzf = fmaxf(fabsf(zlo - P.z), fabsf(zlo + gd.cell - P.z)) + slack;
					const float h2 = rin * rin - (dyf * dyf + dzf * dzf);
					if (h2 > 0.0f) {
						const float half = bound_sqrtf(h2) * 0.9999f - slack;
						int xi0 = (int)ceilf((P.x - half - gd.mn[0]) * gd.inv_cell + 1e-3f);
						int xi1 = (int)floorf((P.x + half - gd.mn[0]) * gd.inv_cell - 1e-3f) - 1;
						xi0 = max(xi0, max(x0, 1)); xi1 = min(xi1, min(x1, gd.dim[0] - 2));
						if (xi0 <= xi1) {
							const uint32_t i0 = cs[rb + xi0], i1 = cs[rb + xi1 + 1];
							const float dxf = fmaxf(fabsf(gd.mn[0] + (float)xi0 * gd.cell - P.x),
							                        fabsf(gd.mn[0] + (float)(xi1 + 1) * gd.cell - P.x)) + slack;
							const float far2 = (dxf * dxf + (dyf * dyf + dzf * dzf)) * 1.0001f;  // >= every l2 in the run
							if (!SECOND) {
								nc_l += (int)(i1 - i0);
								a1 = i0; b0 = i1; b1 = s1;
								if (i1 > i0) bound_l = fmaxf(bound_l, far2);
							} else if (i1 > i0 && far2 > mx_in) {
								a0 = i0; a1 = i1;
							}
						}
					}
				}
			}
		}
		auto visit = [&](const float4 q, uint32_t, bool act) {
			const f3 d = mk3(q.x, q.y, q.z) - P;
			const float l2 = dot3(d, d);
			// branch-free, and the maximum taken on the bit patterns (l2 >= 0): fmaxf costs two canonicalising
			// v_max more per photon
			const bool in = act && l2 < r2;
			if (!SECOND) nc_l += in ? 1 : 0;
			mx_l = __uint_as_float(max(__float_as_uint(mx_l), in ? __float_as_uint(l2) : 0u));
		};
		wave_visit_runs<true>(mark, pos, a0, a1, visit);
		if (__ballot(b1 > b0)) wave_visit_runs<true>(mark, pos, b0, b1, visit);
	}
}

// wave_shell_sweep for TWO points at once (neighbours in the queue, i.e. on the screen: A and B a fraction of a cell apart).
// One walk of the grid rows, one flattening of the runs and one load per photon serve both points, and the distance
// arithmetic runs two points per instruction (packed FP32: the photon's coordinate against (A, B) -- the same IEEE
// operations per component, in dot3's order, so each point gets the bits it would get alone).  The walk is made for the
// pair's midpoint C with en >= |A - C| = |B - C|: a cell within 0.999 r - en of C is interior to both spheres (counted
// from the table), a cell beyond r + en is outside both, the photons between are tested against both.  bound_l as in
// wave_shell_sweep, for C: the farthest interior photon is within sqrt(bound) + en of either point.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void wave_shell_sweep_pair(uint32_t* mark, const GridDesc& gd, const uint32_t* __restrict__ cs, const float4* __restrict__ pos,
                                                      f3 C, f3 A, f3 B, float radius, float en, float r2, int& common_l, int& nca_l, int& ncb_l,
                                                      uint32_t& mxa_u, uint32_t& mxb_u, float& bound_l)
{
	const float rho = (radius + en) * 1.0001f + 1e-30f;
	const float rin = radius * 0.999f - en;
	const int y0 = cell_coord(C.y - rho, gd.mn[1], gd.inv_cell, gd.dim[1]), y1 = cell_coord(C.y + rho, gd.mn[1], gd.inv_cell, gd.dim[1]);
	const int z0 = cell_coord(C.z - rho, gd.mn[2], gd.inv_cell, gd.dim[2]), z1 = cell_coord(C.z + rho, gd.mn[2], gd.inv_cell, gd.dim[2]);
	const int ny = y1 - y0 + 1, nrows = ny * (z1 - z0 + 1);
	const int lane = lane_id();
	const uint32_t ny_magic = row_div_magic(ny);
	const bool rows_small = nrows < 65536 && ny <= 4096 && ny > 1;
	const float slack = gd.cell * 1e-3f;
	const v2f PX = {A.x, B.x}, PY = {A.y, B.y}, PZ = {A.z, B.z};
	for (int base = 0; base < nrows; base += 64) {
		const int row = base + lane;
		uint32_t a0 = 0, a1 = 0, b0 = 0, b1 = 0;
		if (row < nrows) {
			const int zr = row_div(row, ny, ny_magic, rows_small);
			const int y = y0 + (row - zr * ny), z = z0 + zr;
			int x0, x1;
			if (row_chord(gd, C, rho, y, z, x0, x1)) {
				const uint32_t rb = (uint32_t)((z * gd.dim[1] + y) * gd.dim[0]);
				a0 = cs[rb + x0]; a1 = cs[rb + x1 + 1];
				const bool border = y == 0 || y == gd.dim[1] - 1 || z == 0 || z == gd.dim[2] - 1;
				if (a1 > a0 && !border && rin > 0.0f) {   // (rin <= 0: a pair farther apart than the radius has no common interior)
					const float ylo = gd.mn[1] + (float)y * gd.cell, zlo = gd.mn[2] + (float)z * gd.cell;
					const float dyf = fmaxf(fabsf(ylo - C.y), fabsf(ylo + gd.cell - C.y)) + slack;
					const float dzf = fmaxf(fabsf(zlo - C.z), fabsf(zlo + gd.cell - C.z)) + slack;
					const float h2 = rin * rin - (dyf * dyf + dzf * dzf);
					if (h2 > 0.0f) {
						const float half = bound_sqrtf(h2) * 0.9999f - slack;
						int xi0 = (int)ceilf((C.x - half - gd.mn[0]) * gd.inv_cell + 1e-3f);
						int xi1 = (int)floorf((C.x + half - gd.mn[0]) * gd.inv_cell - 1e-3f) - 1;
						xi0 = max(xi0, max(x0, 1)); xi1 = min(xi1, min(x1, gd.dim[0] - 2));
						if (xi0 <= xi1) {
							const uint32_t i0 = cs[rb + xi0], i1 = cs[rb + xi1 + 1];
							const float dxf = fmaxf(fabsf(gd.mn[0] + (float)xi0 * gd.cell - C.x),
							                        fabsf(gd.mn[0] + (float)(xi1 + 1) * gd.cell - C.x)) + slack;
							const float far2 = (dxf * dxf + (dyf * dyf + dzf * dzf)) * 1.0001f;  // >= |q - C|^2 in the run
							common_l += (int)(i1 - i0);
							b0 = i1; b1 = a1; a1 = i0;
							if (i1 > i0) bound_l = fmaxf(bound_l, far2);
						}
					}
				}
			}
		}
		auto visit = [&](const float4 q, uint32_t, bool act) {
			const v2f qx = {q.x, q.x}, qy = {q.y, q.y}, qz = {q.z, q.z};
			const v2f dx = qx - PX, dy = qy - PY, dz = qz - PZ;
			const v2f l2 = dx * dx + (dy * dy + dz * dz);   // dot3's order
			const bool ina = act && l2.x < r2, inb = act && l2.y < r2;
			nca_l += ina ? 1 : 0;
			ncb_l += inb ? 1 : 0;
			mxa_u = max(mxa_u, ina ? __float_as_uint(l2.x) : 0u);
			mxb_u = max(mxb_u, inb ? __float_as_uint(l2.y) : 0u);
		};
		wave_visit_runs<true>(mark, pos, a0, a1, visit);
		if (__ballot(b1 > b0)) wave_visit_runs<true>(mark, pos, b0, b1, visit);
	}
}

// PhotonMap::radiance_estimate (rt_photon_map.cpp:149-208), one lane per shading point.
//   pass 1   all photons within `radius`: their number, max_dist2 over ALL of them (:159-169), and a
//            64-bin histogram of the squared distance over [0, r^2)
//   select   the `count`-th smallest squared distance tau by histogram refinement: the bin holding the
//            count-th is split into 64 again (at most SEL_LEVELS times), each time reading only the
//            cells that can hold a photon below the bin's upper edge; when at most SEL_LIST
//            candidates remain they are ranked in LDS.  Ties at tau are taken in traversal order.
//   pass 3   the weighted sum over the selected photons (:178-197)
// Histograms and candidate lists are per lane, laid out [slot][lane] in LDS (conflict free).
constexpr int GBLOCK = 128;
constexpr int SEL_BINS = 64;
constexpr int SEL_LEVELS = 4;
constexpr int SEL_LIST = 64;
constexpr int SEL_BELOW = 256;  // wave kernel: selected photons queued for the sum

struct SelLevel {
	float lo, scale;  // bin(l2) = clamp(int((l2 - lo) * scale), 0, 63)
	int bin;          // the bin the search descended into
};

__device__ __forceinline__ int sel_bin(float l2, float lo, float scale)
{
	int b = (int)((l2 - lo) * scale);
	return min(max(b, 0), SEL_BINS - 1);
}

// 21 bits of x, y, z interleaved (x lowest)
__device__ __forceinline__ uint64_t spread3_21(uint32_t v)
{
	uint64_t x = v & 0x1fffffu;
	x = (x | (x << 32)) & 0x1f00000000ffffull;
	x = (x | (x << 16)) & 0x1f0000ff0000ffull;
	x = (x | (x << 8)) & 0x100f00f00f00f00full;
	x = (x | (x << 4)) & 0x10c30c30c30c30c3ull;
	x = (x | (x << 2)) & 0x1249249249249249ull;
	return x;
}
__device__ __forceinline__ uint64_t morton3_63(uint32_t x, uint32_t y, uint32_t z) { return spread3_21(x) | (spread3_21(y) << 1) | (spread3_21(z) << 2); }

__global__ __launch_bounds__(GBLOCK) void k_photon_gather(GridDesc gd, const uint32_t* __restrict__ cell_start,
                                                          const float4* __restrict__ pos,
                                                          const float4* __restrict__ payload, int64_t nph,
                                                          const float* __restrict__ qpos, const float* __restrict__ qnrm,
                                                          int npts, int count, float radius, GatherConst gc,
                                                          float* __restrict__ rgb, uint32_t heavy_threshold,
                                                          int* __restrict__ heavy_queue, int* __restrict__ heavy_count,
                                                          GridDesc gf, uint64_t* __restrict__ heavy_keys)
{
	__shared__ uint32_t s_hist[SEL_BINS * GBLOCK];  // doubles as the candidate list (SEL_LIST <= SEL_BINS)
	uint32_t* hist = s_hist + threadIdx.x;
	int i = blockIdx.x * GBLOCK + threadIdx.x;
	const bool valid = i < npts;
	f3 P = mk3(0, 0, 0), N = mk3(0, 0, 0);
	if (valid) { P = ld3(qpos + 3 * (size_t)i); N = ld3(qnrm + 3 * (size_t)i); }
	// Load balance: the number of photons around a point spans five orders of magnitude (C5: median 0,
	// 99th percentile 7e4, maximum 4e5 within the radius), and a lane walks its neighbourhood alone.  A
	// point whose cube holds more than `heavy_threshold` photons is handed to k_photon_gather_wave, which
	// puts a whole wave on it; one queue slot per point, one atomic per wave.
	bool heavy = false;
	if (valid && nph > 0) heavy = cube_population(gd, cell_start, P, radius) > heavy_threshold;
	{
		const uint64_t hm = __ballot(heavy);
		if (hm) {
			int base = 0;
			if (lane_id() == 0) base = atomicAdd(heavy_count, (int)__popcll(hm));
			base = __shfl(base, 0);
			if (heavy) {
				const int slot = base + (int)__popcll(hm & lanemask_lt());
				heavy_queue[slot] = i;
				// the point's place in space (Morton code of its fine-grid cell), for a queue sorted by position (tuning.gather_sort)
				if (heavy_keys)
					heavy_keys[slot] = morton3_63((uint32_t)cell_coord(P.x, gf.mn[0], gf.inv_cell, gf.dim[0]),
					                              (uint32_t)cell_coord(P.y, gf.mn[1], gf.inv_cell, gf.dim[1]),
					                              (uint32_t)cell_coord(P.z, gf.mn[2], gf.inv_cell, gf.dim[2]));
			}
		}
	}
	if (!valid || heavy) return;
	const float r2 = radius * radius;
	SelLevel lv[SEL_LEVELS];
	int nlv = 0;
	// is l2 among the candidates of refinement level `upto`?
	auto in_levels = [&](float l2, int upto) {
		bool ok = l2 < r2;
		for (int k = 0; k < SEL_LEVELS; ++k)
			if (k < upto) ok = ok && sel_bin(l2, lv[k].lo, lv[k].scale) == lv[k].bin;
		return ok;
	};
	// pass 1
	int nc = 0;
	float max_dist2 = 0.0f;
	lv[0].lo = 0.0f; lv[0].scale = (float)SEL_BINS / r2; lv[0].bin = 0;
	for (int b = 0; b < SEL_BINS; ++b) hist[b * GBLOCK] = 0u;
	if (nph > 0)
		for_each_neighbour(gd, cell_start, pos, P, radius, [&](uint32_t, float l2) {
			if (l2 < r2) {
				++nc;
				if (l2 > max_dist2) max_dist2 = l2;
				++hist[sel_bin(l2, lv[0].lo, lv[0].scale) * GBLOCK];
			}
		});
	uint32_t tau = 0xFFFFFFFFu;  // bit pattern of the count-th smallest l2 (all candidates if nc <= count)
	int take_equal = 0x7fffffff;
	float rho_sel = radius;      // radius that certainly contains everything <= tau
	if (nc > count) {
		int need = count;        // rank (1-based) of the wanted element among the current candidates
		int pop = nc;            // current candidates
		float hi_edge = r2;
		for (;;) {
			// descend into the bin that holds the need-th candidate
			int b = 0, cum = 0;
			for (; b < SEL_BINS - 1; ++b) {
				int h = (int)hist[b * GBLOCK];
				if (cum + h >= need) break;
				cum += h;
			}
			need -= cum;
			pop = (int)hist[b * GBLOCK];
			const float width = 1.0f / lv[nlv].scale;
			const float lo_edge = lv[nlv].lo + (float)b * width;
			// upper bound of the squared distances in this bin, with slack for the rounding of the bin
			// function; the top bin also holds what the clamp folded into it: keep the parent's bound
			if (b < SEL_BINS - 1) hi_edge = fminf(hi_edge, lo_edge + width * 1.001f + fabsf(lo_edge) * 4e-6f);
			lv[nlv].bin = b;
			++nlv;
			rho_sel = fminf(radius, rt_sqrtf(hi_edge) * 1.001f);
			if (pop <= SEL_LIST || nlv == SEL_LEVELS || !(width > 0.0f)) break;
			lv[nlv].lo = lo_edge; lv[nlv].scale = (float)SEL_BINS / width; lv[nlv].bin = 0;
			for (int k = 0; k < SEL_BINS; ++k) hist[k * GBLOCK] = 0u;
			const int upto = nlv;
			for_each_neighbour(gd, cell_start, pos, P, rho_sel, [&](uint32_t, float l2) {
				if (in_levels(l2, upto)) ++hist[sel_bin(l2, lv[upto].lo, lv[upto].scale) * GBLOCK];
			});
		}
		if (pop <= SEL_LIST) {
			// rank the remaining candidates in LDS
			int m = 0;
			const int upto = nlv;
			for_each_neighbour(gd, cell_start, pos, P, rho_sel, [&](uint32_t, float l2) {
				if (in_levels(l2, upto) && m < SEL_LIST) { hist[m * GBLOCK] = __float_as_uint(l2); ++m; }
			});
			for (int a = 0; a < m; ++a) {
				const uint32_t v = hist[a * GBLOCK];
				int lt = 0, le = 0;
				for (int c = 0; c < m; ++c) {
					const uint32_t x = hist[c * GBLOCK];
					lt += x < v; le += x <= v;
				}
				if (lt < need && need <= le) { tau = v; take_equal = need - lt; break; }
			}
		} else {
			// more than SEL_LIST candidates inside one bin of the last level (heavily duplicated
			// distances): bisect on the bit pattern among them
			const int upto = nlv;
			uint32_t lo = 0u, hi = __float_as_uint(r2);
			while (lo < hi) {
				uint32_t mid = lo + ((hi - lo) >> 1);
				int c = 0;
				for_each_neighbour(gd, cell_start, pos, P, rho_sel, [&](uint32_t, float l2) {
					if (in_levels(l2, upto) && __float_as_uint(l2) <= mid) ++c;
				});
				if (c >= need) hi = mid; else lo = mid + 1;
			}
			tau = lo;
			int less = 0;
			for_each_neighbour(gd, cell_start, pos, P, rho_sel, [&](uint32_t, float l2) {
				if (in_levels(l2, upto) && __float_as_uint(l2) < tau) ++less;
			});
			take_equal = need - less;
		}
	}
	// pass 3: weighted sum over the selected photons (:178-197)
	f3 result = mk3(0, 0, 0);
	if (nc > 0) {
		const float denom2 = 2.0f * max_dist2;
		for_each_neighbour(gd, cell_start, pos, P, rho_sel, [&](uint32_t s, float l2) {
			if (!(l2 < r2)) return;
			uint32_t b = __float_as_uint(l2);
			bool sel = b < tau;
			if (!sel && b == tau && take_equal > 0) { sel = true; --take_equal; }
			if (!sel) return;
			float w = 1.0f - powf(gc.e, -gc.beta * (l2 / denom2));
			w *= gc.gauss_denom;
			w = gc.alpha * (1.0f - w);
			float4 a = payload[2 * (size_t)s], c = payload[2 * (size_t)s + 1];
			f3 energy = mk3(a.x, a.y, a.z), dir = mk3(a.w, c.x, c.y);
			float brdf = dot3(N, neg3(dir));
			brdf = fminf(fmaxf(brdf, 0.0f), 1.0f);
			result = result + (energy * brdf) * w;
		});
	}
	// result / (2*PI*max_dist2): 0 * (1/0) = NaN when nothing is in range, as the reference (:204)
	result = div3(result, gc.two_pi * max_dist2);
	st3(rgb + 3 * (size_t)i, result);
}

// The same estimate with ONE WAVE per shading point (the points k_photon_gather queued as heavy): the
// lanes stride over the photon spans with coalesced loads, the histogram and the candidate list are
// the wave's (LDS atomics), and the control flow of the selection is wave-uniform.  Same passes, same
// bins, same tie rule (photons at the k-th distance are taken in walk order) as the per-lane kernel;
// only the order of the floating-point sum differs (lane partial sums, then a butterfly).
#define RT_GATHER_WAVES 4
// PHASE 1: pass 1 alone, (count, max_dist2) per queue entry into `p1`; PHASE 2: the selection and the sum, pass 1's
// results read back; PHASE 0: everything in one kernel.  Two kernels because fused the kernel holds the live state of
// both halves (it spilled ~100 VGPRs and 220 SGPRs at 4 waves/SIMD), and the two halves want different occupancies:
// pass 1 streams photons, the selection is a chain of short dependent sweeps.
#define RT_GATHER_WAVES1 RT_GATHER_WAVES
#define RT_GATHER_WAVES2 RT_GATHER_WAVES
#define RT_GATHER_P1_WAVES 4
template <bool SHELL, int PHASE>
__global__ __launch_bounds__(PHASE == 1 ? 64 * RT_GATHER_P1_WAVES : 64) __attribute__((amdgpu_waves_per_eu(PHASE == 1 ? RT_GATHER_WAVES1 : PHASE == 2 ? RT_GATHER_WAVES2 : RT_GATHER_WAVES,
                                                                    PHASE == 1 ? RT_GATHER_WAVES1 : PHASE == 2 ? RT_GATHER_WAVES2 : RT_GATHER_WAVES)))
void k_photon_gather_wave(GridDesc gd, const uint32_t* __restrict__ cell_start,
                                                           const float4* __restrict__ pos,
                                                           const float4* __restrict__ payload, int64_t nph,
                                                           const float* __restrict__ qpos, const float* __restrict__ qnrm,
                                                           const int* __restrict__ heavy_queue,
                                                           const int* __restrict__ heavy_count, int count, float radius,
                                                           GatherConst gc, float* __restrict__ rgb, int xcd_chunk,
                                                           float2* __restrict__ p1, const int* __restrict__ sub_queue,
                                                           const int* __restrict__ sub_count, GridDesc gdc,
                                                           const uint32_t* __restrict__ cell_start_c,
                                                           const float4* __restrict__ pos_c, uint32_t medium, float pair_extent)
{
	__shared__ uint32_t s_hist[SEL_BINS];
	__shared__ uint32_t s_list[SEL_LIST], s_lslot[SEL_LIST];
	__shared__ float s_bl2[SEL_BELOW];        // photons below the count-th's bin: squared distance and sorted slot
	__shared__ uint32_t s_bslot[SEL_BELOW];
	__shared__ uint32_t s_tau;
	__shared__ int s_take;
	__shared__ uint32_t s_mark[PHASE == 1 ? RT_GATHER_P1_WAVES : 1][64];   // wave_visit_runs' byte table, one per wave
	uint32_t* mark = s_mark[PHASE == 1 ? (threadIdx.x >> 6) : 0];
	static_assert(SEL_BINS == 64, "one histogram bin per lane");
	const int lane = lane_id();
	// sub_queue (PHASE 2 only): positions in the heavy queue that k_photon_select_common left to this kernel
	const int nheavy = sub_queue ? *sub_count : *heavy_count;
	const float r2 = radius * radius;
	// XCD-aware order (rt_trace_dev.h: xcd_block_index): XCD x takes `xcd_chunk` consecutive queue entries of every
	// group of 8 * xcd_chunk.  Queue order is screen order, neighbouring points read nearly the same photons, and the
	// heavy points are bound by the bandwidth those reads get: this way they come from the XCD's own L2.
	// PHASE 1 runs several waves per workgroup, each on a point of its own: neighbours in the queue, hence in the frame,
	// read the same photons at the same time through one L1
	const int nw = PHASE == 1 ? RT_GATHER_P1_WAVES : 1, wv = PHASE == 1 ? (int)(threadIdx.x >> 6) : 0;
	// PHASE 1 with pair_extent > 0: a wave takes queue entries 2 j and 2 j + 1 per round -- together (wave_shell_sweep_pair)
	// when both walk the fine grid's shell and lie within pair_extent of each other, else one after the other as before
	const bool pairs = PHASE == 1 && SHELL && pair_extent > 0.0f && !sub_queue;
	const int per_round = pairs ? 2 : 1;
	const int nrounds = (nheavy + per_round - 1) / per_round;
	for (int round = (int)xcd_block_index(xcd_chunk) * nw + wv; round < nrounds; round += gridDim.x * nw) {
		bool coarse_known = false;   // the pair's cube on the coarse grid holds at most `medium` photons: so does each point's
		if (pairs && 2 * round + 1 < nheavy) {
			const int hq = 2 * round;
			const int ia = __builtin_amdgcn_readfirstlane(heavy_queue[hq]), ib = __builtin_amdgcn_readfirstlane(heavy_queue[hq + 1]);
			f3 A, B;
			A.x = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, qpos[3 * (size_t)ia + 0])));
			A.y = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, qpos[3 * (size_t)ia + 1])));
			A.z = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, qpos[3 * (size_t)ia + 2])));
			B.x = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, qpos[3 * (size_t)ib + 0])));
			B.y = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, qpos[3 * (size_t)ib + 1])));
			B.z = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, qpos[3 * (size_t)ib + 2])));
			const f3 C = mk3(0.5f * (A.x + B.x), 0.5f * (A.y + B.y), 0.5f * (A.z + B.z));
			const f3 da = A - C, db = B - C;
			const float en = bound_sqrtf(fmaxf(dot3(da, da), dot3(db, db))) * 1.0001f + radius * 1e-6f;
			bool both = en <= pair_extent;   // false for a NaN
			if (both && medium > 0u) {
				// neither point may be one that reads its (small) neighbourhood from the coarse grid instead: the cube about the
				// midpoint, widened by en, holds both points' cubes
				const float rho = (radius + en) * 1.0001f + 1e-30f;
				const int y0 = cell_coord(C.y - rho, gdc.mn[1], gdc.inv_cell, gdc.dim[1]), y1 = cell_coord(C.y + rho, gdc.mn[1], gdc.inv_cell, gdc.dim[1]);
				const int z0 = cell_coord(C.z - rho, gdc.mn[2], gdc.inv_cell, gdc.dim[2]), z1 = cell_coord(C.z + rho, gdc.mn[2], gdc.inv_cell, gdc.dim[2]);
				const int ny = y1 - y0 + 1, nrows = ny * (z1 - z0 + 1);
				uint32_t pop = 0;
				for (int row = lane; row < nrows; row += 64) {
					const int zr = row / ny, y = y0 + (row - zr * ny), z = z0 + zr;
					int x0, x1;
					if (row_chord(gdc, C, rho, y, z, x0, x1)) {
						const uint32_t rb = (uint32_t)((z * gdc.dim[1] + y) * gdc.dim[0]);
						pop += cell_start_c[rb + x1 + 1] - cell_start_c[rb + x0];
					}
				}
				both = wave_sum_u(pop) > medium;
				coarse_known = !both;
			}
			if (__builtin_amdgcn_readfirstlane(coarse_known ? 1 : 0)) {
				// a sparse pair: everything within radius + en of the midpoint read once from the coarse grid, tested against both points
				int nca_l = 0, ncb_l = 0;
				uint32_t mxa_u = 0u, mxb_u = 0u;
				const v2f PX = {A.x, B.x}, PY = {A.y, B.y}, PZ = {A.z, B.z};
				const float rho = radius + en;
				const int y0 = cell_coord(C.y - rho * 1.0001f, gdc.mn[1], gdc.inv_cell, gdc.dim[1]), y1 = cell_coord(C.y + rho * 1.0001f, gdc.mn[1], gdc.inv_cell, gdc.dim[1]);
				const int z0 = cell_coord(C.z - rho * 1.0001f, gdc.mn[2], gdc.inv_cell, gdc.dim[2]), z1 = cell_coord(C.z + rho * 1.0001f, gdc.mn[2], gdc.inv_cell, gdc.dim[2]);
				const int ny = y1 - y0 + 1, nrows = ny * (z1 - z0 + 1);
				for (int rbase = 0; rbase < nrows; rbase += 64) {
					const int row = rbase + lane;
					uint32_t s0 = 0, s1 = 0;
					if (row < nrows) {
						const int zr = row / ny, y = y0 + (row - zr * ny), z = z0 + zr;
						int x0, x1;
						if (row_chord(gdc, C, rho * 1.0001f + 1e-30f, y, z, x0, x1)) {
							const uint32_t rb = (uint32_t)((z * gdc.dim[1] + y) * gdc.dim[0]);
							s0 = cell_start_c[rb + x0]; s1 = cell_start_c[rb + x1 + 1];
						}
					}
					wave_visit_runs<true>(mark, pos_c, s0, s1, [&](const float4 q, uint32_t, bool act) {
						const v2f qx = {q.x, q.x}, qy = {q.y, q.y}, qz = {q.z, q.z};
						const v2f dx = qx - PX, dy = qy - PY, dz = qz - PZ;
						const v2f l2 = dx * dx + (dy * dy + dz * dz);   // dot3's order
						const bool ina = act && l2.x < r2, inb = act && l2.y < r2;
						nca_l += ina ? 1 : 0;
						ncb_l += inb ? 1 : 0;
						mxa_u = max(mxa_u, ina ? __float_as_uint(l2.x) : 0u);
						mxb_u = max(mxb_u, inb ? __float_as_uint(l2.y) : 0u);
					});
				}
				const int nca = wave_sum_i(nca_l), ncb = wave_sum_i(ncb_l);
				const float mxa = wave_max_f(__uint_as_float(mxa_u)), mxb = wave_max_f(__uint_as_float(mxb_u));
				if (lane == 0) { p1[hq] = make_float2(__int_as_float(nca), mxa); p1[hq + 1] = make_float2(__int_as_float(ncb), mxb); }
				continue;
			}
			if (__builtin_amdgcn_readfirstlane(both ? 1 : 0)) {
				int common_l = 0, nca_l = 0, ncb_l = 0;
				uint32_t mxa_u = 0u, mxb_u = 0u;
				float bound_l = 0.0f;
				wave_shell_sweep_pair(mark, gd, cell_start, pos, C, A, B, radius, en, r2, common_l, nca_l, ncb_l, mxa_u, mxb_u, bound_l);
				const int common = wave_sum_i(common_l);
				const int nca = wave_sum_i(nca_l) + common, ncb = wave_sum_i(ncb_l) + common;
				const float mxa = wave_max_f(__uint_as_float(mxa_u)), mxb = wave_max_f(__uint_as_float(mxb_u));
				const float bw = wave_max_f(bound_l);
				const float reach = bound_sqrtf(bw) * 1.0001f + en;
				const float far_in = bw > 0.0f ? reach * reach * 1.0001f : 0.0f;   // no interior photon is farther than this from A or B
				if (!(far_in > mxa) && !(far_in > mxb)) {
					if (lane == 0) { p1[hq] = make_float2(__int_as_float(nca), mxa); p1[hq + 1] = make_float2(__int_as_float(ncb), mxb); }
					continue;
				}
			}
		}
		for (int sub = 0; sub < per_round; ++sub) {
		const int hq = per_round * round + sub;
		if (hq >= nheavy) break;
		const int h = sub_queue ? __builtin_amdgcn_readfirstlane(sub_queue[hq]) : hq;
		const int i = __builtin_amdgcn_readfirstlane(heavy_queue[h]);
		f3 P, N;
		P.x = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, qpos[3 * (size_t)i + 0])));
		P.y = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, qpos[3 * (size_t)i + 1])));
		P.z = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, qpos[3 * (size_t)i + 2])));
		N = ld3(qnrm + 3 * (size_t)i);
		SelLevel lv[SEL_LEVELS];
		int nlv = 0;
		auto in_levels = [&](float l2, int upto) {
			bool ok = l2 < r2;
			for (int k = 0; k < SEL_LEVELS; ++k)
				if (k < upto) ok = ok && sel_bin(l2, lv[k].lo, lv[k].scale) == lv[k].bin;
			return ok;
		};
		// pass 1: the number of photons within the radius and max_dist2 over all of them; interiors are counted
		// from the cell table, only the boundary shell is read (wave_shell_sweep)
		int nc;
		float max_dist2;
		if constexpr (PHASE != 2) {
			int nc_l = 0;
			float mx_l = 0.0f, bound_l = 0.0f;
			// A point with few photons around it pays for the fine grid's rows (17 x 17 of them, two calls of the run
			// visitor per 64) and gains nothing from the counted interiors: if its cube on the COARSE grid (5 x 5 rows)
			// holds at most `medium` photons, all of them are read from there in one go.
			bool coarse_pass = coarse_known;
			if (SHELL && PHASE == 1 && medium > 0u && !coarse_known) {
				const float rho = radius * 1.0001f + 1e-30f;
				const int y0 = cell_coord(P.y - rho, gdc.mn[1], gdc.inv_cell, gdc.dim[1]), y1 = cell_coord(P.y + rho, gdc.mn[1], gdc.inv_cell, gdc.dim[1]);
				const int z0 = cell_coord(P.z - rho, gdc.mn[2], gdc.inv_cell, gdc.dim[2]), z1 = cell_coord(P.z + rho, gdc.mn[2], gdc.inv_cell, gdc.dim[2]);
				const int ny = y1 - y0 + 1, nrows = ny * (z1 - z0 + 1);
				uint32_t pop = 0;
				for (int row = lane; row < nrows; row += 64) {
					const int zr = row / ny, y = y0 + (row - zr * ny), z = z0 + zr;
					int x0, x1;
					if (row_chord(gdc, P, rho, y, z, x0, x1)) {
						const uint32_t rb = (uint32_t)((z * gdc.dim[1] + y) * gdc.dim[0]);
						pop += cell_start_c[rb + x1 + 1] - cell_start_c[rb + x0];
					}
				}
				coarse_pass = wave_sum_u(pop) <= medium;
			}
			if (coarse_pass) {
				for_each_neighbour_wave<true>(mark, gdc, cell_start_c, pos_c, P, radius, [&](uint32_t, float l2, bool act) {
					const bool in = act && l2 < r2;
					nc_l += in ? 1 : 0;
					mx_l = __uint_as_float(max(__float_as_uint(mx_l), in ? __float_as_uint(l2) : 0u));
				});
			} else if (SHELL) {
				wave_shell_sweep<false>(mark, gd, cell_start, pos, P, radius, r2, 0.0f, nc_l, mx_l, bound_l);
			} else {
				for_each_neighbour_wave(mark, gd, cell_start, pos, P, radius, [&](uint32_t, float l2, bool act) {
					if (act && l2 < r2) { ++nc_l; mx_l = fmaxf(mx_l, l2); }
				});
			}
			nc = wave_sum_i(nc_l);
			max_dist2 = wave_max_f(mx_l);
			if (SHELL && wave_max_f(bound_l) > max_dist2) {   // rare: an interior run could hold the farthest candidate
				wave_shell_sweep<true>(mark, gd, cell_start, pos, P, radius, r2, max_dist2, nc_l, mx_l, bound_l);
				max_dist2 = wave_max_f(mx_l);
			}
			if constexpr (PHASE == 1) {
				if (lane == 0) p1[h] = make_float2(__int_as_float(nc), max_dist2);
				continue;
			}
		} else {
			const float2 r = p1[h];
			nc = __builtin_amdgcn_readfirstlane(__float_as_int(r.x));
			max_dist2 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, r.y)));
		}
		uint32_t tau = 0xFFFFFFFFu;
		int take_equal = 0x7fffffff;
		float rho_sel = radius;
		f3 result = mk3(0, 0, 0);
		bool summed = false;
		if (nc > count) {
			// level-0 histogram.  Photons lie on surfaces, so their number grows like l2: the count-th is
			// expected near r2 * count / nc.  The histogram first covers [0, 4x that) and reads only the
			// cells within its upper edge; if the count-th is not below the edge after all (its bin would
			// be the top one, whose population the short walk does not know), the range is widened fourfold.
			float top = fminf(r2, r2 * (4.0f * (float)count / (float)nc));
			for (;;) {
				lv[0].lo = 0.0f; lv[0].scale = (float)SEL_BINS / top; lv[0].bin = 0;
				__syncthreads();  // earlier readers are done with the shared state
				s_hist[lane] = 0u;
				__syncthreads();
				const float rho0 = top < r2 ? fminf(radius, rt_sqrtf(top) * 1.001f) : radius;
				for_each_neighbour_wave(mark, gd, cell_start, pos, P, rho0, [&](uint32_t, float l2, bool act) {
					if (act && l2 < r2) atomicAdd(&s_hist[sel_bin(l2, lv[0].lo, lv[0].scale)], 1u);
				});
				__syncthreads();
				if (!(top < r2)) break;
				const int below = wave_sum_i(lane < SEL_BINS - 1 ? (int)s_hist[lane] : 0);
				if (below >= count) break;
				top = fminf(r2, top * 4.0f);
			}
			int need = count;
			int pop = nc;
			float hi_edge = r2;
			for (;;) {
				// the bin that holds the need-th candidate: one bin per lane, prefix sums across the lanes
				const int hh = (int)s_hist[lane];
				int pre = hh;
				for (int off = 1; off < 64; off <<= 1) {
					const int y = __shfl_up(pre, off);
					if (lane >= off) pre += y;
				}
				const uint64_t ge = __ballot(pre >= need) & ~(1ull << (SEL_BINS - 1));
				const int b = ge ? __ffsll((unsigned long long)ge) - 1 : SEL_BINS - 1;
				need -= __builtin_amdgcn_readlane(pre, b) - __builtin_amdgcn_readlane(hh, b);
				pop = __builtin_amdgcn_readlane(hh, b);
				const float width = 1.0f / lv[nlv].scale;
				const float lo_edge = lv[nlv].lo + (float)b * width;
				if (b < SEL_BINS - 1) hi_edge = fminf(hi_edge, lo_edge + width * 1.001f + fabsf(lo_edge) * 4e-6f);
				lv[nlv].bin = b;
				++nlv;
				rho_sel = fminf(radius, rt_sqrtf(hi_edge) * 1.001f);
				if (pop <= SEL_LIST || nlv == SEL_LEVELS || !(width > 0.0f)) break;
				lv[nlv].lo = lo_edge; lv[nlv].scale = (float)SEL_BINS / width; lv[nlv].bin = 0;
				__syncthreads();
				s_hist[lane] = 0u;
				__syncthreads();
				const int upto = nlv;
				for_each_neighbour_wave(mark, gd, cell_start, pos, P, rho_sel, [&](uint32_t, float l2, bool act) {
					if (act && in_levels(l2, upto)) atomicAdd(&s_hist[sel_bin(l2, lv[upto].lo, lv[upto].scale)], 1u);
				});
				__syncthreads();
			}
			if (pop <= SEL_LIST) {
				// One sweep for the rest: a photon in an earlier bin than the count-th's (at the first level where they
				// differ) is certainly selected and is summed (:178-197) at once; the photons of the count-th's own bin
				// go to a list in LDS with their slots, are ranked there, and the selected ones are summed from the list.
				const int upto = nlv;
				if (lane == 0) { s_tau = 0xFFFFFFFFu; s_take = 0x7fffffff; }
				__syncthreads();
				const float denom2 = 2.0f * max_dist2;
				auto contribution = [&](uint32_t s, float l2) {
					float w = 1.0f - powf(gc.e, -gc.beta * (l2 / denom2));
					w *= gc.gauss_denom;
					w = gc.alpha * (1.0f - w);
					const float4 a = payload[2 * (size_t)s], c = payload[2 * (size_t)s + 1];
					const f3 energy = mk3(a.x, a.y, a.z), dir = mk3(a.w, c.x, c.y);
					float brdf = dot3(N, neg3(dir));
					brdf = fminf(fmaxf(brdf, 0.0f), 1.0f);
					return (energy * brdf) * w;
				};
				int m_in = 0, m_below = 0;   // wave-uniform list lengths; slots by ballot rank, so both lists are in walk order
				for_each_neighbour_wave(mark, gd, cell_start, pos, P, rho_sel, [&](uint32_t s, float l2, bool act) {
					int cls = 2;   // -1 below the count-th's bin, 0 in it, 1 above, 2 not a candidate
					if (act && l2 < r2) {
						cls = 0;
						for (int k = 0; k < SEL_LEVELS; ++k)
							if (k < upto && cls == 0) {
								const int bk = sel_bin(l2, lv[k].lo, lv[k].scale);
								cls = bk < lv[k].bin ? -1 : bk > lv[k].bin ? 1 : 0;
							}
					}
					const uint64_t im = __ballot(cls == 0), bm = __ballot(cls == -1);
					if (cls == 0) {
						const int at = m_in + (int)__popcll(im & lanemask_lt());
						if (at < SEL_LIST) { s_list[at] = __float_as_uint(l2); s_lslot[at] = s; }
					}
					if (cls == -1) {
						// summed after the sweep, 64 at a time (here the weight's powf would run for a few lanes per call);
						// what does not fit the list (count > SEL_BELOW) is summed on the spot
						const int at = m_below + (int)__popcll(bm & lanemask_lt());
						if (at < SEL_BELOW) { s_bl2[at] = l2; s_bslot[at] = s; }
						else result = result + contribution(s, l2);
					}
					m_in += (int)__popcll(im);
					m_below += (int)__popcll(bm);
				});
				__syncthreads();
				for (int i = lane; i < min(m_below, SEL_BELOW); i += 64) result = result + contribution(s_bslot[i], s_bl2[i]);
				const int m = min(m_in, SEL_LIST);
				uint32_t v = 0xFFFFFFFFu;
				if (lane < m) {
					v = s_list[lane];
					int lt = 0, le = 0;
					for (int c = 0; c < m; ++c) {
						const uint32_t x = s_list[c];
						lt += x < v; le += x <= v;
					}
					// lanes holding equal values agree on (tau, take_equal): any of them may publish it
					if (lt < need && need <= le) { s_tau = v; s_take = need - lt; }
				}
				__syncthreads();
				tau = s_tau; take_equal = s_take;
				{   // of the photons AT tau the first take_equal of the list are taken
					const bool eq = lane < m && v == tau;
					const uint64_t em = __ballot(eq);
					const bool sel = lane < m && (v < tau || (eq && (int)__popcll(em & lanemask_lt()) < take_equal));
					if (sel) result = result + contribution(s_lslot[lane], __uint_as_float(v));
				}
				result.x = wave_sum_f(result.x); result.y = wave_sum_f(result.y); result.z = wave_sum_f(result.z);
				summed = true;
			} else {
				const int upto = nlv;
				uint32_t lo = 0u, hi = __float_as_uint(r2);
				while (lo < hi) {
					uint32_t mid = lo + ((hi - lo) >> 1);
					int c = 0;
					for_each_neighbour_wave(mark, gd, cell_start, pos, P, rho_sel, [&](uint32_t, float l2, bool act) {
						if (act && in_levels(l2, upto) && __float_as_uint(l2) <= mid) ++c;
					});
					c = wave_sum_i(c);
					if (c >= need) hi = mid; else lo = mid + 1;
				}
				tau = lo;
				int less = 0;
				for_each_neighbour_wave(mark, gd, cell_start, pos, P, rho_sel, [&](uint32_t, float l2, bool act) {
					if (act && in_levels(l2, upto) && __float_as_uint(l2) < tau) ++less;
				});
				take_equal = need - wave_sum_i(less);
			}
		}
		// pass 3 (unless the sweep above has summed already)
		if (nc > 0 && !summed) {
			const float denom2 = 2.0f * max_dist2;
			int taken = 0;  // photons at tau already taken, in walk order (wave-uniform)
			for_each_neighbour_wave(mark, gd, cell_start, pos, P, rho_sel, [&](uint32_t s, float l2, bool act) {
				const uint32_t b = __float_as_uint(l2);
				const bool inr = act && l2 < r2;
				const bool eq = inr && b == tau;
				const uint64_t em = __ballot(eq);
				bool sel = inr && b < tau;
				if (eq) sel = (long long)taken + (long long)__popcll(em & lanemask_lt()) < (long long)take_equal;
				// saturating: take_equal is INT_MAX when every photon within the radius is taken
				taken = (int)min((long long)taken + (long long)__popcll(em), (long long)0x7fffffff);
				if (!sel) return;
				float w = 1.0f - powf(gc.e, -gc.beta * (l2 / denom2));
				w *= gc.gauss_denom;
				w = gc.alpha * (1.0f - w);
				float4 a = payload[2 * (size_t)s], c = payload[2 * (size_t)s + 1];
				f3 energy = mk3(a.x, a.y, a.z), dir = mk3(a.w, c.x, c.y);
				float brdf = dot3(N, neg3(dir));
				brdf = fminf(fmaxf(brdf, 0.0f), 1.0f);
				result = result + (energy * brdf) * w;
			});
			result.x = wave_sum_f(result.x); result.y = wave_sum_f(result.y); result.z = wave_sum_f(result.z);
		}
		result = div3(result, gc.two_pi * max_dist2);
		if (lane == 0) st3(rgb + 3 * (size_t)i, result);
		}   // sub
	}
}

// The selection and the sum for the USUAL heavy point, after pass 1 (PHASE 1 above): more photons within the radius than
// wanted, the count-th distance inside the first histogram's range, at most SEL_LIST photons in its bin.  Two sweeps:
// the level-0 histogram over the cells within the radius expected to hold 4 x count photons, then the sweep that sums
// what lies below the count-th's bin and lists the bin itself (ranked in LDS, the selected ones summed from the list).
// Every other case (fewer photons than wanted, an estimate that fell short, a crowded bin) is queued for PHASE 2 of
// k_photon_gather_wave, which handles all of them.  Same bins, same tie rule, same results; the point of a kernel of its
// own is its size: the general kernel carries eight inlined sweeps, holds 220 VGPRs and spills at 4 waves per SIMD.
#define RT_GATHER_WAVES2C 6
#define RT_GATHER_MARKS_COMMON 1
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RT_GATHER_WAVES2C, RT_GATHER_WAVES2C)))
void k_photon_select_common(GridDesc gd, const uint32_t* __restrict__ cell_start, const float4* __restrict__ pos,
                            const float4* __restrict__ payload, const float* __restrict__ qpos, const float* __restrict__ qnrm,
                            const int* __restrict__ heavy_queue, const int* __restrict__ heavy_count, int count, float radius,
                            GatherConst gc, float* __restrict__ rgb, int xcd_chunk, const float2* __restrict__ p1,
                            int* __restrict__ sub_queue, int* __restrict__ sub_count)
{
	__shared__ uint32_t s_hist[SEL_BINS];
	__shared__ uint32_t s_list[SEL_LIST], s_lslot[SEL_LIST];
	__shared__ float s_bl2[SEL_BELOW];
	__shared__ uint32_t s_bslot[SEL_BELOW];
	__shared__ uint32_t s_tau;
	__shared__ int s_take;
	__shared__ uint32_t s_mark[64];
	uint32_t* mark = s_mark;
	const int lane = lane_id();
	const int nheavy = *heavy_count;
	const float r2 = radius * radius;
	for (int h = (int)xcd_block_index(xcd_chunk); h < nheavy; h += gridDim.x) {
		const int i = __builtin_amdgcn_readfirstlane(heavy_queue[h]);
		f3 P, N;
		P.x = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, qpos[3 * (size_t)i + 0])));
		P.y = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, qpos[3 * (size_t)i + 1])));
		P.z = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, qpos[3 * (size_t)i + 2])));
		N = ld3(qnrm + 3 * (size_t)i);
		const float2 r = p1[h];
		const int nc = __builtin_amdgcn_readfirstlane(__float_as_int(r.x));
		const float max_dist2 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, r.y)));
		bool common = true;
		// nc <= count: every photon within the radius is taken -- the second sweep alone, with all of them "below the bin"
		float scale = (float)SEL_BINS / r2, rho_sel = radius;
		int b = SEL_BINS, need = 0;
		__syncthreads();   // earlier readers are done with the shared state
		if (lane == 0) { s_tau = 0xFFFFFFFFu; s_take = 0x7fffffff; }
		if (nc > count) {
			// level-0 histogram over [0, top): the count-th is expected near r2 * count / nc (photons lie on surfaces)
			// If the count-th is not below the range after all (the top bin also counts what lies beyond it), the range
			// is widened fourfold and the histogram redone: the density around a point is not uniform.
			float top = fminf(r2, r2 * (4.0f * (float)count / (float)nc));
			int hh;
			for (;;) {
				scale = (float)SEL_BINS / top;
				__syncthreads();
				s_hist[lane] = 0u;
				__syncthreads();
				const float rho0 = top < r2 ? fminf(radius, rt_sqrtf(top) * 1.001f) : radius;
				for_each_neighbour_wave<(RT_GATHER_MARKS_COMMON != 0)>(mark, gd, cell_start, pos, P, rho0, [&](uint32_t, float l2, bool act) {
					if (act && l2 < r2) atomicAdd(&s_hist[sel_bin(l2, 0.0f, scale)], 1u);
				});
				__syncthreads();
				hh = (int)s_hist[lane];
				if (!(top < r2) || wave_sum_i(lane < SEL_BINS - 1 ? hh : 0) >= count) break;
				top = fminf(r2, top * 4.0f);
			}
			const int pre = (int)wave_scan_add_u32((uint32_t)hh);
			const uint64_t ge = __ballot(pre >= count) & ~(1ull << (SEL_BINS - 1));
			b = ge ? __ffsll((unsigned long long)ge) - 1 : SEL_BINS - 1;
			need = count - (__builtin_amdgcn_readlane(pre, b) - __builtin_amdgcn_readlane(hh, b));
			if (__builtin_amdgcn_readlane(hh, b) > SEL_LIST) common = false;
			const float width = 1.0f / scale;
			const float lo_edge = (float)b * width;
			float hi_edge = r2;
			if (b < SEL_BINS - 1) hi_edge = fminf(hi_edge, lo_edge + width * 1.001f + fabsf(lo_edge) * 4e-6f);
			rho_sel = fminf(radius, rt_sqrtf(hi_edge) * 1.001f);
		}
		__syncthreads();
		{
			{
				if (common) {
					const float denom2 = 2.0f * max_dist2;
					auto contribution = [&](uint32_t s, float l2) {
						float w = 1.0f - powf(gc.e, -gc.beta * (l2 / denom2));
						w *= gc.gauss_denom;
						w = gc.alpha * (1.0f - w);
						const float4 a = payload[2 * (size_t)s], c = payload[2 * (size_t)s + 1];
						const f3 energy = mk3(a.x, a.y, a.z), dir = mk3(a.w, c.x, c.y);
						float brdf = dot3(N, neg3(dir));
						brdf = fminf(fmaxf(brdf, 0.0f), 1.0f);
						return (energy * brdf) * w;
					};
					f3 result = mk3(0, 0, 0);
					int m_in = 0, m_below = 0;   // wave-uniform list lengths; slots by ballot rank: both lists in walk order
					for_each_neighbour_wave<(RT_GATHER_MARKS_COMMON != 0)>(mark, gd, cell_start, pos, P, rho_sel, [&](uint32_t s, float l2, bool act) {
						const bool cand = act && l2 < r2;
						const int bk = sel_bin(l2, 0.0f, scale);
						const bool in = cand && bk == b, below = cand && bk < b;
						const uint64_t im = __ballot(in), bm = __ballot(below);
						if (in) {
							const int at = m_in + (int)__popcll(im & lanemask_lt());
							if (at < SEL_LIST) { s_list[at] = __float_as_uint(l2); s_lslot[at] = s; }
						}
						if (below) {
							const int at = m_below + (int)__popcll(bm & lanemask_lt());
							if (at < SEL_BELOW) { s_bl2[at] = l2; s_bslot[at] = s; }
							else result = result + contribution(s, l2);
						}
						m_in += (int)__popcll(im);
						m_below += (int)__popcll(bm);
					});
					__syncthreads();
					for (int k = lane; k < min(m_below, SEL_BELOW); k += 64) result = result + contribution(s_bslot[k], s_bl2[k]);
					const int m = min(m_in, SEL_LIST);
					uint32_t v = 0xFFFFFFFFu;
					if (lane < m) {
						v = s_list[lane];
						int lt = 0, le = 0;
						for (int c = 0; c < m; ++c) {
							const uint32_t x = s_list[c];
							lt += x < v; le += x <= v;
						}
						if (lt < need && need <= le) { s_tau = v; s_take = need - lt; }
					}
					__syncthreads();
					const uint32_t tau = s_tau;
					const int take_equal = s_take;
					const bool eq = lane < m && v == tau;
					const uint64_t em = __ballot(eq);
					const bool sel = lane < m && (v < tau || (eq && (int)__popcll(em & lanemask_lt()) < take_equal));
					if (sel) result = result + contribution(s_lslot[lane], __uint_as_float(v));
					result.x = wave_sum_f(result.x); result.y = wave_sum_f(result.y); result.z = wave_sum_f(result.z);
					result = div3(result, gc.two_pi * max_dist2);
					if (lane == 0) st3(rgb + 3 * (size_t)i, result);
				}
			}
		}
		if (!common && lane == 0) sub_queue[atomicAdd(sub_count, 1)] = h;
	}
}

// PhotonMap::lookup (rt_photon_map.cpp:126-142): the energies of the photons with l2 < r^2, one wave per point:
// lane partial sums in walk order, then a butterfly (deterministic, but not the reference's index order).
__global__ __launch_bounds__(64) void k_photon_lookup(GridDesc gd, const uint32_t* __restrict__ cell_start,
                                                      const float4* __restrict__ pos, const float4* __restrict__ payload,
                                                      const float* __restrict__ qpos, int npts, float radius,
                                                      float* __restrict__ rgb)
{
	__shared__ uint32_t s_mark1[64];
	uint32_t* mark = s_mark1;
	const float r2 = radius * radius;
	for (int i = blockIdx.x; i < npts; i += gridDim.x) {
		f3 P;
		P.x = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, qpos[3 * (size_t)i + 0])));
		P.y = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, qpos[3 * (size_t)i + 1])));
		P.z = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, qpos[3 * (size_t)i + 2])));
		f3 sum = mk3(0, 0, 0);
		for_each_neighbour_wave(mark, gd, cell_start, pos, P, radius, [&](uint32_t s, float l2, bool act) {
			if (act && l2 < r2) {
				const float4 a = payload[2 * (size_t)s];
				sum = sum + mk3(a.x, a.y, a.z);
			}
		});
		sum.x = wave_sum_f(sum.x); sum.y = wave_sum_f(sum.y); sum.z = wave_sum_f(sum.z);
		if (lane_id() == 0) st3(rgb + 3 * (size_t)i, sum);
	}
}

// sorted (position | index, payload) records back to the 9-float layout in photon-index order
__global__ __launch_bounds__(256) void k_photon_unpermute(const float4* __restrict__ pos, const float4* __restrict__ payload,
                                                          int64_t n, PhotonRec* __restrict__ out)
{
	int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (s >= n) return;
	const float4 p = pos[s], a = payload[2 * s], c = payload[2 * s + 1];
	PhotonRec r;
	r.p[0] = p.x; r.p[1] = p.y; r.p[2] = p.z;
	r.e[0] = a.x; r.e[1] = a.y; r.e[2] = a.z;
	r.d[0] = a.w; r.d[1] = c.x; r.d[2] = c.y;
	out[__float_as_uint(p.w)] = r;
}

// census: the photons the mandatory first pass of all points reads (the cells of each point's cube)
__global__ __launch_bounds__(256) void k_photon_census(GridDesc gd, const uint32_t* __restrict__ cell_start,
                                                       const float* __restrict__ qpos, int npts, float radius,
                                                       unsigned long long* __restrict__ visits)
{
	int i = blockIdx.x * 256 + threadIdx.x;
	unsigned long long v = 0;
	if (i < npts) v = cube_population(gd, cell_start, ld3(qpos + 3 * (size_t)i), radius);
	for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
	if (lane_id() == 0 && v) atomicAdd(visits, v);
}

float ord2f(uint32_t o)
{
	uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
	float f;
	memcpy(&f, &u, 4);
	return f;
}

}  // namespace

int rt_photon_emit(rt_hip_ctx* ctx, int per_light, int64_t first, int64_t count, float* d_out, int64_t capacity,
                   int64_t* out_stored)
{
	*out_stored = 0;
	if (count == 0) return RT_HIP_OK;
	SceneView sv = rt_scene_view(ctx);
	if (ctx->cfg.geometry != RT_HIP_TRIANGLES) { rt_set_error("photon emission needs triangle geometry"); return RT_HIP_ERR_STATE; }
	int64_t total = (int64_t)per_light * sv.num_lights;
	if (first + count > total) { rt_set_error("photon range [%lld,%lld) exceeds lights x per_light = %lld", (long long)first, (long long)(first + count), (long long)total); return RT_HIP_ERR_ARG; }
	if (count > 0x7FFFFFF0ll) { rt_set_error("more than 2^31 photons per call"); return RT_HIP_ERR_LIMIT; }
	// a photon is stored at most once (emit_photon returns after the store, rt_photon_map.cpp:44-52)
	if (capacity < count) { rt_set_error("photon output capacity %lld < count %lld", (long long)capacity, (long long)count); return RT_HIP_ERR_ARG; }
	hipStream_t st = ctx->stream;
	RT_CHECK(ctx->d_photon_tmp.reserve((size_t)count * sizeof(PhotonRec)));
	RT_CHECK(ctx->d_keys.reserve((size_t)count * 4));
	RT_CHECK(ctx->d_rank.reserve((size_t)count * 4));
	RT_CHECK(ctx->d_counts.reserve(2 * 256 * 4));
	uint32_t* digit = ctx->d_keys.as<uint32_t>();
	uint32_t* dest = ctx->d_rank.as<uint32_t>();
	uint32_t* totals = ctx->d_counts.as<uint32_t>();
	size_t lds = rt_trace_lds(sv, BLOCK);
	hipLaunchKernelGGL(k_photon_emit, dim3(rt_trace_blocks(ctx, count, BLOCK)), dim3(BLOCK), lds, st, sv, per_light, first, count,
	                   (uint32_t)ctx->cfg.seed, (uint32_t)(ctx->cfg.seed >> 32), ctx->cfg.photon_max_dist,
	                   ctx->cfg.photon_absorb, ctx->d_photon_tmp.as<PhotonRec>(), digit);
	RT_HIP_CHECK(hipGetLastError());
	// digit array is clobbered by multisplit's scratch? no: multisplit reads digit, writes dest/hist only
	RT_CHECK(rt_multisplit256(ctx, digit, (int)count, dest, totals));
	hipLaunchKernelGGL(k_photon_compact, dim3(nblocks(count, 256)), dim3(256), 0, st, ctx->d_photon_tmp.as<PhotonRec>(),
	                   digit, dest, count, (PhotonRec*)d_out);
	RT_HIP_CHECK(hipGetLastError());
	uint32_t stored = 0;
	RT_HIP_CHECK(hipMemcpyAsync(&stored, totals, 4, hipMemcpyDeviceToHost, st));
	RT_HIP_CHECK(hipStreamSynchronize(st));
	*out_stored = stored;
	return RT_HIP_OK;
}

// One uniform grid over the photon bounds [mn, mx]: cell enlarged until the grid has at most max_cells cells; photons
// counting-sorted by cell (rows along x are contiguous in the sorted arrays), cell_start by lower bound.
// One EMPTY cell is added on every side: the border cells of a grid also receive what cell_coord clamps into them, so
// the gather treats them as reaching to infinity and never as interior to a sphere -- and without the margin the
// photons on the scene's floor and outer walls, most of them, sit exactly in those cells.
static int build_one_grid(rt_hip_ctx* ctx, const PhotonRec* ph, int64_t n, const float* mn, const float* mx, float cell,
                          double max_cells, DevBuf& d_pos, DevBuf& d_payload, DevBuf& d_cell_start, int* dim_out, float* cell_out,
                          float* min_out)
{
	hipStream_t st = ctx->stream;
	int dim[3];
	for (;;) {
		double cells = 1;
		for (int k = 0; k < 3; ++k) {
			dim[k] = std::max(1, (int)std::floor((mx[k] - mn[k]) / cell) + 1) + 2;
			cells *= dim[k];
		}
		if (cells <= max_cells) break;
		cell *= 1.5f;
	}
	const int ncells = dim[0] * dim[1] * dim[2];
	GridDesc gd;
	for (int k = 0; k < 3; ++k) { gd.mn[k] = mn[k] - cell; min_out[k] = gd.mn[k]; gd.dim[k] = dim[k]; dim_out[k] = dim[k]; }
	gd.cell = cell; gd.inv_cell = 1.0f / cell;
	*cell_out = cell;
	RT_CHECK(ctx->d_morton.reserve((size_t)n * 8));
	RT_CHECK(ctx->d_morton_tmp.reserve((size_t)n * 8));
	RT_CHECK(ctx->d_order.reserve((size_t)n * 4));
	RT_CHECK(ctx->d_order_tmp.reserve((size_t)n * 4));
	uint64_t* keys = ctx->d_morton.as<uint64_t>();
	uint32_t* order = ctx->d_order.as<uint32_t>();
	hipLaunchKernelGGL(k_photon_cell_keys, dim3(nblocks(n, 256)), dim3(256), 0, st, ph, n, gd, keys, order);
	RT_HIP_CHECK(hipGetLastError());
	int bits = 1;
	while ((1ll << bits) < ncells) ++bits;
	RT_CHECK(rt_radix_sort_pairs64(ctx, keys, ctx->d_morton_tmp.as<uint64_t>(), order, ctx->d_order_tmp.as<uint32_t>(), (int)n, bits));
	RT_CHECK(d_pos.reserve((size_t)n * 16));
	RT_CHECK(d_payload.reserve((size_t)n * 32));
	RT_CHECK(d_cell_start.reserve((size_t)(ncells + 2) * 4));
	hipLaunchKernelGGL(k_photon_permute, dim3(nblocks(n, 256)), dim3(256), 0, st, ph, order, keys, n, ncells, d_pos.as<float4>(),
	                   d_payload.as<float4>());
	hipLaunchKernelGGL(k_photon_cell_start, dim3(nblocks((int64_t)ncells + 1, 256)), dim3(256), 0, st, keys, n, ncells,
	                   d_cell_start.as<uint32_t>());
	RT_HIP_CHECK(hipGetLastError());
	return RT_HIP_OK;
}

#define RT_FINE_CELL_CAP_LOG2 26
int rt_photon_build_grid(rt_hip_ctx* ctx, const float* d_photons, int64_t n)
{
	hipStream_t st = ctx->stream;
	ctx->num_photons = 0;
	if (n > 0x7FFFFFF0ll) { rt_set_error("more than 2^31 photons"); return RT_HIP_ERR_LIMIT; }
	if (n == 0) return RT_HIP_OK;
	const PhotonRec* ph = (const PhotonRec*)d_photons;
	RT_CHECK(ctx->d_bounds.reserve(64));
	uint32_t init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
	RT_HIP_CHECK(hipMemcpyAsync(ctx->d_bounds.p, init, sizeof(init), hipMemcpyHostToDevice, st));
	hipLaunchKernelGGL(k_photon_bounds, dim3(nblocks(n, 256)), dim3(256), 0, st, ph, n, ctx->d_bounds.as<uint32_t>());
	uint32_t ob[6];
	RT_HIP_CHECK(hipMemcpyAsync(ob, ctx->d_bounds.p, sizeof(ob), hipMemcpyDeviceToHost, st));
	RT_HIP_CHECK(hipStreamSynchronize(st));
	float mn[3], mx[3];
	for (int k = 0; k < 3; ++k) { mn[k] = ord2f(ob[k]); mx[k] = ord2f(ob[3 + k]); }
	const float radius = ctx->cfg.photon_radius > 0.0f ? ctx->cfg.photon_radius : 1.0f;
	// Two grids over the same photons.  Coarse, cell = radius / 2: what the lane-per-point gather walks (a point with
	// few photons nearby pays per grid row, so rows must be few).  Fine, cell = radius / 8: what the wave-per-point
	// gather of the photon-rich points walks -- cells entirely inside the sphere are counted from the cell table and
	// only the boundary shell is read, and the selection passes read a neighbourhood as small as the k-th distance.
	RT_CHECK(build_one_grid(ctx, ph, n, mn, mx, radius * 0.5f, (double)(1 << 22), ctx->d_photons, ctx->d_photon_sorted, ctx->d_cell_start,
	                        ctx->photon_grid_dim, &ctx->photon_cell, ctx->photon_grid_min));
	RT_CHECK(build_one_grid(ctx, ph, n, mn, mx, radius / (float)ctx->tuning.gather_fine_cells, (double)(1ll << RT_FINE_CELL_CAP_LOG2), ctx->d_photons_f, ctx->d_photon_sorted_f,
	                        ctx->d_cell_start_f, ctx->photon_grid_dim_f, &ctx->photon_cell_f, ctx->photon_grid_min_f));
	RT_HIP_CHECK(hipStreamSynchronize(st));
	ctx->num_photons = n;
	return RT_HIP_OK;
}

int rt_photon_gather(rt_hip_ctx* ctx, const float* d_pos, const float* d_nrm, int npts, int count, float radius,
                     float* d_rgb, bool coherent)
{
	if (npts <= 0) return RT_HIP_OK;
	GridDesc gd;
	for (int k = 0; k < 3; ++k) { gd.mn[k] = ctx->photon_grid_min[k]; gd.dim[k] = ctx->num_photons ? ctx->photon_grid_dim[k] : 1; }
	gd.cell = ctx->photon_cell; gd.inv_cell = 1.0f / ctx->photon_cell;
	// rt_photon_map.cpp:13-18, evaluated in float exactly as the reference's file-scope constants
	const float ALPHA = 0.918, BETA = 1.953, PI = 3.141592653589793, E = 2.718281828459045;
	GatherConst gc;
	gc.alpha = ALPHA; gc.beta = BETA; gc.e = E;
	gc.gauss_denom = 1 / (1 - powf(E, -BETA));
	gc.two_pi = 2 * PI;
	// points with more than `heavy` photons in their cube go to the wave-per-point kernel (queue and count stay
	// on the device).  tuning.gather_heavy is the threshold (0: every point with a photon nearby takes
	// the wave kernel; a huge value: none does) -- the parity tests run both extremes.
	const uint32_t heavy = ctx->tuning.gather_heavy;
	// wave kernel, pass 1: at most this many photons in the coarse cube -> all of them are read from the coarse grid;
	// 0 = never, every queued point walks the fine grid's boundary shell (tests force either path)
	const uint32_t medium = ctx->tuning.gather_medium;
	// counts (16 ints) | heavy queue (npts, padded to even) | pass-1 results (npts + 1 float2) | sub-queue (npts)
	RT_CHECK(ctx->d_heavy.reserve(((size_t)2 * npts + 16 + 2) * sizeof(int) + ((size_t)npts + 1) * sizeof(float2)));
	int* heavy_count = ctx->d_heavy.as<int>();
	int* sub_count = heavy_count + 1;              // heavy-queue positions left to the general selection kernel
	int* heavy_queue = heavy_count + 16;
	float2* pass1 = (float2*)(heavy_queue + npts + (npts & 1));   // (count, max_dist2) per queue entry, between the wave kernels' launches
	int* sub_queue = (int*)(pass1 + npts + 1);
	RT_HIP_CHECK(hipMemsetAsync(heavy_count, 0, 2 * sizeof(int), ctx->stream));
	ctx->stats.gather_points += npts;
	if (unsigned long long* cen = ctx->num_photons > 0 ? rt_census_ptr(ctx, 2) : nullptr) {
		hipLaunchKernelGGL(k_photon_census, dim3(nblocks(npts, 256)), dim3(256), 0, ctx->stream, gd, ctx->d_cell_start.as<uint32_t>(),
		                   d_pos, npts, radius, cen);
		RT_HIP_CHECK(hipGetLastError());
	}
	RT_CHECK(rt_begin_trace_timing(ctx));
	// tuning.gather_sort: the queue of the wave kernels sorted by the points' positions (Morton order of their fine-grid cells), so
	// that neighbours in the queue are neighbours in space whatever order the caller's points come in: the pairs of the first pass
	// (tuning.gather_pair) form, and the XCD-aware walk of the queue reads each part of the map from one L2.  The results are
	// written by point index and do not depend on the queue's order.  1: unless the caller says its points are ordered already
	// (level 0 of a frame: 8 x 8 tiles); 2: always.
	// (1: not for a few thousand points or a sparse map either -- the sort is a dozen launches and a host wait for the queue's length,
	// 0.15 ms that a point's gather has to be worth: with 0.5 M photons stored the C5 frame measured 5.00 ms sorted against 4.84)
	const bool sort_queue = ctx->num_photons > 0 && ctx->tuning.gather_fine &&
	                        (ctx->tuning.gather_sort >= 2 || (ctx->tuning.gather_sort == 1 && !coherent && npts >= 16384 && ctx->num_photons >= (1ll << 21)));
	GridDesc gfk = gd;
	uint64_t* heavy_keys = nullptr;
	if (sort_queue) {
		for (int k = 0; k < 3; ++k) { gfk.mn[k] = ctx->photon_grid_min_f[k]; gfk.dim[k] = ctx->photon_grid_dim_f[k]; }
		gfk.cell = ctx->photon_cell_f; gfk.inv_cell = 1.0f / ctx->photon_cell_f;
		RT_CHECK(ctx->d_morton.reserve((size_t)npts * 8));
		RT_CHECK(ctx->d_morton_tmp.reserve((size_t)npts * 8));
		RT_CHECK(ctx->d_order_tmp.reserve((size_t)npts * 4));
		heavy_keys = ctx->d_morton.as<uint64_t>();
	}
	hipLaunchKernelGGL(k_photon_gather, dim3(nblocks(npts, GBLOCK)), dim3(GBLOCK), 0, ctx->stream, gd,
	                   ctx->d_cell_start.as<uint32_t>(), ctx->d_photons.as<float4>(), ctx->d_photon_sorted.as<float4>(),
	                   ctx->num_photons, d_pos, d_nrm, npts, count, radius, gc, d_rgb, heavy, heavy_queue, heavy_count, gfk, heavy_keys);
	RT_HIP_CHECK(hipGetLastError());
	if (sort_queue) {
		int nheavy = 0;
		RT_HIP_CHECK(hipMemcpyAsync(&nheavy, heavy_count, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
		RT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
		int maxdim = std::max(gfk.dim[0], std::max(gfk.dim[1], gfk.dim[2])), bits = 1;
		while ((1 << bits) < maxdim) ++bits;
		RT_CHECK(rt_radix_sort_pairs64(ctx, heavy_keys, ctx->d_morton_tmp.as<uint64_t>(), (uint32_t*)heavy_queue, ctx->d_order_tmp.as<uint32_t>(),
		                               nheavy, 3 * bits));
	}
	if (ctx->num_photons > 0) {
		// one wave per queued point, grid-stride over a queue whose length only the device knows
		// a multiple of 8 * chunk blocks, so that the XCD-aware renumbering is a bijection
		const int chunk = ctx->tuning.gather_xcd_chunk;
		const int unit = 8 * std::max(1, chunk);
		const int waves = (std::min(npts, 64 * 1024) + unit - 1) / unit * unit;
		if (ctx->tuning.gather_fine) {
			GridDesc gf;
			for (int k = 0; k < 3; ++k) { gf.mn[k] = ctx->photon_grid_min_f[k]; gf.dim[k] = ctx->photon_grid_dim_f[k]; }
			gf.cell = ctx->photon_cell_f; gf.inv_cell = 1.0f / ctx->photon_cell_f;
			const int chunk1 = std::max(1, chunk / RT_GATHER_P1_WAVES), unit1 = 8 * chunk1;
			const int blocks1 = ((std::min(npts, 64 * 1024) + RT_GATHER_P1_WAVES - 1) / RT_GATHER_P1_WAVES + unit1 - 1) / unit1 * unit1;
			hipLaunchKernelGGL((k_photon_gather_wave<true, 1>), dim3(blocks1), dim3(64 * RT_GATHER_P1_WAVES), 0, ctx->stream, gf, ctx->d_cell_start_f.as<uint32_t>(),
			                   ctx->d_photons_f.as<float4>(), ctx->d_photon_sorted_f.as<float4>(), ctx->num_photons, d_pos, d_nrm,
			                   heavy_queue, heavy_count, count, radius, gc, d_rgb, chunk > 0 ? chunk1 : 0, pass1, (const int*)nullptr, (const int*)nullptr, gd, ctx->d_cell_start.as<uint32_t>(), ctx->d_photons.as<float4>(), medium,
			                   ctx->photon_cell_f * (float)ctx->tuning.gather_pair * 0.01f);
			// the usual heavy point in the lean kernel; what it declines, in the general one
			hipLaunchKernelGGL(k_photon_select_common, dim3(waves), dim3(64), 0, ctx->stream, gf, ctx->d_cell_start_f.as<uint32_t>(),
			                   ctx->d_photons_f.as<float4>(), ctx->d_photon_sorted_f.as<float4>(), d_pos, d_nrm, heavy_queue, heavy_count, count,
			                   radius, gc, d_rgb, chunk, pass1, sub_queue, sub_count);
			hipLaunchKernelGGL((k_photon_gather_wave<true, 2>), dim3(waves), dim3(64), 0, ctx->stream, gf, ctx->d_cell_start_f.as<uint32_t>(),
			                   ctx->d_photons_f.as<float4>(), ctx->d_photon_sorted_f.as<float4>(), ctx->num_photons, d_pos, d_nrm,
			                   heavy_queue, heavy_count, count, radius, gc, d_rgb, chunk, pass1, (const int*)sub_queue, (const int*)sub_count, gd, ctx->d_cell_start.as<uint32_t>(), ctx->d_photons.as<float4>(), medium, 0.0f);
		} else {
			hipLaunchKernelGGL((k_photon_gather_wave<false, 0>), dim3(waves), dim3(64), 0, ctx->stream, gd, ctx->d_cell_start.as<uint32_t>(),
			                   ctx->d_photons.as<float4>(), ctx->d_photon_sorted.as<float4>(), ctx->num_photons, d_pos, d_nrm,
			                   heavy_queue, heavy_count, count, radius, gc, d_rgb, chunk, pass1, (const int*)nullptr, (const int*)nullptr, gd, ctx->d_cell_start.as<uint32_t>(), ctx->d_photons.as<float4>(), medium, 0.0f);
		}
		RT_HIP_CHECK(hipGetLastError());
	}
	RT_CHECK(rt_end_trace_timing(ctx, 2));
	return RT_HIP_OK;
}

int rt_photon_lookup(rt_hip_ctx* ctx, const float* d_pos, int npts, float radius, float* d_rgb)
{
	if (npts <= 0) return RT_HIP_OK;
	if (ctx->num_photons == 0) {   // an empty map sums to zero (rt_photon_map.cpp:130)
		RT_HIP_CHECK(hipMemsetAsync(d_rgb, 0, (size_t)npts * 12, ctx->stream));
		return RT_HIP_OK;
	}
	GridDesc gd;
	for (int k = 0; k < 3; ++k) { gd.mn[k] = ctx->photon_grid_min[k]; gd.dim[k] = ctx->photon_grid_dim[k]; }
	gd.cell = ctx->photon_cell; gd.inv_cell = 1.0f / ctx->photon_cell;
	hipLaunchKernelGGL(k_photon_lookup, dim3(std::min(npts, 64 * 1024)), dim3(64), 0, ctx->stream, gd, ctx->d_cell_start.as<uint32_t>(),
	                   ctx->d_photons.as<float4>(), ctx->d_photon_sorted.as<float4>(), d_pos, npts, radius, d_rgb);
	RT_HIP_CHECK(hipGetLastError());
	return RT_HIP_OK;
}

int rt_photon_unpermute(rt_hip_ctx* ctx, float* d_out)
{
	if (ctx->num_photons == 0) return RT_HIP_OK;
	hipLaunchKernelGGL(k_photon_unpermute, dim3(nblocks(ctx->num_photons, 256)), dim3(256), 0, ctx->stream, ctx->d_photons.as<float4>(),
	                   ctx->d_photon_sorted.as<float4>(), ctx->num_photons, (PhotonRec*)d_out);
	RT_HIP_CHECK(hipGetLastError());
	return RT_HIP_OK;
}

