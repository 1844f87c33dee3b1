// raytracer_amd/csrc/k_sort.hip -- stable 256-way multisplit (one counting-sort pass) and the
// LSD radix sort built from it.  Hand-written for wave64: one wave per workgroup, digit
// histograms and running bucket cursors in LDS, in-wave ranks from 8 ballots (match-any on the
// 8-bit digit) + popcount of the lower-lane mask.  Used by the LBVH build (63-bit Morton keys)
// and by the group-by-material step of processRays (rt_raytracer.cpp:527-534).
#include "rt_ctx.h"

namespace {

constexpr int MS_WAVE = 64;
constexpr int MS_ROUNDS = 32;
constexpr int MS_TILE = MS_WAVE * MS_ROUNDS;  // 2048 elements per workgroup

__global__ __launch_bounds__(MS_WAVE) void k_ms_hist(const uint32_t* __restrict__ digit, int n, int nb,
                                                     uint32_t* __restrict__ ghist)
{
	__shared__ uint32_t h[256];
	for (int k = threadIdx.x; k < 256; k += MS_WAVE) h[k] = 0;
	__syncthreads();
	int base = blockIdx.x * MS_TILE;
	for (int r = 0; r < MS_ROUNDS; ++r) {
		int i = base + r * MS_WAVE + threadIdx.x;
		if (i < n) atomicAdd(&h[digit[i] & 255u], 1u);
	}
	__syncthreads();
	for (int k = threadIdx.x; k < 256; k += MS_WAVE) ghist[(size_t)k * nb + blockIdx.x] = h[k];
}

// Workgroup d scans the nb per-tile counters of digit d in place (exclusive, coalesced: the counters
// are digit-major) and writes the digit's total.  256 workgroups instead of one: a single workgroup
// walking all 256 x nb counters took 250 us for a 2 M-element multisplit.
__global__ __launch_bounds__(256) void k_ms_scan(uint32_t* __restrict__ data, int nb, uint32_t* __restrict__ totals)
{
	__shared__ uint32_t wsum[4];
	uint32_t* row = data + (size_t)blockIdx.x * nb;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t carry = 0;
	for (int base = 0; base < nb; base += 256) {
		const int k = base + threadIdx.x;
		const uint32_t v = k < nb ? row[k] : 0u;
		uint32_t x = v;
		for (int off = 1; off < 64; off <<= 1) {
			uint32_t y = __shfl_up(x, off);
			if (lane >= off) x += y;
		}
		if (lane == 63) wsum[wave] = x;
		__syncthreads();
		uint32_t wbase = 0, tot = 0;
		for (int w = 0; w < 4; ++w) { if (w < wave) wbase += wsum[w]; tot += wsum[w]; }
		if (k < nb) row[k] = carry + wbase + x - v;
		carry += tot;
		__syncthreads();
	}
	if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

__global__ __launch_bounds__(MS_WAVE) void k_ms_rank(const uint32_t* __restrict__ digit, int n, int nb,
                                                     const uint32_t* __restrict__ gstart,
                                                     const uint32_t* __restrict__ totals, uint32_t* __restrict__ dest)
{
	__shared__ uint32_t cur[256];
	{
		// start of digit d = exclusive scan of the 256 digit totals (4 per lane + a wave scan)
		uint32_t t[4], s = 0;
		for (int k = 0; k < 4; ++k) { t[k] = totals[4 * threadIdx.x + k]; s += t[k]; }
		uint32_t x = s;
		for (int off = 1; off < 64; off <<= 1) {
			uint32_t y = __shfl_up(x, off);
			if ((int)threadIdx.x >= off) x += y;
		}
		uint32_t run = x - s;
		for (int k = 0; k < 4; ++k) {
			const int d = 4 * threadIdx.x + k;
			cur[d] = run + gstart[(size_t)d * nb + blockIdx.x];
			run += t[k];
		}
	}
	__syncthreads();
	int base = blockIdx.x * MS_TILE;
	const uint64_t lt = (1ull << threadIdx.x) - 1ull;
	for (int r = 0; r < MS_ROUNDS; ++r) {
		int i = base + r * MS_WAVE + threadIdx.x;
		bool active = i < n;
		uint32_t d = active ? (digit[i] & 255u) : 0u;
		uint64_t act = __ballot(active);
		uint64_t same = act;
#pragma unroll
		for (int b = 0; b < 8; ++b) {
			bool bit = (d >> b) & 1u;
			uint64_t m = __ballot(active && bit);
			same &= bit ? m : ~m;
		}
		uint32_t basec = active ? cur[d] : 0u;
		__syncthreads();  // every lane has read its cursor before any leader advances it
		if (active) {
			dest[i] = basec + (uint32_t)__popcll(same & lt);
			// the highest lane of each digit group advances the cursor by the group size
			if ((same >> threadIdx.x) == 1ull) cur[d] = basec + (uint32_t)__popcll(same);
		}
		__syncthreads();
	}
}

__global__ void k_extract_digit(const uint64_t* __restrict__ keys, int n, int shift, uint32_t* __restrict__ digit)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) digit[i] = (uint32_t)((keys[i] >> shift) & 255ull);
}

__global__ void k_scatter_pairs(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                const uint32_t* __restrict__ dest, int n, uint64_t* __restrict__ okeys,
                                uint32_t* __restrict__ ovals)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) {
		uint32_t d = dest[i];
		okeys[d] = keys[i];
		ovals[d] = vals[i];
	}
}

}  // namespace

int rt_multisplit256(rt_hip_ctx* ctx, const uint32_t* d_digit, int n, uint32_t* d_dest, uint32_t* d_bucket_totals)
{
	if (n <= 0) {
		if (d_bucket_totals) RT_HIP_CHECK(hipMemsetAsync(d_bucket_totals, 0, 256 * sizeof(uint32_t), ctx->stream));
		return RT_HIP_OK;
	}
	int nb = (n + MS_TILE - 1) / MS_TILE;
	RT_CHECK(ctx->d_hist.reserve(((size_t)256 * nb + 256) * sizeof(uint32_t)));
	uint32_t* gh = ctx->d_hist.as<uint32_t>();
	uint32_t* totals = d_bucket_totals ? d_bucket_totals : gh + (size_t)256 * nb;
	hipLaunchKernelGGL(k_ms_hist, dim3(nb), dim3(MS_WAVE), 0, ctx->stream, d_digit, n, nb, gh);
	hipLaunchKernelGGL(k_ms_scan, dim3(256), dim3(256), 0, ctx->stream, gh, nb, totals);
	hipLaunchKernelGGL(k_ms_rank, dim3(nb), dim3(MS_WAVE), 0, ctx->stream, d_digit, n, nb, gh, totals, d_dest);
	RT_HIP_CHECK(hipGetLastError());
	return RT_HIP_OK;
}

// LSD radix sort of (64-bit key, 32-bit value) pairs, 8 bits per pass.  On return the sorted
// data is in d_keys / d_vals (an even number of passes is always run).
int rt_radix_sort_pairs64(rt_hip_ctx* ctx, uint64_t* d_keys, uint64_t* d_keys_tmp, uint32_t* d_vals,
                          uint32_t* d_vals_tmp, int n, int key_bits)
{
	if (n <= 1) return RT_HIP_OK;
	int passes = (key_bits + 7) / 8;
	if (passes & 1) ++passes;
	RT_CHECK(ctx->d_keys.reserve((size_t)n * sizeof(uint32_t)));
	RT_CHECK(ctx->d_rank.reserve((size_t)n * sizeof(uint32_t)));
	uint32_t* digit = ctx->d_keys.as<uint32_t>();
	uint32_t* dest = ctx->d_rank.as<uint32_t>();
	uint64_t* ki = d_keys; uint64_t* ko = d_keys_tmp;
	uint32_t* vi = d_vals; uint32_t* vo = d_vals_tmp;
	int g = (n + 255) / 256;
	for (int p = 0; p < passes; ++p) {
		hipLaunchKernelGGL(k_extract_digit, dim3(g), dim3(256), 0, ctx->stream, ki, n, p * 8, digit);
		RT_CHECK(rt_multisplit256(ctx, digit, n, dest, nullptr));
		hipLaunchKernelGGL(k_scatter_pairs, dim3(g), dim3(256), 0, ctx->stream, ki, vi, dest, n, ko, vo);
		RT_HIP_CHECK(hipGetLastError());
		uint64_t* tk = ki; ki = ko; ko = tk;
		uint32_t* tv = vi; vi = vo; vo = tv;
	}
	return RT_HIP_OK;
}
