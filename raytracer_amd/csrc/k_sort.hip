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

// Exclusive scan of `len` counters in place (digit-major order), one workgroup of 1024 threads.
__global__ __launch_bounds__(1024) void k_ms_scan(uint32_t* __restrict__ data, int len, int nb,
                                                  uint32_t* __restrict__ bucket_totals)
{
	__shared__ uint32_t part[1024];
	int tid = threadIdx.x;
	int chunk = (len + 1023) / 1024;
	int lo = tid * chunk, hi = min(lo + chunk, len);
	uint32_t s = 0;
	for (int k = lo; k < hi; ++k) s += data[k];
	part[tid] = s;
	__syncthreads();
	// Hillis-Steele inclusive scan over 1024 partials
	for (int off = 1; off < 1024; off <<= 1) {
		uint32_t v = (tid >= off) ? part[tid - off] : 0;
		__syncthreads();
		part[tid] += v;
		__syncthreads();
	}
	uint32_t run = (tid == 0) ? 0 : part[tid - 1];
	for (int k = lo; k < hi; ++k) {
		uint32_t v = data[k];
		data[k] = run;
		run += v;
	}
	__syncthreads();
	if (bucket_totals) {
		// total of bucket d = start(d+1) - start(d); data is now exclusive starts, digit-major
		for (int d = tid; d < 256; d += 1024) {
			uint32_t start = data[(size_t)d * nb];
			uint32_t next = (d == 255) ? part[1023] : data[(size_t)(d + 1) * nb];
			bucket_totals[d] = next - start;
		}
	}
}

__global__ __launch_bounds__(MS_WAVE) void k_ms_rank(const uint32_t* __restrict__ digit, int n, int nb,
                                                     const uint32_t* __restrict__ gstart, uint32_t* __restrict__ dest)
{
	__shared__ uint32_t cur[256];
	for (int k = threadIdx.x; k < 256; k += MS_WAVE) cur[k] = gstart[(size_t)k * nb + blockIdx.x];
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
	RT_CHECK(ctx->d_hist.reserve((size_t)256 * nb * sizeof(uint32_t)));
	uint32_t* gh = ctx->d_hist.as<uint32_t>();
	hipLaunchKernelGGL(k_ms_hist, dim3(nb), dim3(MS_WAVE), 0, ctx->stream, d_digit, n, nb, gh);
	hipLaunchKernelGGL(k_ms_scan, dim3(1), dim3(1024), 0, ctx->stream, gh, 256 * nb, nb, d_bucket_totals);
	hipLaunchKernelGGL(k_ms_rank, dim3(nb), dim3(MS_WAVE), 0, ctx->stream, d_digit, n, nb, gh, d_dest);
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
