// raytracer_amd/csrc/k_noalloc.hip -- rt_hip_gbuffer_select_noalloc_device: the G-buffer selection
// (Rasterizer::getRayTraceData, ras_rasterizer.cpp:509-536,556-561) for callers that cannot
// allocate device memory (code under HIP graph capture, tight serving loops, harnesses that forbid
// hipMalloc).  rt_hip_gbuffer_select_device keeps a digit and a destination word per texel plus the
// multisplit's histograms in context buffers that grow on demand; this variant needs one count per
// 2048-texel tile and keeps those in a file-scope device array, so nothing is ever allocated.
//
// The array exists once per GPU (every device that loads the code object holds its own copy at its
// own address) and keeps the previous call's contents: every word a call reads it has written
// first (tile counts [0, nb), the total behind them), so nothing has to start at zero.  Calls that
// overlap on one GPU -- two contexts, two threads -- race on it.
#include <mutex>

#include "rt_ctx.h"

namespace {

constexpr int NA_WAVE = 64;
constexpr int NA_ROUNDS = 32;
constexpr int NA_TILE = NA_WAVE * NA_ROUNDS;                      // 2048 texels per workgroup
constexpr int NA_MAX_TILES = RT_HIP_NOALLOC_MAX_TEXELS / NA_TILE; // 8192
constexpr int NA_SCAN_BLOCK = 1024;
constexpr int NA_WORDS = NA_MAX_TILES + 64;                       // tile counts, then the total

// 33 KB per GPU, held until the process ends.  In host code this name is only a handle for
// hipGetSymbolAddress: the address kernels and copies use is g_scratch_ptr[device].
static __device__ __align__(256) uint32_t scratch[NA_WORDS];

constexpr int NA_MAX_DEVICES = 64;
uint32_t* g_scratch_ptr[NA_MAX_DEVICES] = {};

__device__ __forceinline__ bool na_keep(const float* __restrict__ nrm_rgba, int i, int n, int interlace, int phase)
{
	if (i >= n) return false;
	if (interlace > 1 && (i % interlace) != phase) return false;
	return nrm_rgba[4 * (size_t)i + 3] > 0.0f;
}

// one wave per tile: how many texels of the tile are kept
__global__ __launch_bounds__(NA_WAVE) void k_na_count(const float* __restrict__ nrm_rgba, int n, int interlace, int phase,
                                                      uint32_t* __restrict__ tile_count)
{
	const int base = blockIdx.x * NA_TILE;
	uint32_t c = 0;
	for (int r = 0; r < NA_ROUNDS; ++r)
		c += (uint32_t)__popcll(__ballot(na_keep(nrm_rgba, base + r * NA_WAVE + threadIdx.x, n, interlace, phase)));
	if (threadIdx.x == 0) tile_count[blockIdx.x] = c;
}

// one workgroup: exclusive scan of the nb tile counts in place, total to tile_count[NA_MAX_TILES]
__global__ __launch_bounds__(NA_SCAN_BLOCK) void k_na_scan(uint32_t* __restrict__ tile_count, int nb, int* __restrict__ count_out)
{
	__shared__ uint32_t wsum[NA_SCAN_BLOCK / 64];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t carry = 0;
	for (int base = 0; base < nb; base += NA_SCAN_BLOCK) {
		const int k = base + threadIdx.x;
		const uint32_t v = k < nb ? tile_count[k] : 0u;
		uint32_t x = v;
		for (int off = 1; off < 64; off <<= 1) {
			uint32_t y = __shfl_up(x, off);
			if (lane >= off) x += y;
		}
		if (lane == 63) wsum[wave] = x;
		__syncthreads();
		uint32_t wbase = 0, tot = 0;
		for (int w = 0; w < NA_SCAN_BLOCK / 64; ++w) { if (w < wave) wbase += wsum[w]; tot += wsum[w]; }
		if (k < nb) tile_count[k] = carry + wbase + x - v;
		carry += tot;
		__syncthreads();
	}
	if (threadIdx.x == 0) { tile_count[NA_MAX_TILES] = carry; if (count_out) *count_out = (int)carry; }
}

// one wave per tile: kept texels go to tile start + rank inside the tile (ascending texel index: stable)
__global__ __launch_bounds__(NA_WAVE) void k_na_scatter(const float* __restrict__ nrm_rgba, const float* __restrict__ pos_rgb,
                                                        int n, int interlace, int phase, const uint32_t* __restrict__ tile_start,
                                                        float* __restrict__ pos_out, float* __restrict__ nrm_out,
                                                        int* __restrict__ index_out)
{
	const int base = blockIdx.x * NA_TILE;
	const uint64_t lower = (1ull << threadIdx.x) - 1ull;
	uint32_t run = tile_start[blockIdx.x];
	for (int r = 0; r < NA_ROUNDS; ++r) {
		const int i = base + r * NA_WAVE + threadIdx.x;
		const bool keep = na_keep(nrm_rgba, i, n, interlace, phase);
		const uint64_t m = __ballot(keep);
		if (keep) {
			const size_t d = run + (uint32_t)__popcll(m & lower);
			for (int k = 0; k < 3; ++k) {
				// (*normals)[i][k] = buffer * 2.0 - 1.0 in double, stored to float (ras_rasterizer.cpp:533-535)
				nrm_out[3 * d + k] = (float)((double)nrm_rgba[4 * (size_t)i + k] * 2.0 - 1.0);
				pos_out[3 * d + k] = pos_rgb[3 * (size_t)i + k];
			}
			if (index_out) index_out[d] = i;
		}
		run += (uint32_t)__popcll(m);
	}
}

}  // namespace

// Set-up, once per device, with that device current (rt_hip_create calls it): the device address of
// this GPU's copy of `scratch`.
int rt_noalloc_setup(int device)
{
	if (device < 0 || device >= NA_MAX_DEVICES) { rt_set_error("device %d beyond the %d the no-alloc entry keeps scratch for", device, NA_MAX_DEVICES); return RT_HIP_ERR_LIMIT; }
	static std::mutex mu;  // contexts may be created from several threads at once
	std::lock_guard<std::mutex> lock(mu);
	if (g_scratch_ptr[device]) return RT_HIP_OK;
	uint32_t* p = nullptr;
	RT_HIP_CHECK(hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch)));
	if (!p) { rt_set_error("hipGetSymbolAddress returned a null address for the no-alloc scratch"); return RT_HIP_ERR_HIP; }
	g_scratch_ptr[device] = p;
	return RT_HIP_OK;
}

// out_count == nullptr: nothing is copied to the host and nothing is waited for -- the three kernels are the whole call,
// which is what makes it legal on a capturing stream; the count is then left in *d_count_out (device memory of the caller).
int rt_gbuffer_select_noalloc(rt_hip_ctx* ctx, const float* d_nrm_rgba, const float* d_pos_rgb, int n, int interlace,
                              int phase, float* d_pos_out, float* d_nrm_out, int* d_index_out, int* d_count_out, int* out_count)
{
	if (out_count) *out_count = 0;
	if (n > RT_HIP_NOALLOC_MAX_TEXELS) {
		rt_set_error("rt_hip_gbuffer_select_noalloc_device: %d texels exceed the fixed scratch (at most %d)", n, RT_HIP_NOALLOC_MAX_TEXELS);
		return RT_HIP_ERR_LIMIT;
	}
	const int device = ctx->cfg.device;
	uint32_t* sp = (device >= 0 && device < NA_MAX_DEVICES) ? g_scratch_ptr[device] : nullptr;
	if (!sp) { rt_set_error("no-alloc scratch of device %d was not set up", device); return RT_HIP_ERR_STATE; }
	hipStream_t st = ctx->stream;
	const int nb = (n + NA_TILE - 1) / NA_TILE;   // n == 0: the scan alone, which leaves a count of 0
	if (nb > 0) hipLaunchKernelGGL(k_na_count, dim3(nb), dim3(NA_WAVE), 0, st, d_nrm_rgba, n, interlace, phase, sp);
	hipLaunchKernelGGL(k_na_scan, dim3(1), dim3(NA_SCAN_BLOCK), 0, st, sp, nb, d_count_out);
	if (nb > 0) hipLaunchKernelGGL(k_na_scatter, dim3(nb), dim3(NA_WAVE), 0, st, d_nrm_rgba, d_pos_rgb, n, interlace, phase, sp, d_pos_out,
	                               d_nrm_out, d_index_out);
	RT_HIP_CHECK(hipGetLastError());
	if (!out_count) return RT_HIP_OK;
	// the count comes back through the context's pinned word (allocated by rt_hip_create)
	uint32_t* h_cnt = (uint32_t*)ctx->h_small;
	RT_HIP_CHECK(hipMemcpyAsync(h_cnt, sp + NA_MAX_TILES, 4, hipMemcpyDeviceToHost, st));
	RT_HIP_CHECK(hipStreamSynchronize(st));
	*out_count = (int)*h_cnt;
	return RT_HIP_OK;
}
