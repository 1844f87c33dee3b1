// raytracer_amd/csrc/k_lbvh.hip -- GPU LBVH build.  Replaces the reference's per-frame
// divide-and-conquer partitioning (rt_raytracer.cpp:250-313) with a persistent acceleration
// structure built entirely on the device:
//   1. 63-bit Morton codes of the triangle-box centres, normalised to the scene bounds
//   2. LSD radix sort (k_sort.hip: wave64 ballots + LDS cursors)
//   3. Karras 2012 hierarchy, one thread per internal node
//   4. bottom-up refit with one arrival counter per node (the second arriver continues); every
//      subtree box is written to a 64-byte slot of its own so no cache line has two writers, and
//      the hand-off is an agent-scope release/acquire pair (the XCD L2s are not coherent)
//   5. 64-byte nodes carrying both children's boxes, leaf records permuted to Morton order
// "SAH-free": the topology is purely the Morton split; only boxes are refitted.
#include "rt_ctx.h"

namespace {

__device__ __forceinline__ uint64_t expand21(uint32_t v)
{
	uint64_t x = v & 0x1fffffull;
	x = (x | x << 32) & 0x1f00000000ffffull;
	x = (x | x << 16) & 0x1f0000ff0000ffull;
	x = (x | x << 8) & 0x100f00f00f00f00full;
	x = (x | x << 4) & 0x10c30c30c30c30c3ull;
	x = (x | x << 2) & 0x1249249249249249ull;
	return x;
}

__global__ __launch_bounds__(256) void k_morton(const float4* __restrict__ tri_box, int n, float bx, float by,
                                                float bz, float sx, float sy, float sz,
                                                uint64_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
	int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	float4 mn = tri_box[2 * (size_t)i], mx = tri_box[2 * (size_t)i + 1];
	float cx = ((mn.x + mx.x) * 0.5f - bx) * sx;
	float cy = ((mn.y + mx.y) * 0.5f - by) * sy;
	float cz = ((mn.z + mx.z) * 0.5f - bz) * sz;
	uint32_t ix = (uint32_t)fminf(fmaxf(cx * 2097152.0f, 0.0f), 2097151.0f);
	uint32_t iy = (uint32_t)fminf(fmaxf(cy * 2097152.0f, 0.0f), 2097151.0f);
	uint32_t iz = (uint32_t)fminf(fmaxf(cz * 2097152.0f, 0.0f), 2097151.0f);
	keys[i] = (expand21(ix) << 2) | (expand21(iy) << 1) | expand21(iz);
	vals[i] = (uint32_t)i;
}

// Common-prefix length of sorted keys i and j; equal keys fall back to the index (Karras 2012, 4).
__device__ __forceinline__ int delta(const uint64_t* __restrict__ keys, int n, int i, int j)
{
	if (j < 0 || j >= n) return -1;
	uint64_t a = keys[i], b = keys[j];
	if (a == b) return 64 + __clz((unsigned)(i ^ j));
	return __clzll((long long)(a ^ b));
}

// child encoding: >= 0 internal node index, < 0 ~leaf slot
__global__ __launch_bounds__(256) void k_hierarchy(const uint64_t* __restrict__ keys, int n, int* __restrict__ left,
                                                   int* __restrict__ right, int* __restrict__ node_parent,
                                                   int* __restrict__ leaf_parent)
{
	int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n - 1) return;
	int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
	int dmin = delta(keys, n, i, i - d);
	int lmax = 2;
	while (delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
	int l = 0;
	for (int t = lmax >> 1; t >= 1; t >>= 1)
		if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
	int j = i + l * d;
	int dnode = delta(keys, n, i, j);
	int s = 0;
	int t = l;
	do {
		t = (t + 1) >> 1;
		if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
	} while (t > 1);
	int gamma = i + s * d + min(d, 0);
	int lo = min(i, j), hi = max(i, j);
	int lc = (lo == gamma) ? ~gamma : gamma;
	int rc = (hi == gamma + 1) ? ~(gamma + 1) : gamma + 1;
	left[i] = lc;
	right[i] = rc;
	if (lc >= 0) node_parent[lc] = i; else leaf_parent[~lc] = i;
	if (rc >= 0) node_parent[rc] = i; else leaf_parent[~rc] = i;
	if (i == 0) node_parent[0] = -1;
}

struct __attribute__((aligned(64))) BoxSlot {
	float mn[3], mx[3];
	float pad[10];
};
static_assert(sizeof(BoxSlot) == 64, "one cache line per subtree box");

// Slots [0, n-1) = internal nodes, [n-1, 2n-1) = leaves.
__global__ __launch_bounds__(256) void k_refit(const float4* __restrict__ tri_box, const uint32_t* __restrict__ order,
                                               int n, float pad, const int* __restrict__ left,
                                               const int* __restrict__ right, const int* __restrict__ node_parent,
                                               const int* __restrict__ leaf_parent, BoxSlot* slots,
                                               unsigned int* arrive)
{
	int j = blockIdx.x * 256 + threadIdx.x;
	if (j >= n) return;
	uint32_t prim = order[j];
	float4 a = tri_box[2 * (size_t)prim], b = tri_box[2 * (size_t)prim + 1];
	float mn[3] = {a.x - pad, a.y - pad, a.z - pad}, mx[3] = {b.x + pad, b.y + pad, b.z + pad};
	BoxSlot* me = slots + (n - 1) + j;
	for (int k = 0; k < 3; ++k) { me->mn[k] = mn[k]; me->mx[k] = mx[k]; }
	int p = leaf_parent[j];
	int self = ~j;
	while (p >= 0) {
		__threadfence();  // release my subtree box (agent scope: write back this XCD's L2)
		unsigned int prev = atomicAdd(&arrive[p], 1u);
		if (prev == 0u) return;  // the sibling subtree is not finished; its thread will continue
		__threadfence();  // acquire: drop stale lines before reading the sibling's box
		int sib = (left[p] == self) ? right[p] : left[p];
		const BoxSlot* sb = (sib >= 0) ? (slots + sib) : (slots + (n - 1) + (~sib));
		for (int k = 0; k < 3; ++k) {
			float smn = __builtin_nontemporal_load(&sb->mn[k]);
			float smx = __builtin_nontemporal_load(&sb->mx[k]);
			mn[k] = fminf(mn[k], smn);
			mx[k] = fmaxf(mx[k], smx);
		}
		BoxSlot* ps = slots + p;
		for (int k = 0; k < 3; ++k) { ps->mn[k] = mn[k]; ps->mx[k] = mx[k]; }
		self = p;
		p = node_parent[p];
	}
}

__global__ __launch_bounds__(256) void k_emit_nodes(const BoxSlot* __restrict__ slots, int n,
                                                    const int* __restrict__ left, const int* __restrict__ right,
                                                    const int* __restrict__ node_parent, BvhNode* __restrict__ nodes)
{
	int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n - 1) return;
	int lc = left[i], rc = right[i];
	const BoxSlot* l = (lc >= 0) ? (slots + lc) : (slots + (n - 1) + (~lc));
	const BoxSlot* r = (rc >= 0) ? (slots + rc) : (slots + (n - 1) + (~rc));
	BvhNode nd;
	nd.q0 = make_float4(l->mn[0], l->mn[1], l->mn[2], l->mx[0]);
	nd.q1 = make_float4(l->mx[1], l->mx[2], r->mn[0], r->mn[1]);
	nd.q2 = make_float4(r->mn[2], r->mx[0], r->mx[1], r->mx[2]);
	nd.left = lc; nd.right = rc; nd.parent = node_parent[i]; nd.pad = 0;
	nodes[i] = nd;
}

__global__ __launch_bounds__(256) void k_permute_tris(const TriIsect* __restrict__ in, const uint32_t* __restrict__ order,
                                                      int n, TriIsect* __restrict__ out)
{
	int j = blockIdx.x * 256 + threadIdx.x;
	if (j < n) out[j] = in[order[j]];
}

__global__ __launch_bounds__(256) void k_depth(const int* __restrict__ node_parent, const int* __restrict__ leaf_parent,
                                               int n, int* __restrict__ max_depth)
{
	int j = blockIdx.x * 256 + threadIdx.x;
	int d = 0;
	if (j < n) {
		int p = leaf_parent[j];
		while (p >= 0) { ++d; p = node_parent[p]; }
	}
	// wave max, then one atomic per wave
	for (int off = 32; off >= 1; off >>= 1) d = max(d, __shfl_xor(d, off));
	if ((threadIdx.x & 63) == 0) atomicMax(max_depth, d);
}

}  // namespace

int rt_build_lbvh(rt_hip_ctx* ctx)
{
	hipStream_t st = ctx->stream;
	const int n = ctx->num_tris;
	const int g = (n + 255) / 256;
	RT_CHECK(ctx->d_morton.reserve((size_t)n * 8));
	RT_CHECK(ctx->d_morton_tmp.reserve((size_t)n * 8));
	RT_CHECK(ctx->d_order.reserve((size_t)n * 4));
	RT_CHECK(ctx->d_order_tmp.reserve((size_t)n * 4));
	uint64_t* keys = ctx->d_morton.as<uint64_t>();
	uint32_t* order = ctx->d_order.as<uint32_t>();

	float ext[3];
	for (int k = 0; k < 3; ++k) ext[k] = ctx->bounds[3 + k] - ctx->bounds[k];
	float emax = std::max(std::max(ext[0], ext[1]), ext[2]);
	if (!(emax > 0.0f)) emax = 1.0f;
	// one cubic grid over the longest extent keeps cells isotropic
	float s = 1.0f / emax;
	hipLaunchKernelGGL(k_morton, dim3(g), dim3(256), 0, st, ctx->d_tri_box.as<float4>(), n, ctx->bounds[0],
	                   ctx->bounds[1], ctx->bounds[2], s, s, s, keys, order);
	RT_HIP_CHECK(hipGetLastError());
	RT_CHECK(rt_radix_sort_pairs64(ctx, keys, ctx->d_morton_tmp.as<uint64_t>(), order,
	                               ctx->d_order_tmp.as<uint32_t>(), n, 63));

	// topology arrays: left, right, node_parent (n-1 each), leaf_parent (n), arrive (n-1), depth (1)
	size_t ints = (size_t)4 * n + 8;
	RT_CHECK(ctx->d_parent_leaf.reserve(ints * 4));
	int* left = ctx->d_parent_leaf.as<int>();
	int* right = left + n;
	int* node_parent = right + n;
	int* leaf_parent = node_parent + n;
	RT_CHECK(ctx->d_flags.reserve((size_t)(n + 16) * 4));
	unsigned int* arrive = ctx->d_flags.as<unsigned int>();
	int* max_depth = (int*)(arrive + n);
	RT_HIP_CHECK(hipMemsetAsync(arrive, 0, (size_t)(n + 16) * 4, st));
	RT_CHECK(ctx->d_misc.reserve((size_t)(2 * n) * sizeof(BoxSlot)));
	BoxSlot* slots = ctx->d_misc.as<BoxSlot>();

	hipLaunchKernelGGL(k_hierarchy, dim3(g), dim3(256), 0, st, keys, n, left, right, node_parent, leaf_parent);
	float pad = 1e-5f * emax;
	hipLaunchKernelGGL(k_refit, dim3(g), dim3(256), 0, st, ctx->d_tri_box.as<float4>(), order, n, pad, left, right,
	                   node_parent, leaf_parent, slots, arrive);
	RT_CHECK(ctx->d_nodes.reserve((size_t)(n - 1) * sizeof(BvhNode)));
	hipLaunchKernelGGL(k_emit_nodes, dim3(g), dim3(256), 0, st, slots, n, left, right, node_parent,
	                   ctx->d_nodes.as<BvhNode>());
	// leaf-ordered intersection records (tri_isect currently holds original order)
	RT_CHECK(ctx->d_in_a.reserve((size_t)n * sizeof(TriIsect)));
	RT_HIP_CHECK(hipMemcpyAsync(ctx->d_in_a.p, ctx->d_tri_isect.p, (size_t)n * sizeof(TriIsect), hipMemcpyDeviceToDevice, st));
	hipLaunchKernelGGL(k_permute_tris, dim3(g), dim3(256), 0, st, ctx->d_in_a.as<TriIsect>(), order, n,
	                   ctx->d_tri_isect.as<TriIsect>());
	hipLaunchKernelGGL(k_depth, dim3(g), dim3(256), 0, st, node_parent, leaf_parent, n, max_depth);
	RT_HIP_CHECK(hipGetLastError());
	int depth = 0;
	RT_HIP_CHECK(hipMemcpyAsync(&depth, max_depth, 4, hipMemcpyDeviceToHost, st));
	RT_HIP_CHECK(hipStreamSynchronize(st));
	ctx->bvh_depth = depth;
	ctx->num_nodes = n - 1;
	if (depth + 1 > 96) {
		// pathological (massively duplicated centres): fall back to brute force rather than overflow
		rt_set_error("LBVH depth %d exceeds the traversal stack limit (96)", depth);
		return RT_HIP_ERR_LIMIT;
	}
	// overflow part of the traversal stacks: entries beyond the LDS-resident ones, one column per
	// thread of a persistent trace launch
	int over = depth + 1 - rt_lds_depth();
	if (over > 0) RT_CHECK(ctx->d_spill.reserve((size_t)over * ctx->max_trace_threads * sizeof(int)));
	ctx->use_bvh = 1;
	return RT_HIP_OK;
}
