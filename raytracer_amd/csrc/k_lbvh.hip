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
//
// Bottom-up boxes, level by level.  A node's children sit one level deeper, so processing the levels from the
// deepest to the root -- one launch per level over that level's node list -- makes every node read finished child
// boxes, with the kernel boundary as the only synchronisation.  (The first version walked up from every leaf with an
// arrival counter per node and an agent-scope release / acquire pair per step: correct, but 1.5 ms for 262k triangles --
// each fence writes back the XCD's L2 -- against ~0.15 ms for all levels this way.)
__global__ __launch_bounds__(256) void k_leaf_slots(const float4* __restrict__ tri_box, const uint32_t* __restrict__ order,
                                                    int n, float pad, BoxSlot* __restrict__ slots)
{
	int j = blockIdx.x * 256 + threadIdx.x;
	if (j >= n) return;
	uint32_t prim = order[j];
	float4 a = tri_box[2 * (size_t)prim], b = tri_box[2 * (size_t)prim + 1];
	BoxSlot* me = slots + (n - 1) + j;
	me->mn[0] = a.x - pad; me->mn[1] = a.y - pad; me->mn[2] = a.z - pad;
	me->mx[0] = b.x + pad; me->mx[1] = b.y + pad; me->mx[2] = b.z + pad;
}

// depth of every internal node (root = 0) and the per-depth histogram
constexpr int MAX_LEVELS = 128;
__global__ __launch_bounds__(256) void k_node_depth(const int* __restrict__ node_parent, int n_internal, int* __restrict__ depth,
                                                    int* __restrict__ hist)
{
	__shared__ int h[MAX_LEVELS];
	if (threadIdx.x < MAX_LEVELS) h[threadIdx.x] = 0;
	__syncthreads();
	int i = blockIdx.x * 256 + threadIdx.x;
	if (i < n_internal) {
		int d = 0;
		for (int p = node_parent[i]; p >= 0; p = node_parent[p]) ++d;
		depth[i] = d;
		atomicAdd(&h[min(d, MAX_LEVELS - 1)], 1);
	}
	__syncthreads();
	if (threadIdx.x < MAX_LEVELS && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

__global__ __launch_bounds__(256) void k_level_scatter(const int* __restrict__ depth, int n_internal, int* __restrict__ cursor,
                                                       int* __restrict__ level_nodes)
{
	int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n_internal) return;
	// one returning atomic per (wave, depth value): lanes of equal depth take consecutive places
	const int d = depth[i];
	uint64_t todo = __ballot(1);
	while (todo) {
		const int lead = __ffsll((unsigned long long)todo) - 1;
		const int dl = __shfl(d, lead);
		const uint64_t same = __ballot(d == dl) & todo;
		int base = 0;
		if (lane_id() == lead) base = atomicAdd(&cursor[dl], (int)__popcll(same));
		base = __shfl(base, lead);
		if (d == dl) level_nodes[base + (int)__popcll(same & lanemask_lt())] = i;
		todo &= ~same;
	}
}


// ---- PLOC: parallel locally-ordered clustering (Meister & Bittner 2018) ---------------------------
// Bottom-up agglomeration over the Morton-ordered leaves: every cluster looks `r` places to either
// side for the neighbour whose union with it has the smallest surface area; mutual nearest
// neighbours merge; the survivors are compacted in order; repeat until one cluster is left.  The
// topology follows surface area instead of Morton bits, which is what cuts node visits per ray.
// Internal node ids are handed out backwards (the k-th merge gets id n-2-k), so the root is node 0
// and parents sit at lower addresses than their children.
struct __attribute__((aligned(16))) CBox {
	float4 lo, hi;  // xyz used
};

__device__ __forceinline__ float union_half_area(const CBox& a, const CBox& b)
{
	float dx = fmaxf(a.hi.x, b.hi.x) - fminf(a.lo.x, b.lo.x);
	float dy = fmaxf(a.hi.y, b.hi.y) - fminf(a.lo.y, b.lo.y);
	float dz = fmaxf(a.hi.z, b.hi.z) - fminf(a.lo.z, b.lo.z);
	return dx * dy + (dy * dz + dz * dx);
}
__device__ __forceinline__ CBox union_box(const CBox& a, const CBox& b)
{
	CBox u;
	u.lo = make_float4(fminf(a.lo.x, b.lo.x), fminf(a.lo.y, b.lo.y), fminf(a.lo.z, b.lo.z), 0.0f);
	u.hi = make_float4(fmaxf(a.hi.x, b.hi.x), fmaxf(a.hi.y, b.hi.y), fmaxf(a.hi.z, b.hi.z), 0.0f);
	return u;
}
__device__ __forceinline__ void store_slot(BoxSlot* s, const CBox& b)
{
	s->mn[0] = b.lo.x; s->mn[1] = b.lo.y; s->mn[2] = b.lo.z;
	s->mx[0] = b.hi.x; s->mx[1] = b.hi.y; s->mx[2] = b.hi.z;
}

constexpr int PB = 256;       // clusters per block in the grid-wide phases
constexpr int PR_MAX = 64;    // largest search radius
constexpr int PTAIL = 1024;   // cluster count at which one workgroup finishes the build in LDS

__global__ __launch_bounds__(256) void k_ploc_init(const float4* __restrict__ tri_box, const uint32_t* __restrict__ order,
                                                   int n, float pad, BoxSlot* __restrict__ slots, int* __restrict__ cl,
                                                   CBox* __restrict__ cbox)
{
	int j = blockIdx.x * 256 + threadIdx.x;
	if (j >= n) return;
	uint32_t prim = order[j];
	float4 a = tri_box[2 * (size_t)prim], b = tri_box[2 * (size_t)prim + 1];
	CBox c;
	c.lo = make_float4(a.x - pad, a.y - pad, a.z - pad, 0.0f);
	c.hi = make_float4(b.x + pad, b.y + pad, b.z + pad, 0.0f);
	store_slot(slots + (n - 1) + j, c);
	cl[j] = ~j;
	cbox[j] = c;
}

// Nearest neighbour of cluster i among [i-r, i+r].  Candidate i^1 is examined first and later
// candidates must be strictly better, so that runs of identical boxes pair up (0,1)(2,3).. instead
// of forming a chain that merges one pair per round.  force: skip the search, N[i] = i^1.
__device__ __forceinline__ int ploc_nearest(const CBox* __restrict__ win, int base, int i, int c, int r, int force)
{
	int j0 = i ^ 1;
	if (force) return j0 < c ? j0 : i;
	const CBox me = win[i - base];
	float best = __builtin_inff();
	int bj = i;
	if (j0 < c) { best = union_half_area(me, win[j0 - base]); bj = j0; }
	int lo = max(0, i - r), hi = min(c - 1, i + r);
	for (int j = lo; j <= hi; ++j) {
		if (j == i || j == j0) continue;
		float d = union_half_area(me, win[j - base]);
		if (d < best) { best = d; bj = j; }
	}
	return bj;
}

// The rounds are driven from the DEVICE: the number of clusters, the nodes made so far, which of the two cluster lists is
// current and the force flag live in `state` (PS_*), every kernel of a round reads them, k_ploc_advance moves them on.  The
// host queues several rounds at a time and reads the state back once per batch (a read-back per round was half the
// build: 22 rounds x a stream synchronise).  Rounds queued after the list has shrunk to PTAIL return at once.
enum { PS_C = 0, PS_CREATED = 1, PS_CUR = 2, PS_FORCE = 3, PS_ROUNDS = 4, PS_ERROR = 5 };

__global__ __launch_bounds__(PB) void k_ploc_nn_d(const CBox* __restrict__ cb0, const CBox* __restrict__ cb1,
                                                  const int* __restrict__ state, int r, int* __restrict__ nn)
{
	__shared__ CBox win[PB + 2 * PR_MAX];
	const int c = state[PS_C];
	if (c <= PTAIL || (int)(blockIdx.x * PB) >= c) return;
	const CBox* cbox = state[PS_CUR] ? cb1 : cb0;
	const int base = blockIdx.x * PB - r;
	for (int t = threadIdx.x; t < PB + 2 * r; t += PB) {
		int g = base + t;
		if (g >= 0 && g < c) win[t] = cbox[g];
	}
	__syncthreads();
	int i = blockIdx.x * PB + threadIdx.x;
	if (i < c) nn[i] = ploc_nearest(win, base, i, c, r, state[PS_FORCE]);
}

// flag: 0 dies (absorbed by its partner), 1 survives unchanged, 2 survives as the new parent node
__device__ __forceinline__ int ploc_flag(const int* nn, int i, int c)
{
	if (i >= c) return 0;
	int j = nn[i];
	if (j != i && nn[j] == i) return i < j ? 2 : 0;
	return 1;
}

// exclusive prefix over the workgroup of two flags at once; totals in tot_a / tot_b (all threads)
template <int THREADS>
__device__ __forceinline__ void block_scan2(bool a, bool b, int& pre_a, int& pre_b, int& tot_a, int& tot_b)
{
	__shared__ int wa[THREADS / 64], wb[THREADS / 64];
	const uint64_t ma = __ballot(a), mb = __ballot(b);
	const int wave = threadIdx.x >> 6;
	__syncthreads();  // protect wa/wb against the previous call's readers
	if ((threadIdx.x & 63) == 0) { wa[wave] = __popcll(ma); wb[wave] = __popcll(mb); }
	__syncthreads();
	int ba = 0, bb = 0, ta = 0, tb = 0;
	for (int w = 0; w < THREADS / 64; ++w) {
		if (w < wave) { ba += wa[w]; bb += wb[w]; }
		ta += wa[w]; tb += wb[w];
	}
	pre_a = ba + __popcll(ma & lanemask_lt());
	pre_b = bb + __popcll(mb & lanemask_lt());
	tot_a = ta; tot_b = tb;
}

__global__ __launch_bounds__(PB) void k_ploc_count_d(const int* __restrict__ nn, const int* __restrict__ state,
                                                    uint8_t* __restrict__ flag, int2* __restrict__ block_counts)
{
	const int c = state[PS_C];
	if (c <= PTAIL || (int)(blockIdx.x * PB) >= c) return;
	int i = blockIdx.x * PB + threadIdx.x;
	int f = ploc_flag(nn, i, c);
	if (i < c) flag[i] = (uint8_t)f;
	int pa, pb, ta, tb;
	block_scan2<PB>(f != 0, f == 2, pa, pb, ta, tb);
	if (threadIdx.x == 0) block_counts[blockIdx.x] = make_int2(ta, tb);
}

// One workgroup: exclusive scan of the per-block counts in place; totals[0] = survivors.
// state != nullptr: nb comes from the device-side cluster count (and nothing happens once the list is short).
__global__ __launch_bounds__(1024) void k_ploc_scan(int2* __restrict__ block_counts, int nb, int* __restrict__ totals,
                                                   const int* __restrict__ state)
{
	if (state) {
		const int c = state[PS_C];
		if (c <= PTAIL) return;
		nb = (c + PB - 1) / PB;
	}
	__shared__ int sa[1024], sb[1024];
	__shared__ int carry_a, carry_b;
	if (threadIdx.x == 0) { carry_a = 0; carry_b = 0; }
	__syncthreads();
	for (int base = 0; base < nb; base += 1024) {
		int b = base + threadIdx.x;
		int2 v = b < nb ? block_counts[b] : make_int2(0, 0);
		sa[threadIdx.x] = v.x; sb[threadIdx.x] = v.y;
		__syncthreads();
		for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
			int xa = threadIdx.x >= off ? sa[threadIdx.x - off] : 0;
			int xb = threadIdx.x >= off ? sb[threadIdx.x - off] : 0;
			__syncthreads();
			sa[threadIdx.x] += xa; sb[threadIdx.x] += xb;
			__syncthreads();
		}
		if (b < nb) block_counts[b] = make_int2(carry_a + sa[threadIdx.x] - v.x, carry_b + sb[threadIdx.x] - v.y);
		__syncthreads();
		if (threadIdx.x == 1023) { carry_a += sa[1023]; carry_b += sb[1023]; }
		__syncthreads();
	}
	if (threadIdx.x == 0) { totals[0] = carry_a; totals[1] = carry_b; }
}

__device__ __forceinline__ void ploc_link(int id, int a, int b, int* left, int* right, int* node_parent, int* leaf_parent)
{
	left[id] = a; right[id] = b;
	if (a >= 0) node_parent[a] = id; else leaf_parent[~a] = id;
	if (b >= 0) node_parent[b] = id; else leaf_parent[~b] = id;
}

__global__ __launch_bounds__(PB) void k_ploc_merge_d(const int* __restrict__ cl0, const int* __restrict__ cl1,
                                                    const CBox* __restrict__ cb0, const CBox* __restrict__ cb1,
                                                    const int* __restrict__ nn, const uint8_t* __restrict__ flag,
                                                    const int2* __restrict__ block_prefix, const int* __restrict__ state, int n,
                                                    int* __restrict__ left, int* __restrict__ right,
                                                    int* __restrict__ node_parent, int* __restrict__ leaf_parent,
                                                    BoxSlot* __restrict__ slots)
{
	const int c = state[PS_C];
	if (c <= PTAIL || (int)(blockIdx.x * PB) >= c) return;
	const int created = state[PS_CREATED], cur = state[PS_CUR];
	const int* cl = cur ? cl1 : cl0;
	const CBox* cbox = cur ? cb1 : cb0;
	int* cl_out = const_cast<int*>(cur ? cl0 : cl1);
	CBox* cbox_out = const_cast<CBox*>(cur ? cb0 : cb1);
	int i = blockIdx.x * PB + threadIdx.x;
	int f = i < c ? flag[i] : 0;
	int pa, pb, ta, tb;
	block_scan2<PB>(f != 0, f == 2, pa, pb, ta, tb);
	if (f == 0) return;
	const int2 bp = block_prefix[blockIdx.x];
	const int pos = bp.x + pa;
	if (f == 1) {
		cl_out[pos] = cl[i];
		cbox_out[pos] = cbox[i];
	} else {
		const int j = nn[i];
		const int id = (n - 2) - (created + bp.y + pb);
		ploc_link(id, cl[i], cl[j], left, right, node_parent, leaf_parent);
		CBox u = union_box(cbox[i], cbox[j]);
		store_slot(slots + id, u);
		cl_out[pos] = id;
		cbox_out[pos] = u;
	}
}

// End of a round: the survivors become the list.  No progress is only possible with exact distance ties; a forced round
// pairs neighbours.  A count that grew, or more than halved, is a broken round: PS_ERROR, and the rounds stop.
__global__ void k_ploc_advance(int* __restrict__ state, const int* __restrict__ totals)
{
	const int c = state[PS_C];
	if (c <= PTAIL) return;
	const int c_next = totals[0];
	if (c_next > c || c_next < (c + 1) / 2) { state[PS_ERROR] = c_next + 1; state[PS_C] = 0; return; }
	const int rounds = state[PS_ROUNDS] + 1;
	state[PS_ROUNDS] = rounds;
	state[PS_FORCE] = (c_next == c || rounds > 512) ? 1 : 0;
	state[PS_CREATED] = state[PS_CREATED] + (c - c_next);
	state[PS_C] = c_next;
	state[PS_CUR] = state[PS_CUR] ^ 1;
}

// The last rounds (c <= PTAIL) in one workgroup, clusters in LDS.
__global__ __launch_bounds__(PTAIL) void k_ploc_tail(const int* __restrict__ cl_in, const CBox* __restrict__ cbox_in, int c0,
                                                    int r, int n, int created0, int* __restrict__ left,
                                                    int* __restrict__ right, int* __restrict__ node_parent,
                                                    int* __restrict__ leaf_parent, BoxSlot* __restrict__ slots)
{
	__shared__ CBox sbox[2][PTAIL];
	__shared__ int scl[2][PTAIL];
	__shared__ int snn[PTAIL];
	const int i = threadIdx.x;
	if (i < c0) { sbox[0][i] = cbox_in[i]; scl[0][i] = cl_in[i]; }
	int c = c0, created = created0, cur = 0, force = 0;
	__syncthreads();
	while (c > 1) {
		if (i < c) snn[i] = ploc_nearest(sbox[cur], 0, i, c, r, force);
		__syncthreads();
		const int f = ploc_flag(snn, i, c);
		int pa, pb, ta, tb;
		block_scan2<PTAIL>(f != 0, f == 2, pa, pb, ta, tb);
		if (f == 1) {
			scl[cur ^ 1][pa] = scl[cur][i];
			sbox[cur ^ 1][pa] = sbox[cur][i];
		} else if (f == 2) {
			const int j = snn[i];
			const int id = (n - 2) - (created + pb);
			ploc_link(id, scl[cur][i], scl[cur][j], left, right, node_parent, leaf_parent);
			CBox u = union_box(sbox[cur][i], sbox[cur][j]);
			store_slot(slots + id, u);
			scl[cur ^ 1][pa] = id;
			sbox[cur ^ 1][pa] = u;
		}
		force = (tb == 0);  // no mutual pair (only possible with exact ties): pair neighbours next round
		created += tb;
		c = ta;
		cur ^= 1;
		__syncthreads();
	}
	if (i == 0) node_parent[0] = -1;
}

// 16-bit grid coordinate of a plane, rounded outward; the comparison is made against the float the
// traversal's arithmetic stands for (org + q * scale evaluated in float), so the quantised box
// contains the float box.
__device__ __forceinline__ uint32_t quant_lo(float v, float org, float scale)
{
	int q = (int)floorf((v - org) / scale);
	q = min(max(q, 0), 65535);
	while (q > 0 && __builtin_fmaf((float)q, scale, org) > v) --q;
	return (uint32_t)q;
}
__device__ __forceinline__ uint32_t quant_hi(float v, float org, float scale)
{
	int q = (int)ceilf((v - org) / scale);
	q = min(max(q, 0), 65535);
	while (q < 65535 && __builtin_fmaf((float)q, scale, org) < v) ++q;
	return (uint32_t)q;
}

__device__ __forceinline__ uint32_t qnode_ref(int child)
{
	return child >= 0 ? (uint32_t)child * (uint32_t)sizeof(QNode) : ~((uint32_t)(~child) * (uint32_t)sizeof(TriIsect));
}

// list == nullptr: every internal node, boxes already in `slots` (the builders that make boxes as they go).
// list != nullptr: the `count` nodes of one tree level; each also writes its own box = the union of its children's
// (the level-by-level refit), so that the next launch, one level up, finds it.
__device__ __forceinline__ void emit_node(BoxSlot* __restrict__ slots, int n, int i, bool own_box, const int* __restrict__ left,
                                          const int* __restrict__ right, const int* __restrict__ node_parent,
                                          BvhNode* __restrict__ nodes, const QFrame& qf, QNode* __restrict__ qnodes)
{
	int lc = left[i], rc = right[i];
	const BoxSlot* l = (lc >= 0) ? (slots + lc) : (slots + (n - 1) + (~lc));
	const BoxSlot* r = (rc >= 0) ? (slots + rc) : (slots + (n - 1) + (~rc));
	if (own_box) {
		BoxSlot* me = slots + i;
		for (int a = 0; a < 3; ++a) { me->mn[a] = fminf(l->mn[a], r->mn[a]); me->mx[a] = fmaxf(l->mx[a], r->mx[a]); }
	}
	BvhNode nd;
	nd.q0 = make_float4(l->mn[0], l->mn[1], l->mn[2], l->mx[0]);
	nd.q1 = make_float4(l->mx[1], l->mx[2], r->mn[0], r->mn[1]);
	nd.q2 = make_float4(r->mn[2], r->mx[0], r->mx[1], r->mx[2]);
	nd.left = lc; nd.right = rc; nd.parent = node_parent[i]; nd.pad = 0;
	nodes[i] = nd;
	uint32_t lq[6], rq[6];
	for (int k = 0; k < 3; ++k) {
		lq[k] = quant_lo(l->mn[k], qf.org[k], qf.scale[k]); lq[3 + k] = quant_hi(l->mx[k], qf.org[k], qf.scale[k]);
		rq[k] = quant_lo(r->mn[k], qf.org[k], qf.scale[k]); rq[3 + k] = quant_hi(r->mx[k], qf.org[k], qf.scale[k]);
	}
	QNode qn;
	qn.w0 = make_uint4(lq[0] | lq[3] << 16, lq[1] | lq[4] << 16, lq[2] | lq[5] << 16, rq[0] | rq[3] << 16);
	// child references as the traversal uses them: byte offset of the child's QNode (>= 0), or ~(byte offset of the leaf's TriIsect)
	qn.w1 = make_uint4(rq[1] | rq[4] << 16, rq[2] | rq[5] << 16, qnode_ref(lc), qnode_ref(rc));
	qnodes[i] = qn;
}

__global__ __launch_bounds__(256) void k_emit_nodes(BoxSlot* __restrict__ slots, int n, const int* __restrict__ list, int count,
                                                    const int* __restrict__ left, const int* __restrict__ right,
                                                    const int* __restrict__ node_parent, BvhNode* __restrict__ nodes,
                                                    QFrame qf, QNode* __restrict__ qnodes)
{
	int k = blockIdx.x * 256 + threadIdx.x;
	if (k >= count) return;
	emit_node(slots, n, list ? list[k] : k, list != nullptr, left, right, node_parent, nodes, qf, qnodes);
}

// The top of the tree in ONE launch: levels top.last down to 0, each at most TOP_BLOCK nodes, by one workgroup with a
// barrier between levels (a launch per level costs ~8 us and the upper levels hold a handful of nodes each).
constexpr int TOP_BLOCK = 1024;
constexpr int TOP_LEVELS = 32;
struct TopLevels {
	int last;                    // deepest level handled here
	int offset[TOP_LEVELS + 1];  // level d = list[offset[d], offset[d + 1])
};
__global__ __launch_bounds__(TOP_BLOCK) void k_emit_nodes_top(BoxSlot* __restrict__ slots, int n, const int* __restrict__ list,
                                                              TopLevels top, const int* __restrict__ left,
                                                              const int* __restrict__ right, const int* __restrict__ node_parent,
                                                              BvhNode* __restrict__ nodes, QFrame qf, QNode* __restrict__ qnodes)
{
	for (int d = top.last; d >= 0; --d) {
		for (int k = top.offset[d] + threadIdx.x; k < top.offset[d + 1]; k += TOP_BLOCK)
			emit_node(slots, n, list[k], true, left, right, node_parent, nodes, qf, qnodes);
		__threadfence_block();
		__syncthreads();
	}
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

// Refit quality: sum of the half-areas of the internal nodes' boxes (slots [0, n-1)) -- the SAH cost of the
// tree's inner part up to the constant root area, which the host divides by.
__global__ __launch_bounds__(256) void k_area_sum(const BoxSlot* __restrict__ slots, int n_internal, float* __restrict__ sum)
{
	__shared__ float ws[4];
	int i = blockIdx.x * 256 + threadIdx.x;
	float a = 0.0f;
	if (i < n_internal) {
		const BoxSlot& s = slots[i];
		const float dx = s.mx[0] - s.mn[0], dy = s.mx[1] - s.mn[1], dz = s.mx[2] - s.mn[2];
		a = dx * dy + (dy * dz + dz * dx);
	}
	for (int off = 32; off >= 1; off >>= 1) a += __shfl_xor(a, off);
	if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = a;
	__syncthreads();
	if (threadIdx.x == 0) atomicAdd(sum, (ws[0] + ws[1]) + (ws[2] + ws[3]));
}

}  // namespace

// Area sum of the internal nodes relative to the root's (node 0) area; the slots must be complete.
static int tree_cost(rt_hip_ctx* ctx, const BoxSlot* slots, int n, float* cost)
{
	hipStream_t st = ctx->stream;
	RT_CHECK(ctx->d_bounds.reserve(64));
	float* d_sum = ctx->d_bounds.as<float>() + 8;
	RT_HIP_CHECK(hipMemsetAsync(d_sum, 0, 4, st));
	hipLaunchKernelGGL(k_area_sum, dim3((n - 1 + 255) / 256), dim3(256), 0, st, slots, n - 1, d_sum);
	RT_HIP_CHECK(hipGetLastError());
	float* h = (float*)ctx->h_small + 512;  // pinned
	RT_HIP_CHECK(hipMemcpyAsync(h, d_sum, 4, hipMemcpyDeviceToHost, st));
	RT_HIP_CHECK(hipMemcpyAsync(h + 1, slots, 24, hipMemcpyDeviceToHost, st));  // the root's box
	RT_HIP_CHECK(hipStreamSynchronize(st));
	const float dx = h[4] - h[1], dy = h[5] - h[2], dz = h[6] - h[3];
	const float root = dx * dy + (dy * dz + dz * dx);
	*cost = root > 0.0f ? h[0] / root : 0.0f;
	return RT_HIP_OK;
}

// The internal nodes grouped by depth (ctx->d_level_nodes, ctx->level_offsets): computed once per topology, at the
// first refit (or at the build, for the LBVH builder, which makes its boxes this way too).
static int compute_levels(rt_hip_ctx* ctx, int n, const int* node_parent)
{
	hipStream_t st = ctx->stream;
	const int ni = n - 1;
	RT_CHECK(ctx->d_level_nodes.reserve(((size_t)2 * ni + 2 * MAX_LEVELS) * sizeof(int)));
	int* level_nodes = ctx->d_level_nodes.as<int>();
	int* depth = level_nodes + ni;
	int* hist = depth + ni;
	int* cursor = hist + MAX_LEVELS;
	RT_HIP_CHECK(hipMemsetAsync(hist, 0, MAX_LEVELS * sizeof(int), st));
	hipLaunchKernelGGL(k_node_depth, dim3((ni + 255) / 256), dim3(256), 0, st, node_parent, ni, depth, hist);
	RT_HIP_CHECK(hipGetLastError());
	int* h = (int*)ctx->h_small + 256;  // pinned
	RT_HIP_CHECK(hipMemcpyAsync(h, hist, MAX_LEVELS * sizeof(int), hipMemcpyDeviceToHost, st));
	RT_HIP_CHECK(hipStreamSynchronize(st));
	if (h[MAX_LEVELS - 1] != 0) { rt_set_error("BVH deeper than %d levels", MAX_LEVELS - 1); return RT_HIP_ERR_LIMIT; }
	ctx->level_offsets.assign(1, 0);
	for (int d = 0; d < MAX_LEVELS && ctx->level_offsets.back() < ni; ++d) ctx->level_offsets.push_back(ctx->level_offsets.back() + h[d]);
	if (ctx->level_offsets.back() != ni) { rt_set_error("BVH levels cover %d of %d nodes", ctx->level_offsets.back(), ni); return RT_HIP_ERR_STATE; }
	for (size_t d = 0; d + 1 < ctx->level_offsets.size(); ++d) h[MAX_LEVELS + d] = ctx->level_offsets[d];
	RT_HIP_CHECK(hipMemcpyAsync(cursor, h + MAX_LEVELS, (ctx->level_offsets.size() - 1) * sizeof(int), hipMemcpyHostToDevice, st));
	hipLaunchKernelGGL(k_level_scatter, dim3((ni + 255) / 256), dim3(256), 0, st, depth, ni, cursor, level_nodes);
	RT_HIP_CHECK(hipGetLastError());
	RT_HIP_CHECK(hipStreamSynchronize(st));  // h_small is reused
	return RT_HIP_OK;
}

// Leaf boxes from the triangle boxes, then every level from the deepest to the root: box, float node and
// quantised node of each internal node in one pass (ctx->qframe must be final).
static int refit_by_levels(rt_hip_ctx* ctx, int n, float pad, const uint32_t* order, const int* left, const int* right,
                           const int* node_parent, BoxSlot* slots)
{
	hipStream_t st = ctx->stream;
	hipLaunchKernelGGL(k_leaf_slots, dim3((n + 255) / 256), dim3(256), 0, st, ctx->d_tri_box.as<float4>(), order, n, pad, slots);
	const int* level_nodes = ctx->d_level_nodes.as<int>();
	// the levels from the root down to the first one with more than TOP_BLOCK nodes go into one launch at the end
	const int nlevels = (int)ctx->level_offsets.size() - 1;
	TopLevels top;
	top.last = -1;
	while (top.last + 1 < nlevels && top.last + 1 < TOP_LEVELS &&
	       ctx->level_offsets[top.last + 2] - ctx->level_offsets[top.last + 1] <= TOP_BLOCK) ++top.last;
	for (int d = 0; d <= TOP_LEVELS; ++d) top.offset[d] = ctx->level_offsets[std::min(d, nlevels)];
	for (int d = nlevels - 1; d > top.last; --d) {
		const int lo = ctx->level_offsets[d], cnt = ctx->level_offsets[d + 1] - lo;
		if (cnt <= 0) continue;
		hipLaunchKernelGGL(k_emit_nodes, dim3((cnt + 255) / 256), dim3(256), 0, st, slots, n, level_nodes + lo, cnt, left, right,
		                   node_parent, ctx->d_nodes.as<BvhNode>(), ctx->qframe, ctx->d_qnodes.as<QNode>());
	}
	if (top.last >= 0)
		hipLaunchKernelGGL(k_emit_nodes_top, dim3(1), dim3(TOP_BLOCK), 0, st, slots, n, level_nodes, top, left, right, node_parent,
		                   ctx->d_nodes.as<BvhNode>(), ctx->qframe, ctx->d_qnodes.as<QNode>());
	RT_HIP_CHECK(hipGetLastError());
	return RT_HIP_OK;
}

static void set_qframe(rt_hip_ctx* ctx, float pad)
{
	// quantisation grid: the padded scene box, 65535 steps per axis (a flat axis gets a unit step)
	for (int k = 0; k < 3; ++k) {
		ctx->qframe.org[k] = ctx->bounds[k] - 2.0f * pad;
		float e = (ctx->bounds[3 + k] + 2.0f * pad) - ctx->qframe.org[k];
		ctx->qframe.scale[k] = e > 0.0f ? e / 65535.0f : 1.0f;
		ctx->qframe.reach[k] = std::fabs(ctx->qframe.org[k]) + 65535.0f * ctx->qframe.scale[k];
	}
}

// PLOC rounds on the sorted leaves: fills left/right/node_parent/leaf_parent and every node's box slot.
static int build_ploc(rt_hip_ctx* ctx, int n, float pad, const uint32_t* order, int* left, int* right,
                      int* node_parent, int* leaf_parent, BoxSlot* slots)
{
	hipStream_t st = ctx->stream;
	// search radius: 8 measured best on atrium-262k (nodes/ray 28.1; 16: 30.3, 32: 28.5, 64: 30.7; LBVH 31.4)
	const int r = std::min(PR_MAX, std::max(1, ctx->tuning.ploc_radius));
	const int nb0 = (n + PB - 1) / PB;
	// scratch: 2 cluster lists, 2 cluster box lists, nn, flag, block counts, totals
	size_t bytes = (size_t)n * (2 * 4 + 2 * sizeof(CBox) + 4 + 1) + (size_t)nb0 * sizeof(int2) + 256 + 64;
	RT_CHECK(ctx->d_ploc.reserve(bytes));
	char* base = ctx->d_ploc.as<char>();
	CBox* cbox[2] = {(CBox*)base, (CBox*)base + n};
	int* cl[2] = {(int*)(cbox[1] + n), (int*)(cbox[1] + n) + n};
	int* nn = cl[1] + n;
	int2* block_counts = (int2*)(nn + n);
	int* totals = (int*)(block_counts + nb0 + 1);
	uint8_t* flag = (uint8_t*)(totals + 16);
	hipLaunchKernelGGL(k_ploc_init, dim3((n + 255) / 256), dim3(256), 0, st, ctx->d_tri_box.as<float4>(), order, n, pad,
	                   slots, cl[0], cbox[0]);
	int c = n, created = 0, cur = 0;
	int* h_state = (int*)ctx->h_small;   // pinned
	int* state = totals + 8;
	if (c > PTAIL) {
		const int init[6] = {n, 0, 0, 0, 0, 0};
		memcpy(h_state, init, sizeof(init));
		RT_HIP_CHECK(hipMemcpyAsync(state, h_state, sizeof(init), hipMemcpyHostToDevice, st));
		RT_HIP_CHECK(hipStreamSynchronize(st));   // h_state is read back into below
	}
	constexpr int ROUNDS_PER_BATCH = 8;
	while (c > PTAIL) {
		const int nb = (c + PB - 1) / PB;         // c only shrinks: the grid of this batch covers all of its rounds
		for (int k = 0; k < ROUNDS_PER_BATCH; ++k) {
			hipLaunchKernelGGL(k_ploc_nn_d, dim3(nb), dim3(PB), 0, st, cbox[0], cbox[1], state, r, nn);
			hipLaunchKernelGGL(k_ploc_count_d, dim3(nb), dim3(PB), 0, st, nn, state, flag, block_counts);
			hipLaunchKernelGGL(k_ploc_scan, dim3(1), dim3(1024), 0, st, block_counts, nb, totals, (const int*)state);
			hipLaunchKernelGGL(k_ploc_merge_d, dim3(nb), dim3(PB), 0, st, cl[0], cl[1], cbox[0], cbox[1], nn, flag, block_counts, state, n,
			                   left, right, node_parent, leaf_parent, slots);
			hipLaunchKernelGGL(k_ploc_advance, dim3(1), dim3(1), 0, st, state, totals);
		}
		RT_HIP_CHECK(hipGetLastError());
		RT_HIP_CHECK(hipMemcpyAsync(h_state, state, 6 * sizeof(int), hipMemcpyDeviceToHost, st));
		RT_HIP_CHECK(hipStreamSynchronize(st));
		if (h_state[PS_ERROR]) { rt_set_error("PLOC round produced %d clusters from at most %d", h_state[PS_ERROR] - 1, c); return RT_HIP_ERR_STATE; }
		c = h_state[PS_C]; created = h_state[PS_CREATED]; cur = h_state[PS_CUR];
	}
	hipLaunchKernelGGL(k_ploc_tail, dim3(1), dim3(PTAIL), 0, st, cl[cur], cbox[cur], c, r, n, created, left, right,
	                   node_parent, leaf_parent, slots);
	RT_HIP_CHECK(hipGetLastError());
	return RT_HIP_OK;
}

// ---- binned SAH on the host (RT_HIP_ACCEL_SAH) ------------------------------------------------------
// Top-down surface-area-heuristic build for scenes that stay put: the triangle boxes come back from the
// device, one host thread splits [lo, hi) at the cheapest of 3 x (SAH_BINS - 1) centroid planes (every
// candidate of a range of <= SAH_SWEEP, sorted per axis) down to single triangles, and the topology goes
// back in the arrays the device builders fill, so node emission, quantisation and traversal are shared.
// Node ids are handed out in pre-order: root 0, parents below their children, a left subtree contiguous.
namespace {

constexpr int SAH_BINS = 32;
constexpr int SAH_SWEEP = 16;

struct HBox {
	float lo[3], hi[3];
	void clear() { for (int k = 0; k < 3; ++k) { lo[k] = __builtin_inff(); hi[k] = -__builtin_inff(); } }
	void grow(const float* mn, const float* mx) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], mn[k]); hi[k] = std::max(hi[k], mx[k]); } }
	void grow(const HBox& b) { grow(b.lo, b.hi); }
	float half_area() const
	{
		float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
		return dx * dy + (dy * dz + dz * dx);
	}
};

// Best split of idx[lo, hi) by binned SAH; returns the position of the first element of the right side.
int sah_split(const std::vector<float4>& box, std::vector<uint32_t>& idx, int lo, int hi, std::vector<HBox>& scratch)
{
	const int count = hi - lo;
	auto centre = [&](uint32_t p, int k) {
		const float* mn = &box[2 * (size_t)p].x; const float* mx = &box[2 * (size_t)p + 1].x;
		return 0.5f * (mn[k] + mx[k]);
	};
	float cmin[3], cmax[3];
	for (int k = 0; k < 3; ++k) { cmin[k] = __builtin_inff(); cmax[k] = -__builtin_inff(); }
	for (int i = lo; i < hi; ++i)
		for (int k = 0; k < 3; ++k) { float c = centre(idx[i], k); cmin[k] = std::min(cmin[k], c); cmax[k] = std::max(cmax[k], c); }
	float best = __builtin_inff();
	int best_axis = -1, best_pos = 0;
	if (count <= SAH_SWEEP) {
		// every candidate: sort by centre, suffix boxes, sweep
		uint32_t tmp[3][SAH_SWEEP];
		HBox* suf = scratch.data();
		for (int k = 0; k < 3; ++k) {
			if (!(cmax[k] > cmin[k])) continue;
			std::copy(idx.begin() + lo, idx.begin() + hi, tmp[k]);
			std::stable_sort(tmp[k], tmp[k] + count, [&](uint32_t a, uint32_t b) { return centre(a, k) < centre(b, k); });
			HBox acc; acc.clear();
			for (int i = count - 1; i >= 1; --i) { acc.grow(&box[2 * (size_t)tmp[k][i]].x, &box[2 * (size_t)tmp[k][i] + 1].x); suf[i] = acc; }
			acc.clear();
			for (int i = 1; i < count; ++i) {
				acc.grow(&box[2 * (size_t)tmp[k][i - 1]].x, &box[2 * (size_t)tmp[k][i - 1] + 1].x);
				float cost = acc.half_area() * (float)i + suf[i].half_area() * (float)(count - i);
				if (cost < best) { best = cost; best_axis = k; best_pos = i; }
			}
		}
		if (best_axis < 0) return lo + count / 2;
		std::copy(tmp[best_axis], tmp[best_axis] + count, idx.begin() + lo);
		return lo + best_pos;
	}
	HBox* bins = scratch.data();              // [3][SAH_BINS]
	HBox* suf = scratch.data() + 3 * SAH_BINS; // [SAH_BINS]
	int cnt[3][SAH_BINS];
	float scale[3];
	for (int k = 0; k < 3; ++k) {
		scale[k] = cmax[k] > cmin[k] ? (float)SAH_BINS / (cmax[k] - cmin[k]) : 0.0f;
		for (int b = 0; b < SAH_BINS; ++b) { bins[k * SAH_BINS + b].clear(); cnt[k][b] = 0; }
	}
	auto bin_of = [&](uint32_t p, int k) { return std::min(SAH_BINS - 1, std::max(0, (int)((centre(p, k) - cmin[k]) * scale[k]))); };
	for (int i = lo; i < hi; ++i) {
		const uint32_t p = idx[i];
		const float* mn = &box[2 * (size_t)p].x; const float* mx = &box[2 * (size_t)p + 1].x;
		for (int k = 0; k < 3; ++k) {
			if (scale[k] == 0.0f) continue;
			int b = bin_of(p, k);
			bins[k * SAH_BINS + b].grow(mn, mx); ++cnt[k][b];
		}
	}
	for (int k = 0; k < 3; ++k) {
		if (scale[k] == 0.0f) continue;
		HBox acc; acc.clear();
		for (int b = SAH_BINS - 1; b >= 1; --b) { if (cnt[k][b]) acc.grow(bins[k * SAH_BINS + b]); suf[b] = acc; }
		acc.clear();
		int nl = 0;
		for (int b = 1; b < SAH_BINS; ++b) {
			if (cnt[k][b - 1]) acc.grow(bins[k * SAH_BINS + b - 1]);
			nl += cnt[k][b - 1];
			if (nl == 0 || nl == count) continue;
			float cost = acc.half_area() * (float)nl + suf[b].half_area() * (float)(count - nl);
			if (cost < best) { best = cost; best_axis = k; best_pos = b; }
		}
	}
	if (best_axis < 0) return lo + count / 2;   // all centres coincide
	auto mid = std::partition(idx.begin() + lo, idx.begin() + hi, [&](uint32_t p) { return bin_of(p, best_axis) < best_pos; });
	int m = (int)(mid - idx.begin());
	return (m == lo || m == hi) ? lo + count / 2 : m;
}

}  // namespace

static int build_sah_host(rt_hip_ctx* ctx, int n, float pad, uint32_t* d_order, int* d_left, int* d_right,
                          int* d_node_parent, int* d_leaf_parent, BoxSlot* d_slots)
{
	hipStream_t st = ctx->stream;
	std::vector<float4> box(2 * (size_t)n);
	RT_HIP_CHECK(hipMemcpyAsync(box.data(), ctx->d_tri_box.p, box.size() * sizeof(float4), hipMemcpyDeviceToHost, st));
	RT_HIP_CHECK(hipStreamSynchronize(st));
	std::vector<uint32_t> idx(n);
	for (int i = 0; i < n; ++i) idx[i] = (uint32_t)i;
	std::vector<int> left(n - 1), right(n - 1), node_parent(n - 1), leaf_parent(n);
	std::vector<HBox> scratch(4 * SAH_BINS);
	struct Job { int lo, hi, parent, is_right; };
	std::vector<Job> stack;
	stack.push_back({0, n, -1, 0});
	int next_id = 0;
	while (!stack.empty()) {
		const Job j = stack.back();
		stack.pop_back();
		const int id = next_id++;
		node_parent[id] = j.parent;
		if (j.parent >= 0) (j.is_right ? right : left)[j.parent] = id;
		const int mid = (j.hi - j.lo == 2) ? j.lo + 1 : sah_split(box, idx, j.lo, j.hi, scratch);
		// right first onto the stack, so the left subtree takes the next ids
		if (j.hi - mid == 1) { right[id] = ~mid; leaf_parent[mid] = id; } else stack.push_back({mid, j.hi, id, 1});
		if (mid - j.lo == 1) { left[id] = ~j.lo; leaf_parent[j.lo] = id; } else stack.push_back({j.lo, mid, id, 0});
	}
	if (next_id != n - 1) { rt_set_error("SAH build made %d nodes for %d triangles", next_id, n); return RT_HIP_ERR_STATE; }
	// boxes: leaves padded like the device builders', internal nodes bottom-up (children have higher ids)
	std::vector<BoxSlot> slots(2 * (size_t)n - 1);
	for (int j = 0; j < n; ++j) {
		const float4 a = box[2 * (size_t)idx[j]], b = box[2 * (size_t)idx[j] + 1];
		BoxSlot& s = slots[(size_t)(n - 1) + j];
		s.mn[0] = a.x - pad; s.mn[1] = a.y - pad; s.mn[2] = a.z - pad;
		s.mx[0] = b.x + pad; s.mx[1] = b.y + pad; s.mx[2] = b.z + pad;
	}
	for (int i = n - 2; i >= 0; --i) {
		const BoxSlot& l = left[i] >= 0 ? slots[left[i]] : slots[(size_t)(n - 1) + (~left[i])];
		const BoxSlot& r = right[i] >= 0 ? slots[right[i]] : slots[(size_t)(n - 1) + (~right[i])];
		for (int k = 0; k < 3; ++k) { slots[i].mn[k] = std::min(l.mn[k], r.mn[k]); slots[i].mx[k] = std::max(l.mx[k], r.mx[k]); }
	}
	RT_HIP_CHECK(hipMemcpyAsync(d_order, idx.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
	RT_HIP_CHECK(hipMemcpyAsync(d_left, left.data(), (size_t)(n - 1) * 4, hipMemcpyHostToDevice, st));
	RT_HIP_CHECK(hipMemcpyAsync(d_right, right.data(), (size_t)(n - 1) * 4, hipMemcpyHostToDevice, st));
	RT_HIP_CHECK(hipMemcpyAsync(d_node_parent, node_parent.data(), (size_t)(n - 1) * 4, hipMemcpyHostToDevice, st));
	RT_HIP_CHECK(hipMemcpyAsync(d_leaf_parent, leaf_parent.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
	RT_HIP_CHECK(hipMemcpyAsync(d_slots, slots.data(), slots.size() * sizeof(BoxSlot), hipMemcpyHostToDevice, st));
	RT_HIP_CHECK(hipStreamSynchronize(st));   // the host vectors die here
	return RT_HIP_OK;
}

int rt_build_lbvh(rt_hip_ctx* ctx)
{
	hipStream_t st = ctx->stream;
	const int n = ctx->num_tris;
	const int g = (n + 255) / 256;
	if (n > RT_MAX_BVH_TRIS) { rt_set_error("%d triangles: the BVH takes at most %d (32-bit child references)", n, RT_MAX_BVH_TRIS); return RT_HIP_ERR_LIMIT; }
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
	const bool sah = ctx->cfg.accel == RT_HIP_ACCEL_SAH;
	if (!sah) {
		// one cubic grid over the longest extent keeps cells isotropic
		float s = 1.0f / emax;
		hipLaunchKernelGGL(k_morton, dim3(g), dim3(256), 0, st, ctx->d_tri_box.as<float4>(), n, ctx->bounds[0],
		                   ctx->bounds[1], ctx->bounds[2], s, s, s, keys, order);
		RT_HIP_CHECK(hipGetLastError());
		RT_CHECK(rt_radix_sort_pairs64(ctx, keys, ctx->d_morton_tmp.as<uint64_t>(), order,
		                               ctx->d_order_tmp.as<uint32_t>(), n, 63));
	}

	// topology arrays: left, right, node_parent (n-1 each), leaf_parent (n); the measured depth behind d_flags' first n words
	size_t ints = (size_t)4 * n + 8;
	RT_CHECK(ctx->d_parent_leaf.reserve(ints * 4));
	int* left = ctx->d_parent_leaf.as<int>();
	int* right = left + n;
	int* node_parent = right + n;
	int* leaf_parent = node_parent + n;
	RT_CHECK(ctx->d_flags.reserve((size_t)(n + 16) * 4));
	unsigned int* arrive = ctx->d_flags.as<unsigned int>();
	int* max_depth = (int*)(arrive + n);
	RT_HIP_CHECK(hipMemsetAsync(max_depth, 0, 64, st));
	RT_CHECK(ctx->d_misc.reserve((size_t)(2 * n) * sizeof(BoxSlot)));
	BoxSlot* slots = ctx->d_misc.as<BoxSlot>();

	float pad = 1e-5f * emax;
	const bool ploc = ctx->cfg.accel == RT_HIP_ACCEL_PLOC ||
	                  (ctx->cfg.accel == RT_HIP_ACCEL_AUTO && ctx->tuning.auto_accel == RT_HIP_ACCEL_PLOC);
	if (sah) {
		RT_CHECK(build_sah_host(ctx, n, pad, order, left, right, node_parent, leaf_parent, slots));
	} else if (ploc) {
		RT_CHECK(build_ploc(ctx, n, pad, order, left, right, node_parent, leaf_parent, slots));
	} else {
		hipLaunchKernelGGL(k_hierarchy, dim3(g), dim3(256), 0, st, keys, n, left, right, node_parent, leaf_parent);
		RT_HIP_CHECK(hipGetLastError());
	}
	RT_CHECK(ctx->d_nodes.reserve((size_t)(n - 1) * sizeof(BvhNode)));
	RT_CHECK(ctx->d_qnodes.reserve((size_t)(n - 1) * sizeof(QNode)));
	set_qframe(ctx, pad);
	ctx->level_offsets.clear();  // new topology: the refit's level lists are made when first needed
	if (!sah && !ploc) {
		// the Karras hierarchy has no boxes yet: make them level by level, nodes in the same pass
		RT_CHECK(compute_levels(ctx, n, node_parent));
		RT_CHECK(refit_by_levels(ctx, n, pad, order, left, right, node_parent, slots));
	} else {
		hipLaunchKernelGGL(k_emit_nodes, dim3(g), dim3(256), 0, st, slots, n, (const int*)nullptr, n - 1, left, right, node_parent,
		                   ctx->d_nodes.as<BvhNode>(), ctx->qframe, ctx->d_qnodes.as<QNode>());
	}
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
	// kept for refits: the leaf order (d_order is scratch that the photon grid reuses) and the tree's area sum
	RT_CHECK(ctx->d_leaf_order.reserve((size_t)n * 4));
	RT_HIP_CHECK(hipMemcpyAsync(ctx->d_leaf_order.p, order, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
	RT_CHECK(tree_cost(ctx, slots, n, &ctx->bvh_cost_built));
	ctx->bvh_cost_now = ctx->bvh_cost_built;
	if (depth > ctx->tuning.max_bvh_depth) {
		// pathological input (PLOC has no depth bound on masses of coincident triangles): the tree is not used,
		// brute force over the same records serves the scene
		if (!ctx->warned_depth) {
			ctx->warned_depth = true;
			fprintf(stderr, "librt_hip: BVH depth %d exceeds the traversal stack limit (%d); tracing %d triangles by brute force\n",
			        depth, ctx->tuning.max_bvh_depth, n);
		}
		ctx->use_bvh = 0;
		ctx->bvh_depth = 0;
		return RT_HIP_OK;
	}
	ctx->use_bvh = 1;
	RT_CHECK(rt_reserve_spill(ctx));
	return RT_HIP_OK;
}

// Overflow part of the traversal stacks: the entries beyond the LDS-resident ones, one column per thread of the
// largest trace grid (rt_trace_blocks never launches more than max_trace_threads).
int rt_reserve_spill(rt_hip_ctx* ctx)
{
	const int over = ctx->bvh_depth + 2 - ctx->tuning.lds_stack_depth;  // = SceneView::stack_depth - lds depth
	// (one set of columns per lane of the frame pipeline: frames of different lanes trace at the same time)
	if (ctx->use_bvh && over > 0) RT_CHECK(ctx->d_spill.reserve((size_t)over * ctx->max_trace_threads * sizeof(int) * (size_t)std::max(1, ctx->tuning.frame_streams)));
	return RT_HIP_OK;
}

// Refit: the boxes of the kept topology (left / right / parents in d_parent_leaf, leaf order in d_leaf_order) from
// the triangle boxes k_soup has just rewritten; new quantisation grid over the new scene bounds, new traversal
// nodes, leaf records permuted into the kept order.  Everything a build writes except the topology.
int rt_refit_lbvh(rt_hip_ctx* ctx, const TriIsect* d_isect_src, float* cost)
{
	hipStream_t st = ctx->stream;
	const int n = ctx->num_tris;
	const int g = (n + 255) / 256;
	int* left = ctx->d_parent_leaf.as<int>();
	int* right = left + n;
	int* node_parent = right + n;
	BoxSlot* slots = ctx->d_misc.as<BoxSlot>();
	const uint32_t* order = ctx->d_leaf_order.as<uint32_t>();
	float emax = 0.0f;
	for (int k = 0; k < 3; ++k) emax = std::max(emax, ctx->bounds[3 + k] - ctx->bounds[k]);
	if (!(emax > 0.0f)) emax = 1.0f;
	const float pad = 1e-5f * emax;
	if (ctx->level_offsets.empty()) RT_CHECK(compute_levels(ctx, n, node_parent));
	set_qframe(ctx, pad);
	RT_CHECK(refit_by_levels(ctx, n, pad, order, left, right, node_parent, slots));
	hipLaunchKernelGGL(k_permute_tris, dim3(g), dim3(256), 0, st, d_isect_src, order, n, ctx->d_tri_isect.as<TriIsect>());
	RT_HIP_CHECK(hipGetLastError());
	RT_CHECK(tree_cost(ctx, slots, n, cost));
	return RT_HIP_OK;
}
