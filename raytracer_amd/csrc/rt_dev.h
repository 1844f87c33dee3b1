// raytracer_amd/csrc/rt_dev.h -- device-side types, exact-arithmetic helpers and the traversal
// core shared by every kernel file (gfx950 only; compiled with -ffp-contract=off so that the
// intersection arithmetic is the reference's operation-for-operation; FMA is used only where it
// is written out as __builtin_fmaf, i.e. in the conservative box tests).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define RT_MISS 0xFFFFFFFFu
#define RT_MAX_BVH_TRIS (1 << 25)  // 32-bit child references: 2^25 x 48-byte leaves and 2^25 x 32-byte nodes stay below 2^31
#define RT_WAVE 64

struct f3 {
	float x, y, z;
};

__host__ __device__ __forceinline__ f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__host__ __device__ __forceinline__ f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__host__ __device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__host__ __device__ __forceinline__ f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__host__ __device__ __forceinline__ f3 neg3(f3 a) { return mk3(-a.x, -a.y, -a.z); }
// Reduction order of Eigen's 3-vector dot (Redux.h:98-101): a0*b0 + (a1*b1 + a2*b2).
__host__ __device__ __forceinline__ float dot3(f3 a, f3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
__host__ __device__ __forceinline__ f3 cross3(f3 a, f3 b)
{
	return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// IEEE-correct square root: HIP's __fsqrt_rn lowers to the *native* (approximate) sqrt unless
// OCML_BASIC_ROUNDED_OPERATIONS is defined; sqrtf under
// -fhip-fp32-correctly-rounded-divide-sqrt is correctly rounded, which parity needs.
__device__ __forceinline__ float rt_sqrtf(float x) { return __builtin_sqrtf(x); }
// Eigen's v / s multiplies by the reciprocal (Functors.h:455-456); both IEEE-rounded.
__device__ __forceinline__ f3 div3(f3 a, float s) { float inv = 1.0f / s; return a * inv; }
__device__ __forceinline__ float norm3(f3 a) { return rt_sqrtf(dot3(a, a)); }
__device__ __forceinline__ f3 normalized3(f3 a) { return div3(a, norm3(a)); }
// (1-u-v)*a + u*b + v*c evaluated left to right (rt_raytracer.cpp:520, rt_accel_array.cpp:127)
__device__ __forceinline__ f3 bary3(float u, float v, f3 a, f3 b, f3 c)
{
	float w = (1.0f - u - v);
	return (a * w + b * u) + c * v;
}
__device__ __forceinline__ f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }
__device__ __forceinline__ void st3(float* p, f3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

// ---- geometry records in HBM ---------------------------------------------------------------
// 64-byte BVH node: both children's boxes live in the parent, one aligned 64-B fetch per visit.
//   q0 = (l.min.x, l.min.y, l.min.z, l.max.x)
//   q1 = (l.max.y, l.max.z, r.min.x, r.min.y)
//   q2 = (r.min.z, r.max.x, r.max.y, r.max.z)
//   q3 = (bits left, bits right, bits parent, 0)   child >= 0: internal node, child < 0: ~leaf slot
struct __attribute__((aligned(64))) BvhNode {
	float4 q0, q1, q2;
	int left, right, parent, pad;
};
static_assert(sizeof(BvhNode) == 64, "BVH node is one 64-byte line");

// 32-byte traversal node: the same two child boxes on a 16-bit grid over the (padded) scene box,
// mins rounded down and maxes up, so each quantised box contains the float box.  The traversal
// kernels are bound by the vector-memory pipeline (TA/TD 81-87 % busy with 64-byte nodes,
// profiles/), not by HBM or the VALU; halving the bytes per visit is what moves them.  The float
// nodes (BvhNode) remain the build's output and what rt_hip_get_bvh returns.
// A plane pair (min | max << 16) per axis and child, so that one rotate by 0 or 16 bits puts the plane
// the ray enters through in the low half-word:
//   w0 = (l.x, l.y, l.z, r.x)   w1 = (r.y, r.z, ref left, ref right)        with  c.k = c.min.k | c.max.k << 16
// A child reference is what the traversal adds to a base pointer: the byte offset of the child's QNode (>= 0), or
// ~(byte offset of the leaf's TriIsect) (< 0), so that a fetch needs no address arithmetic beyond the load's own
// scalar-base + 32-bit-offset form.  Offsets stay below 0x7ffffffc (the traversal's sentinels): RT_MAX_BVH_TRIS.
struct __attribute__((aligned(32))) QNode {
	uint4 w0, w1;
};
static_assert(sizeof(QNode) == 32, "quantised node is half a BvhNode");

struct QFrame {  // plane = org + q * scale, per axis
	float org[3];
	float scale[3];
	float reach[3];  // |org| + 65535 * scale: the largest |plane coordinate| of the axis (error bound of the slab test)
};

// 48-byte intersection record in leaf (Morton) order: the M-T operands, nothing else.
//   a = (v0.xyz, bits prim id), b = (e1.xyz, 0), c = (e2.xyz, 0)
struct __attribute__((aligned(16))) TriIsect {
	float4 a, b, c;
};
static_assert(sizeof(TriIsect) == 48, "triangle fetch is 48 bytes");

// Per-primitive (original id order) attributes for hit-record construction.
//   p0 = (v0.xyz, bits material), p1 = (v1.xyz, 0), p2 = (v2.xyz, 0)            48 B
//   n0.xyz n1.xyz n2.xyz t0.xy t1.xy t2.xy pad                                    64 B
struct __attribute__((aligned(16))) TriPos {
	float4 p0, p1, p2;
};
struct __attribute__((aligned(16))) TriAttr {
	float n0[3], n1[3], n2[3];
	float t0[2], t1[2], t2[2];
	float pad;
};
static_assert(sizeof(TriAttr) == 64, "attribute record is 64 bytes");

struct SphereRec {
	float cx, cy, cz, r;
};

struct MaterialRec {  // device copy of rt_material without the texture slot
	float diffuse[3];
	float specular[3];
	float shininess, reflectivity, alpha, ior;
	float pad[2];
};
static_assert(sizeof(MaterialRec) == 48, "material record");

// Material::texture (data/data.h:25): one record per material beside the material table, texels of
// all textures in one RGBA8 pool (little-endian words, row 0 = bottom scanline as FreeImage_GetBits
// and glTexImage2D see it).  w == 0: the material has no texture.
struct TexRec {
	uint32_t offset;  // first texel in the pool
	int w, h;
	int pad;
};

struct LightRec {
	float position[3];
	float color[3];
	float energy;
	float pad;
};

struct Hit {
	uint32_t prim;
	float t, u, v;
};

struct TraceParams {
	float eps_det, eps_t, t_max;
};

// ---- Moeller-Trumbore with backface culling, the reference's operation order ----------------
// rt_raytracer.cpp:326-368 / rt_accel_array.cpp:87-121.  Returns true and updates (t,u,v) when
// the triangle is hit in front of the current best; `prim` breaks exact ties towards the lower
// id (brute force keeps the lowest index by construction, rt_accel_array.cpp:116).
__device__ __forceinline__ bool mt_closest(f3 O, f3 D, f3 v0, f3 e1, f3 e2, uint32_t prim, const TraceParams& tp,
                                           Hit& best)
{
	f3 P = cross3(D, e2);
	float det = dot3(P, e1);
	if (det < tp.eps_det) return false;
	f3 T = O - v0;
	float u = dot3(T, P);
	if (u < 0.0f || u > det) return false;
	f3 Q = cross3(T, e1);
	float v = dot3(D, Q);
	if (v < 0.0f || u + v > det) return false;
	float t = dot3(e2, Q);
	float inv_det = 1.0f / det;  // correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt)
	t *= inv_det;
	u *= inv_det;
	v *= inv_det;
	if (t < tp.eps_t) return false;
	// the tie rule applies between two real hits only: while `best` is still the initial {t_max, RT_MISS}, a hit at
	// exactly t == t_max is rejected as the reference rejects it (t < min_t, rt_accel_array.cpp:116)
	if (t < best.t || (t == best.t && best.prim != RT_MISS && prim < best.prim)) {
		best.t = t; best.u = u; best.v = v; best.prim = prim;
		return true;
	}
	return false;
}

// Any-hit with intersect's acceptance tests (eps_det, eps_t, t < t_max).
__device__ __forceinline__ bool mt_any(f3 O, f3 D, f3 v0, f3 e1, f3 e2, const TraceParams& tp)
{
	f3 P = cross3(D, e2);
	float det = dot3(P, e1);
	if (det < tp.eps_det) return false;
	f3 T = O - v0;
	float u = dot3(T, P);
	if (u < 0.0f || u > det) return false;
	f3 Q = cross3(T, e1);
	float v = dot3(D, Q);
	if (v < 0.0f || u + v > det) return false;
	float t = dot3(e2, Q);
	float inv_det = 1.0f / det;
	t *= inv_det;
	if (t < tp.eps_t) return false;
	return t < tp.t_max;
}

// AccelArray::occlude (rt_accel_array.cpp:45-73): t is computed and never tested.
__device__ __forceinline__ bool mt_occlude_ref(f3 O, f3 D, f3 v0, f3 e1, f3 e2, float eps_det)
{
	f3 P = cross3(D, e2);
	float det = dot3(P, e1);
	if (det < eps_det) return false;
	f3 T = O - v0;
	f3 Q = cross3(T, e1);
	float inv = 1.0f / det;
	float uu = dot3(P, T) * inv;
	float vv = dot3(Q, D) * inv;
	return uu >= 0.0f && vv >= 0.0f && uu + vv <= 1.0f && 1.0f >= uu;
}

// ---- conservative ray/box (slab) test ------------------------------------------------------
// Not the reference's Woo test (rt_raytracer.cpp:194-248, which ignores ray.t and only culls):
// a slab test on boxes that were padded at build time and rounded outward onto a 16-bit grid.  The
// roundings of the test itself are bounded per ray and folded into its additive constants (ray_pre),
// so it never rejects a box whose triangle the exact M-T test would accept; there is no
// multiplicative widening of the interval.
typedef float v2f __attribute__((ext_vector_type(2)));

// Ray constants for the slab test against quantised planes: t = q * a + b with a = scale / d and
// b = (org - o) / d, one packed-FP32 FMA per (entry, exit) plane pair, 6 per node visit.  |d| < 1e-20 is replaced by
// +-1e-20 before inversion (no inf, no 0 * inf NaN).  fma(q, a, b) is monotone in q, so for a > 0 the min plane's t is
// the smaller one (and the larger for a < 0): choosing the entry plane by the sign of d gives exactly min / max of the
// two, without computing them.
struct RayPre {
	float ax, ay, az;
	v2f bx, by, bz;        // (B_entry, B_exit), see ray_pre
	uint32_t ex, ey, ez;   // v_perm selectors: the entry plane's 16 bits under the exponent of 2^23
	uint32_t xx, xy, xz;   // ... the exit plane's
};

__device__ __forceinline__ float safe_rcp(float d)
{
	float s = (fabsf(d) < 1e-20f) ? copysignf(1e-20f, d) : d;
	return 1.0f / s;
}

__device__ __forceinline__ v2f mk2(float a, float b) { v2f r; r.x = a; r.y = b; return r; }

// The 16-bit plane index q never goes through an integer-to-float conversion: v_perm_b32 drops its two bytes under the
// bit pattern of 2^23 (0x4B000000), which reads as the float m = 2^23 + q exactly, and t = fma(m, a, B) with
// B = b - 2^23 * a.  One v_perm per plane (12 per node visit) instead of a rotate per pair and a convert per plane
// (6 + 12).  The price is the rounding of B, whose magnitude is that of 2^23 * a: up to one grid step |a| in t, so the
// entry planes are taken one step earlier and the exit planes one step later (B -+ |a|), i.e. every box one grid cell
// (extent / 65535) larger on each side than stored -- the test only culls, so results cannot change, and a cell is
// ~1 % of a leaf box of the bench scenes.  delta (the rounding bound of ray_pre's b, see there) enters twice, once for b
// and once for B's own rounding when |b| dominates 2^23 |a| (origins more than ~128 scene extents away).
// Which half of the stored (min | max << 16) pair is the entry plane depends on the sign of the direction: that choice
// is in the selectors.
__device__ __forceinline__ RayPre ray_pre(const QFrame& qf, f3 o, f3 d)
{
	RayPre r;
	const float ix = safe_rcp(d.x), iy = safe_rcp(d.y), iz = safe_rcp(d.z);
	r.ax = qf.scale[0] * ix; r.ay = qf.scale[1] * iy; r.az = qf.scale[2] * iz;
	const float bx = (qf.org[0] - o.x) * ix, by = (qf.org[1] - o.y) * iy, bz = (qf.org[2] - o.z) * iz;
	// delta = 2^-21 x the largest magnitude that enters t = q * a + b on this axis (|b|, and the plane and
	// origin coordinates over |d|): a few ulps of every rounding between the float planes the build
	// compared against and this t, whatever the distance of the ray's origin or of the scene from 0.
	const float eps = 4.76837158203125e-07f;
	const float ex = __builtin_fmaf(fabsf(ix), qf.reach[0] + fabsf(o.x), fabsf(bx)) * eps;
	const float ey = __builtin_fmaf(fabsf(iy), qf.reach[1] + fabsf(o.y), fabsf(by)) * eps;
	const float ez = __builtin_fmaf(fabsf(iz), qf.reach[2] + fabsf(o.z), fabsf(bz)) * eps;
	const float two23 = 8388608.0f;
	r.bx = mk2(__builtin_fmaf(-two23, r.ax, bx - 2.0f * ex) - fabsf(r.ax), __builtin_fmaf(-two23, r.ax, bx + 2.0f * ex) + fabsf(r.ax));
	r.by = mk2(__builtin_fmaf(-two23, r.ay, by - 2.0f * ey) - fabsf(r.ay), __builtin_fmaf(-two23, r.ay, by + 2.0f * ey) + fabsf(r.ay));
	r.bz = mk2(__builtin_fmaf(-two23, r.az, bz - 2.0f * ez) - fabsf(r.az), __builtin_fmaf(-two23, r.az, bz + 2.0f * ez) + fabsf(r.az));
	// selector bytes: 0-3 = bytes of the plane pair, 0x0c = 0x00, 7 = the top byte of 0x4B000000
	const uint32_t lo = 0x070c0100u, hi = 0x070c0302u;
	const bool nx = (__float_as_uint(ix) >> 31) != 0, ny = (__float_as_uint(iy) >> 31) != 0, nz = (__float_as_uint(iz) >> 31) != 0;
	r.ex = nx ? hi : lo; r.xx = nx ? lo : hi;
	r.ey = ny ? hi : lo; r.xy = ny ? lo : hi;
	r.ez = nz ? hi : lo; r.xz = nz ? lo : hi;
	return r;
}

__device__ __forceinline__ v2f fma2(v2f q, v2f a, v2f b) { return __builtin_elementwise_fma(q, a, b); }
// (t at the entry plane, t at the exit plane) of one axis; w = min | max << 16
__device__ __forceinline__ v2f slab_axis(uint32_t w, uint32_t sel_entry, uint32_t sel_exit, float a, v2f b)
{
	const uint32_t k = 0x4B000000u;
	const v2f m = mk2(__uint_as_float(__builtin_amdgcn_perm(k, w, sel_entry)), __uint_as_float(__builtin_amdgcn_perm(k, w, sel_exit)));
	return fma2(m, mk2(a, a), b);
}

// Both child boxes of a node against one ray.  Hit iff the slab interval intersects [0, tlimit]; entry
// distances in tl / tr.
__device__ __forceinline__ void slab2(const RayPre& r, uint4 w0, uint4 w1, float tlimit, bool& hl, bool& hr,
                                      float& tl, float& tr)
{
	const v2f lx = slab_axis(w0.x, r.ex, r.xx, r.ax, r.bx), ly = slab_axis(w0.y, r.ey, r.xy, r.ay, r.by), lz = slab_axis(w0.z, r.ez, r.xz, r.az, r.bz);
	const v2f qx = slab_axis(w0.w, r.ex, r.xx, r.ax, r.bx), qy = slab_axis(w1.x, r.ey, r.xy, r.ay, r.by), qz = slab_axis(w1.y, r.ez, r.xz, r.az, r.bz);
	const float tnl = fmaxf(fmaxf(lx.x, ly.x), fmaxf(lz.x, 0.0f));
	const float tfl = fminf(fminf(lx.y, ly.y), lz.y);
	const float tnr = fmaxf(fmaxf(qx.x, qy.x), fmaxf(qz.x, 0.0f));
	const float tfr = fminf(fminf(qx.y, qy.y), qz.y);
	// No multiplicative widening of [tn, tf] here: the roundings of t are covered by what ray_pre folded into B,
	// on top of the build-time pad of 1e-5 x extent and the outward rounding of the grid.
	tl = tnl; tr = tnr;
	hl = tnl <= fminf(tfl, tlimit);
	hr = tnr <= fminf(tfr, tlimit);
}

// ---- wave helpers ---------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ uint64_t lanemask_lt()
{
	return (1ull << lane_id()) - 1ull;
}
