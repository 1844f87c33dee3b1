// oracle/rt_oracle.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Plain-C++ CPU restatement of the reference's ray-tracing hot path (Kupoman/raytracer), written
// from the reference's behaviour, NOT a copy of its code: no Eigen, flat arrays, a C ABI.  It is
// the checker for the HIP product path; only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load it.  The product (raytracer_amd/csrc) never links or calls it.
//
// Pinning: every function below marked [pinned] is compared bit for bit (ids, t, u, v) or to the
// stated tolerance against the compiled reference (oracle/_ref/libref_oracle.so, built from the
// reference sources by oracle/Makefile) in tests/test_oracle_vs_ref.py, and against the golden
// vectors under tests/golden/ (generated from the compiled reference by oracle/gen_golden.py)
// in tests/test_oracle_golden.py.  Functions marked [unpinned] restate code the reference
// cannot execute here (dead `#if 0` shade(), GLSL shaders, GL texture formats): parity for those
// is "unpinned" and says so in DESIGN.md.
//
// Arithmetic notes that matter for bit-exactness against Eigen 3.0.6 as compiled by g++ -O2 on
// x86-64 (no FMA contraction):
//   * a.dot(b) on Vector3f reduces as a0*b0 + (a1*b1 + a2*b2)   (externs/Eigen/src/Core/Redux.h:98-101)
//   * v / s multiplies by (1.0f / s)                            (externs/Eigen/src/Core/Functors.h:455-456)
//   * cross() is the textbook component formula                 (externs/Eigen/src/Geometry/OrthoMethods.h:49-51)
//   * Matrix4f * Vector4f accumulates column by column: ((c0*x + c1*y) + c2*z) + c3*w
#include "rt_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <vector>
#include <math.h>

namespace {

struct V3 {
	float x, y, z;
	float& operator[](int i) { return (&x)[i]; }
	const float& operator[](int i) const { return (&x)[i]; }
};
struct V2 { float x, y; };

inline V3 mk(float x, float y, float z) { V3 r = {x, y, z}; return r; }
inline V3 ld3(const float* p) { return mk(p[0], p[1], p[2]); }
inline void st3(float* p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
inline V3 add(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 sub(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 mul(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
inline V3 neg(V3 a) { return mk(-a.x, -a.y, -a.z); }
inline float dot(V3 a, V3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
inline V3 cross(V3 a, V3 b)
{
	return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline float norm(V3 a) { return sqrtf(dot(a, a)); }
inline V3 divs(V3 a, float s) { float inv = 1.0f / s; return mul(a, inv); }  // Eigen: v / s
inline V3 normalized(V3 a) { return divs(a, norm(a)); }
// (1-u-v)*a + u*b + v*c, Eigen left-to-right
inline V3 bary(float u, float v, V3 a, V3 b, V3 c)
{
	float w = (1 - u - v);
	return add(add(mul(a, w), mul(b, u)), mul(c, v));
}

struct Material {
	V3 diffuse, specular;
	float shininess, reflectivity, alpha, ior;
	int texture;  // index into orc_scene::textures, -1 = NULL (data/data.h:25)
};
// data/data.h:7-16: RGBA8, rows in FreeImage_GetBits order (row 0 = the bottom scanline), which is
// also the order glTexImage2D receives (ras_rasterizer.cpp:132): texel (x, y) = px[4 * (y*w + x)].
struct Texture { int w, h; std::vector<uint8_t> px; };
struct Light { V3 position, color; float energy; };
struct Mesh {
	std::vector<V3> verts, normals;
	std::vector<V2> texcoords;
	std::vector<uint32_t> faces;  // 3 per face
	float model[16];              // column-major
	int material;
	V3 position;
};
// rt_raytracer.h:10-24
struct Tri {
	V3 v0, v1, v2, n0, n1, n2;
	V2 t0, t1, t2;
	V3 e1, e2;
	int material;
	V3 centroid;
};
struct DACTri { V3 mn, mx; unsigned id; };
struct DACRay { V3 o, d; float t; unsigned id; };
struct DACRes { float u, v; unsigned tidx; V3 o, d; float t; };
struct Sphere { V3 position; float radius; V3 diffuse; };
struct Photon { V3 position, energy, direction; float distance2; };

// Matrix4f * (v,1) -> xyz, column accumulation order.
inline V3 xform(const float* m, V3 v)
{
	V3 r;
	for (int k = 0; k < 3; ++k) {
		float acc = m[0 * 4 + k] * v.x;
		acc = m[1 * 4 + k] * v.y + acc;
		acc = m[2 * 4 + k] * v.z + acc;
		acc = m[3 * 4 + k] * 1.0f + acc;
		r[k] = acc;
	}
	return r;
}

// inverse-transpose of a 4x4 (column-major in, column-major out), computed in double with
// cofactors.  The reference uses Eigen's SSE 4x4 inverse (rt_raytracer.cpp:468); results agree to
// ~1e-7 relative and exactly for identity / pure translations.  Affects hit-record normals only.
void inverse_transpose(const float* m, float* out)
{
	double a[4][4], inv[4][4];
	for (int c = 0; c < 4; ++c)
		for (int r = 0; r < 4; ++r) a[r][c] = m[c * 4 + r];
	double det = 0;
	double cof[4][4];
	for (int r = 0; r < 4; ++r)
		for (int c = 0; c < 4; ++c) {
			double sm[3][3];
			int rr = 0;
			for (int i = 0; i < 4; ++i) {
				if (i == r) continue;
				int cc = 0;
				for (int j = 0; j < 4; ++j) {
					if (j == c) continue;
					sm[rr][cc++] = a[i][j];
				}
				++rr;
			}
			double d3 = sm[0][0] * (sm[1][1] * sm[2][2] - sm[1][2] * sm[2][1]) -
			            sm[0][1] * (sm[1][0] * sm[2][2] - sm[1][2] * sm[2][0]) +
			            sm[0][2] * (sm[1][0] * sm[2][1] - sm[1][1] * sm[2][0]);
			cof[r][c] = (((r + c) & 1) ? -d3 : d3);
		}
	for (int c = 0; c < 4; ++c) det += a[0][c] * cof[0][c];
	for (int r = 0; r < 4; ++r)
		for (int c = 0; c < 4; ++c) inv[r][c] = cof[c][r] / det;  // inverse = adj / det
	// transpose of inverse, stored column-major: out[c*4+r] = invT[r][c] = inv[c][r]
	for (int c = 0; c < 4; ++c)
		for (int r = 0; r < 4; ++r) out[c * 4 + r] = (float)inv[c][r];
}

const float EPS_F = std::numeric_limits<float>::epsilon();  // rt_raytracer.cpp:18

// Texture::lookup, data/data.cpp:35-57 [unpinned: FreeImage absent].  Returns 0..255 per channel.
// FreeImage_GetPixelColor fails for x == width / y == height (u or v exactly 1.0) and the reference
// then returns an uninitialised RGBQUAD; the restatement clamps to the last texel instead.
V3 texture_lookup(const Texture& t, float u, float v)
{
	if (fabs(u) > 1) u = fmodf(u, 1.0f);
	if (fabs(v) > 1) v = fmodf(v, 1.0f);
	if (u < 0.0) u = (float)(u + 1.0);
	if (v < 0.0) v = (float)(v + 1.0);
	int x = (int)(u * (float)(unsigned)t.w);
	int y = (int)(v * (float)(unsigned)t.h);
	x = std::min(std::max(x, 0), t.w - 1);
	y = std::min(std::max(y, 0), t.h - 1);
	const uint8_t* c = &t.px[4 * ((size_t)y * t.w + x)];
	return mk(c[0], c[1], c[2]);
}

// texture2D on a GL_NEAREST texture with the default GL_REPEAT wrap (ras_rasterizer.cpp:130-135,
// mesh_fs.cpp:29) [unpinned: GL]: i = floor(frac(s) * w) clamped to w-1, UNORM8 -> c / 255.
V3 texture_gl_nearest(const Texture& t, float u, float v)
{
	float fu = u - floorf(u), fv = v - floorf(v);
	int x = std::min(std::max((int)floorf(fu * (float)t.w), 0), t.w - 1);
	int y = std::min(std::max((int)floorf(fv * (float)t.h), 0), t.h - 1);
	const uint8_t* c = &t.px[4 * ((size_t)y * t.w + x)];
	return mk(c[0] / 255.0f, c[1] / 255.0f, c[2] / 255.0f);
}

// "x < 0.00001" with the literal a double (rt_accel_array.cpp:10,94,114): exact in double.
inline bool lt_eps_array(float x) { return (double)x < 0.00001; }

}  // namespace

struct orc_scene {
	std::vector<Material> materials;
	std::vector<Texture> textures;
	std::vector<Mesh> meshes;
	std::vector<Light> lights;
	std::vector<Tri> tris;        // world-space soup (processRays)
	std::vector<DACTri> dtris;
	V3 bmin, bmax;
	bool soup_valid;
	// AccelArray SoA (object space)
	std::vector<V3> a_v0, a_v1, a_v2, a_n0, a_n1, a_n2, a_e1, a_e2;
	std::vector<V2> a_t0, a_t1, a_t2;
	std::vector<int> a_mat;
	bool array_valid;
	std::vector<Sphere> spheres;
	bool spheres_valid;
	std::vector<Photon> photons;
	orc_scene() : soup_valid(false), array_valid(false), spheres_valid(false) {}
};

namespace {

// ---- a8: soup rebuild, rt_raytracer.cpp:457-503 ------------------------------------------------
void build_soup(orc_scene* s)
{
	s->tris.clear();
	s->dtris.clear();
	if (s->meshes.empty() || s->meshes[0].verts.empty()) { s->soup_valid = true; return; }
	s->bmin = s->bmax = s->meshes[0].verts[0];  // untransformed seed, as the reference (:464)
	for (size_t i = 0; i < s->meshes.size(); ++i) {
		const Mesh& m = s->meshes[i];
		float nm[16];
		inverse_transpose(m.model, nm);
		size_t nf = m.faces.size() / 3;
		for (size_t j = 0; j < nf; ++j) {
			uint32_t i0 = m.faces[3 * j], i1 = m.faces[3 * j + 1], i2 = m.faces[3 * j + 2];
			Tri t;
			t.v0 = xform(m.model, m.verts[i0]);
			t.v1 = xform(m.model, m.verts[i1]);
			t.v2 = xform(m.model, m.verts[i2]);
			t.n0 = xform(nm, m.normals[i0]);
			t.n1 = xform(nm, m.normals[i1]);
			t.n2 = xform(nm, m.normals[i2]);
			t.t0 = m.texcoords[i0];
			t.t1 = m.texcoords[i1];
			t.t2 = m.texcoords[i2];
			t.e1 = sub(t.v1, t.v0);
			t.e2 = sub(t.v2, t.v0);
			t.material = m.material;
			t.centroid = divs(add(add(t.v0, t.v1), t.v2), 3.0f);
			DACTri d;
			d.id = (unsigned)s->tris.size();
			for (int k = 0; k < 3; ++k) {
				d.mn[k] = std::min(std::min(t.v0[k], t.v1[k]), t.v2[k]);
				d.mx[k] = std::max(std::max(t.v0[k], t.v1[k]), t.v2[k]);
				s->bmin[k] = std::min(s->bmin[k], d.mn[k]);
				s->bmax[k] = std::max(s->bmax[k], d.mx[k]);
			}
			s->tris.push_back(t);
			s->dtris.push_back(d);
		}
	}
	s->soup_valid = true;
}

// ---- a10: Woo et al. ray/box, rt_raytracer.cpp:194-248 -----------------------------------------
bool ray_aabb(const V3& o, const V3& d, const V3& mn, const V3& mx)
{
	int quadrant[3];  // 0 right, 1 left, 2 middle
	float cand[3] = {0, 0, 0}, maxt[3];
	bool inside = true;
	for (int i = 0; i < 3; ++i) {
		if (o[i] < mn[i]) { quadrant[i] = 1; cand[i] = mn[i]; inside = false; }
		else if (o[i] > mx[i]) { quadrant[i] = 0; cand[i] = mx[i]; inside = false; }
		else quadrant[i] = 2;
	}
	if (inside) return true;
	for (int i = 0; i < 3; ++i) {
		// `ray.direction[i] != 0.0` compares against a double zero: same truth value in float
		if (quadrant[i] != 2 && d[i] != 0.0f) maxt[i] = (cand[i] - o[i]) / d[i];
		else maxt[i] = -1.0f;
	}
	int plane = 0;
	for (int i = 1; i < 3; ++i)
		if (maxt[plane] < maxt[i]) plane = i;
	if (maxt[plane] < 0.0f) return false;
	for (int i = 0; i < 3; ++i) {
		if (plane != i) {
			float coord = o[i] + maxt[plane] * d[i];
			if (coord < mn[i] || coord > mx[i]) return false;
		}
	}
	return true;
}

struct Dacrt {
	orc_scene* s;
	std::vector<DACTri> tris;  // permuted copy
	std::vector<DACRay> rays;
	std::vector<DACRes> res;

	struct Cmp {
		const std::vector<Tri>* lut;
		int axis;
		bool operator()(const DACTri& a, const DACTri& b) const
		{
			return (*lut)[a.id].centroid[axis] > (*lut)[b.id].centroid[axis];
		}
	};

	// a11: rt_raytracer.cpp:315-370 (triangle outer, ray inner)
	void naive(int rs, int re, int ts, int te)
	{
		for (int ti = ts; ti < te; ++ti) {
			const Tri& tri = s->tris[tris[ti].id];
			for (int ri = rs; ri < re; ++ri) {
				V3 D = rays[ri].d, O = rays[ri].o;
				V3 P = cross(D, tri.e2);
				float det = dot(P, tri.e1);
				if (det < EPS_F) continue;
				V3 T = sub(O, tri.v0);
				float u = dot(T, P);
				if (u < 0 || u > det) continue;
				V3 Q = cross(T, tri.e1);
				float v = dot(D, Q);
				if (v < 0 || u + v > det) continue;
				float t = dot(tri.e2, Q);
				float inv_det = (float)(1.0 / det);
				t *= inv_det; u *= inv_det; v *= inv_det;
				if (t < EPS_F) continue;
				if (t < rays[ri].t) {
					unsigned id = rays[ri].id;
					rays[ri].t = t;
					res[id].u = u; res[id].v = v; res[id].tidx = tris[ti].id;
					res[id].o = O; res[id].d = D; res[id].t = t;
				}
			}
		}
	}

	// a9: rt_raytracer.cpp:250-313
	void trace(int rs, int re, int ts, int te, V3 bmn, V3 bmx)
	{
		if (re - rs < 8 || te - ts < 8) { naive(rs, re, ts, te); return; }
		V3 dims = sub(bmx, bmn);
		float max_dim = std::max(std::max(dims.x, dims.y), dims.z);
		int axis = (dims.x == max_dim) ? 0 : (dims.y == max_dim) ? 1 : 2;
		int tmid = (ts + te) / 2;
		Cmp cmp = {&s->tris, axis};
		std::nth_element(tris.begin() + ts, tris.begin() + tmid, tris.begin() + te, cmp);
		int tidx[3] = {ts, tmid, te};
		for (int space = 0; space < 2; ++space) {
			V3 mn = tris[tidx[space]].mn, mx = tris[tidx[space]].mx;
			for (int t = tidx[space]; t < tidx[space + 1]; ++t) {
				for (int k = 0; k < 3; ++k) {
					mn[k] = std::min(mn[k], tris[t].mn[k]);
					mx[k] = std::max(mx[k], tris[t].mx[k]);
				}
			}
			int pivot = re;
			for (int i = rs; i < pivot; ++i) {
				if (!ray_aabb(rays[i].o, rays[i].d, mn, mx)) {
					std::swap(rays[pivot - 1], rays[i]);
					--pivot;
					--i;
				}
			}
			trace(rs, pivot, tidx[space], tidx[space + 1], mn, mx);
		}
	}
};

void ensure_array(orc_scene* s)
{
	if (s->array_valid) return;
	// a13: rt_accel_array.cpp:17-32 (object space, no model_mat)
	for (size_t i = 0; i < s->meshes.size(); ++i) {
		const Mesh& m = s->meshes[i];
		size_t nf = m.faces.size() / 3;
		for (size_t j = 0; j < nf; ++j) {
			uint32_t i0 = m.faces[3 * j], i1 = m.faces[3 * j + 1], i2 = m.faces[3 * j + 2];
			s->a_v0.push_back(m.verts[i0]); s->a_v1.push_back(m.verts[i1]); s->a_v2.push_back(m.verts[i2]);
			s->a_n0.push_back(m.normals[i0]); s->a_n1.push_back(m.normals[i1]); s->a_n2.push_back(m.normals[i2]);
			s->a_t0.push_back(m.texcoords[i0]); s->a_t1.push_back(m.texcoords[i1]); s->a_t2.push_back(m.texcoords[i2]);
			s->a_e1.push_back(sub(m.verts[i1], m.verts[i0]));
			s->a_e2.push_back(sub(m.verts[i2], m.verts[i0]));
			s->a_mat.push_back(m.material);
		}
	}
	s->array_valid = true;
}

// a14: rt_accel_array.cpp:75-137.  Returns triangle index or -1; fills the record on a hit.
int array_intersect(orc_scene* s, orc_ray* ray, int* material, float* out_t)
{
	V3 D = ld3(ray->direction), O = ld3(ray->origin);
	float min_t = 1000000, min_u = 0, min_v = 0;
	int min_index = -1;
	size_t n = s->a_v0.size();
	for (size_t i = 0; i < n; ++i) {
		V3 E1 = s->a_e1[i], E2 = s->a_e2[i];
		V3 P = cross(D, E2);
		float det = dot(P, E1);
		if (lt_eps_array(det)) continue;
		V3 T = sub(O, s->a_v0[i]);
		float u = dot(T, P);
		if (u < 0 || u > det) continue;
		V3 Q = cross(T, E1);
		float v = dot(D, Q);
		if (v < 0 || u + v > det) continue;
		float t = dot(E2, Q);
		float inv_det = (float)(1.0 / det);
		t *= inv_det; u *= inv_det; v *= inv_det;
		if (lt_eps_array(t)) continue;
		if (t < min_t) { min_t = t; min_u = u; min_v = v; min_index = (int)i; }
	}
	if (min_index != -1) {
		size_t k = (size_t)min_index;
		st3(ray->position, bary(min_u, min_v, s->a_v0[k], s->a_v1[k], s->a_v2[k]));
		V3 nn = bary(min_u, min_v, s->a_n0[k], s->a_n1[k], s->a_n2[k]);
		st3(ray->normal, normalized(nn));
		float w = (1 - min_u - min_v);
		ray->texcoord[0] = (w * s->a_t0[k].x + min_u * s->a_t1[k].x) + min_v * s->a_t2[k].x;
		ray->texcoord[1] = (w * s->a_t0[k].y + min_u * s->a_t1[k].y) + min_v * s->a_t2[k].y;
		if (material) *material = s->a_mat[k];
		if (out_t) *out_t = min_t;
	}
	return min_index;
}

// a15: rt_accel_array.cpp:45-73 (t is computed and never tested)
bool array_occlude(orc_scene* s, V3 O, V3 D)
{
	size_t n = s->a_v0.size();
	for (size_t i = 0; i < n; ++i) {
		V3 E1 = s->a_e1[i], E2 = s->a_e2[i];
		V3 P = cross(D, E2);
		float det = dot(P, E1);
		if (lt_eps_array(det)) continue;
		V3 T = sub(O, s->a_v0[i]);
		V3 Q = cross(T, E1);
		V3 tuv = divs(mk(dot(Q, E2), dot(P, T), dot(Q, D)), det);
		if (tuv.y >= 0 && tuv.z >= 0 && tuv.y + tuv.z <= 1 && 1 >= tuv.y) return true;
	}
	return false;
}

void ensure_spheres(orc_scene* s)
{
	if (s->spheres_valid) return;
	srand(420);  // rt_accel_spheres.cpp:11
	for (size_t i = 0; i < s->meshes.size(); ++i) {
		Sphere sp;
		sp.position = s->meshes[i].position;
		sp.radius = 1;
		// rt_accel_spheres.cpp:24: Vector3f(rand()%256, rand()%256, rand()%256).  The order in which
		// the three arguments are evaluated is unspecified in C++; g++ on x86-64 evaluates them
		// right to left, so the FIRST draw lands in z.  Pinned against the compiled reference.
		float b = (float)(rand() % 256), g = (float)(rand() % 256), r = (float)(rand() % 256);
		sp.diffuse = mk(r, g, b);
		s->spheres.push_back(sp);
	}
	s->spheres_valid = true;
}

// a17: rt_accel_spheres.cpp:58-104.  Ro = direction (reference quirk), first passing sphere wins.
int spheres_intersect(orc_scene* s, orc_ray* ray)
{
	float min_distance = 1000000;
	for (size_t i = 0; i < s->spheres.size(); ++i) {
		const Sphere& sp = s->spheres[i];
		V3 Rd = ld3(ray->direction);
		V3 Ro = ld3(ray->direction);
		float a = 1;
		V3 oc = sub(Ro, sp.position);
		float b = dot(mul(Rd, 2.0f), oc);
		float c = dot(oc, oc) - sp.radius * sp.radius;
		float disc = (b * b - 4 * a * c);
		if (disc < 0) continue;
		disc = sqrtf(disc);
		float t0 = (float)((-b - disc) * 0.5);
		float t1 = (float)((-b + disc) * 0.5);
		if (t0 < 0 && t1 > 0) continue;
		t0 = std::min(t0, t1);
		if (t0 <= 0.0) continue;
		V3 X = add(Ro, mul(Rd, t0));
		float distance = dot(X, X);
		if (distance < min_distance) {
			st3(ray->position, X);
			st3(ray->normal, divs(sub(X, sp.position), sp.radius));
			return (int)i;
		}
	}
	return -1;
}

// a18: rt_accel_spheres.cpp:38-56
bool spheres_occlude(orc_scene* s, V3 Ro, V3 Rd)
{
	for (size_t i = 0; i < s->spheres.size(); ++i) {
		const Sphere& sp = s->spheres[i];
		float a = 1;
		V3 oc = sub(Ro, sp.position);
		float b = dot(mul(Rd, 2.0f), oc);
		float c = dot(oc, oc) - sp.radius * sp.radius;
		if ((b * b - 4 * a * c) > 0) return true;
	}
	return false;
}

// IAccel::intersect dispatch used by shade() and the photon map.
// Returns true on a hit; *mat_out = material index (triangles) or -(sphere index)-2 (spheres).
struct HitMat { int tri_material; int sphere; };
bool accel_intersect(orc_scene* s, int geometry, orc_ray* r, HitMat* hm)
{
	if (geometry == 0) {
		int m = -1;
		int idx = array_intersect(s, r, &m, NULL);
		if (idx < 0) return false;
		hm->tri_material = m; hm->sphere = -1;
		return true;
	}
	int si = spheres_intersect(s, r);
	if (si < 0) return false;
	hm->tri_material = -1; hm->sphere = si;
	return true;
}

Material material_of(orc_scene* s, const HitMat& hm)
{
	if (hm.sphere >= 0) {
		// `new Material` with only diffuse_color assigned (rt_accel_spheres.cpp:22-24): the other
		// fields are indeterminate in the reference; the restatement defines them as zero.
		Material m;
		memset(&m, 0, sizeof(m));
		m.diffuse = s->spheres[hm.sphere].diffuse;
		m.texture = -1;
		return m;
	}
	return s->materials[hm.tri_material];
}

// ---- a22-a25 photon map ------------------------------------------------------------------------
const float PM_K = 1.0;
const float PM_ALPHA = 0.918;
const float PM_BETA = 1.953;
const float PM_PI = 3.141592653589793;
const float PM_E = 2.718281828459045;
const float PM_GAUSS_DENOM = 1 / (1 - pow(PM_E, -PM_BETA));  // rt_photon_map.cpp:18

inline uint32_t lowbias32(uint32_t x)
{
	x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
	return x;
}
// Counter RNG shared with the HIP photon kernel (raytracer_amd/csrc/rt_rng.h): 31-bit draws, the
// range of libc rand(), keyed by (seed, photon index, draw counter).
struct Rng {
	int mode;  // 0 libc, 1 counter
	uint32_t seed_lo, seed_hi, index, counter;
	int next()
	{
		if (mode == 0) return rand();
		uint32_t h = lowbias32(index ^ seed_lo);
		h = lowbias32(h + 0x9E3779B9U * (counter + 1u) + seed_hi);
		h = lowbias32(h ^ (counter * 0x85EBCA6BU));
		++counter;
		return (int)(h >> 1);
	}
};

// a23: rt_photon_map.cpp:32-69.  `pass++` passes the old value: every level sees pass == 0.
void emit_photon(orc_scene* s, int geometry, Rng& rng, orc_ray* r, V3 energy, float max_dist,
                 int depth_guard)
{
	const float rmax_2 = RAND_MAX / 2.0;
	for (;;) {
		HitMat hm;
		if (!accel_intersect(s, geometry, r, &hm)) return;
		float dist = norm(sub(ld3(r->position), ld3(r->origin)));
		if (dist > max_dist) return;
		float absorb = 0.6;
		if (((float)rng.next()) / RAND_MAX < absorb) {
			Photon p;
			p.position = ld3(r->position);
			p.energy = energy;
			p.direction = ld3(r->direction);
			p.distance2 = 0;
			s->photons.push_back(p);
			return;
		}
		st3(r->origin, ld3(r->position));
		V3 dir = mk(1, 1, 1);
		V3 nrm = ld3(r->normal);
		while (dot(dir, dir) > 1 || dot(dir, nrm) < 0) {
			dir.x = rng.next() / rmax_2 - 1;
			dir.y = rng.next() / rmax_2 - 1;
			dir.z = rng.next() / rmax_2 - 1;
		}
		st3(r->direction, dir);
		Material m = material_of(s, hm);
		energy = mul(m.diffuse, (norm(energy) / 255));
		max_dist = max_dist - dist;
		if (--depth_guard <= 0) return;  // never reached in practice (roulette 0.6 per level)
	}
}

bool photon_less(const Photon* a, const Photon* b) { return a->distance2 < b->distance2; }

// a24: rt_photon_map.cpp:149-208
V3 radiance_estimate(orc_scene* s, V3 position, V3 normal, unsigned count, float radius)
{
	float r2 = radius * radius;
	float max_dist2 = 0;
	std::vector<Photon*> nearest;
	V3 result = mk(0, 0, 0);
	for (size_t i = 0; i < s->photons.size(); ++i) {
		Photon* p = &s->photons[i];
		V3 d = sub(p->position, position);
		float l2 = dot(d, d);
		p->distance2 = l2;
		if (l2 < r2) {
			nearest.push_back(p);
			if (l2 > max_dist2) max_dist2 = l2;
		}
	}
	std::sort(nearest.begin(), nearest.end(), photon_less);
	for (unsigned i = 0; i < nearest.size() && i < count; ++i) {
		Photon* p = nearest[i];
		V3 d = sub(p->position, position);
		float l2 = dot(d, d);
		float weight = 1 - pow(PM_E, -PM_BETA * (l2 / (2 * max_dist2)));
		weight *= PM_GAUSS_DENOM;
		weight = PM_ALPHA * (1 - weight);
		float brdf = dot(normal, neg(p->direction));
		if (brdf > 1.0) brdf = 1.0;
		if (brdf < 0.0) brdf = 0.0;
		result = add(result, mul(mul(p->energy, brdf), weight));
	}
	// FILTER_GUASS typo -> the `#else` normalisation (rt_photon_map.cpp:199-205)
	result = divs(result, (2 * PM_PI * max_dist2));
	return result;
}

// ---- a19: Gen-1 shade(), rt_raytracer.cpp:75-160 [unpinned: dead code in the reference] -------
struct ShadeCtx {
	orc_scene* s;
	const orc_shade_config* cfg;
	int64_t rays;
};

// returns true if `color` was written (pass < bounces)
void shade(ShadeCtx& c, const orc_ray* ray, Material material, bool has_hm, HitMat hm_in, V3* color,
           int pass)
{
	(void)has_hm; (void)hm_in;
	if (!(pass < c.cfg->bounces)) return;
	orc_scene* s = c.s;
	float bias = 0.1;
	float lambert = 0.0;
	float shadow = 1;
	V3 V = ld3(ray->position), N = ld3(ray->normal), I = ld3(ray->direction);
	float ref = material.reflectivity;
	float alpha = material.alpha;
	V3 diff_color = material.diffuse;
	if (material.texture >= 0)  // rt_raytracer.cpp:93-95
		diff_color = texture_lookup(s->textures[material.texture], ray->texcoord[0], ray->texcoord[1]);
	const bool last = (pass + 1 >= c.cfg->bounces);  // children would return without writing

	if (!c.cfg->use_photon_map) {
		for (size_t i = 0; i < s->lights.size(); ++i) {
			V3 L = sub(s->lights[i].position, V);
			lambert += dot(N, normalized(L));
			lambert = std::min(std::max(lambert, 0.0f), 1.0f);
			if (c.cfg->do_shadows) {
				orc_ray lr;
				memset(&lr, 0, sizeof(lr));
				st3(lr.origin, add(V, mul(N, bias)));
				st3(lr.direction, L);
				++c.rays;
				bool blocked;
				if (c.cfg->geometry == 0) {
					float t = 0;
					blocked = array_intersect(s, &lr, NULL, &t) >= 0;
					// corrected variant (not the reference): only hits before shadow_t_max * |L|
					if (blocked && c.cfg->shadow_t_max > 0.0f) blocked = t < c.cfg->shadow_t_max;
				} else {
					blocked = spheres_intersect(s, &lr) >= 0;
				}
				if (blocked) {
					// distance = |ray->position - V| = 0 (rt_raytracer.cpp:115): any hit shadows
					if (0.0f < norm(L)) shadow = (float)std::max(shadow - 0.4, 0.0);
				}
			}
			// specular is accumulated by the reference and never used (:123-125,158)
		}
	}

	V3 ref_color = mk(0, 0, 0);
	if (material.reflectivity > 0 && !last) {
		V3 R = sub(I, mul(N, 2 * dot(I, N)));
		orc_ray rr;
		memset(&rr, 0, sizeof(rr));
		st3(rr.origin, V);
		st3(rr.direction, R);
		HitMat hm;
		++c.rays;
		if (accel_intersect(s, c.cfg->geometry, &rr, &hm)) {
			material = material_of(s, hm);  // Q9: overwrites `material` for the refraction test
			shade(c, &rr, material, true, hm, &ref_color, pass + 1);
		}
	}

	V3 refraction_color = mk(0, 0, 0);
	if (material.alpha > 0.1 && !last) {
		float IoR = material.ior;
		float idn = dot(I, N);
		V3 T = divs(sub(I, mul(N, idn)), IoR);
		// sqrt(1 - ((1 - I.N*I.N))/IoR*IoR): (x/IoR)*IoR, evaluated in float then sqrt in double
		float arg = 1 - ((1 - idn * idn)) / IoR * IoR;
		float sq = (float)sqrt((double)arg);
		T = sub(T, mul(N, sq));
		orc_ray tr;
		memset(&tr, 0, sizeof(tr));
		st3(tr.origin, V);
		st3(tr.direction, T);
		HitMat hm;
		++c.rays;
		if (accel_intersect(s, c.cfg->geometry, &tr, &hm)) {
			Material m2 = material_of(s, hm);
			shade(c, &tr, m2, true, hm, &refraction_color, pass + 1);
		}
	}

	V3 light = mk(lambert, lambert, lambert);
	if (c.cfg->use_photon_map)
		light = radiance_estimate(s, V, N, (unsigned)c.cfg->photon_estimate, c.cfg->photon_radius);
	// shadow * light.array() * ((1.0-ref-alpha)*diff + ref*ref_color + alpha*refraction_color).array()
	float kd = (float)(1.0 - ref - alpha);
	V3 mix = add(add(mul(diff_color, kd), mul(ref_color, ref)), mul(refraction_color, alpha));
	color->x = (shadow * light.x) * mix.x;
	color->y = (shadow * light.y) * mix.y;
	color->z = (shadow * light.z) * mix.z;
}

inline float half_round(float f)
{
	// float -> IEEE binary16 (round to nearest even) -> float, as a GL_RGB16F render target stores it
	uint32_t x;
	memcpy(&x, &f, 4);
	uint32_t sign = x & 0x80000000u;
	uint32_t ax = x & 0x7fffffffu;
	float out;
	if (ax >= 0x7f800000u) { out = f; }                       // inf / nan
	else if (ax >= 0x477ff000u) {                             // >= 65520 -> inf
		uint32_t inf = sign | 0x7f800000u; memcpy(&out, &inf, 4);
	} else if (ax < 0x38800000u) {                            // < 2^-14: half subnormal, step 2^-24
		float a; memcpy(&a, &ax, 4);
		float q = nearbyintf(a * 16777216.0f) / 16777216.0f;  // ties-to-even in default mode
		uint32_t qb; memcpy(&qb, &q, 4); qb |= sign; memcpy(&out, &qb, 4);
	} else {
		uint32_t lsb = (ax >> 13) & 1u;
		uint32_t r = ax + 0xfffu + lsb;
		r &= 0xffffe000u;
		r |= sign; memcpy(&out, &r, 4);
	}
	return out;
}

}  // namespace

extern "C" {

orc_scene* orc_scene_create(void) { return new orc_scene(); }
void orc_scene_destroy(orc_scene* s) { delete s; }

int orc_add_material(orc_scene* s, const float* d, const float* sp, float shininess,
                     float reflectivity, float alpha, float ior)
{
	Material m;
	m.diffuse = ld3(d); m.specular = ld3(sp);
	m.shininess = shininess; m.reflectivity = reflectivity; m.alpha = alpha; m.ior = ior;
	m.texture = -1;
	s->materials.push_back(m);
	return (int)s->materials.size() - 1;
}

// Material::texture = new Texture(path) (data/loader.cpp:110-121) with the decoded image handed in:
// RGBA8, row 0 = bottom scanline.  rgba == NULL or width/height <= 0 removes the texture.
int orc_set_texture(orc_scene* s, int material_id, const uint8_t* rgba, int width, int height)
{
	if (material_id < 0 || material_id >= (int)s->materials.size()) return -1;
	if (!rgba || width <= 0 || height <= 0) { s->materials[material_id].texture = -1; return 0; }
	Texture t;
	t.w = width; t.h = height;
	t.px.assign(rgba, rgba + (size_t)4 * width * height);
	s->textures.push_back(t);
	s->materials[material_id].texture = (int)s->textures.size() - 1;
	return 0;
}

int orc_add_mesh(orc_scene* s, const float* verts, const float* normals, const float* texcoords,
                 int nverts, const uint32_t* faces, int nfaces, const float* model_mat16,
                 int material_id, const float* position3)
{
	Mesh m;
	m.verts.resize(nverts); m.normals.resize(nverts); m.texcoords.resize(nverts);
	for (int i = 0; i < nverts; ++i) {
		m.verts[i] = ld3(verts + 3 * i);
		m.normals[i] = ld3(normals + 3 * i);
		m.texcoords[i].x = texcoords[2 * i]; m.texcoords[i].y = texcoords[2 * i + 1];
	}
	m.faces.assign(faces, faces + 3 * (size_t)nfaces);
	memcpy(m.model, model_mat16, sizeof(m.model));
	m.material = material_id;
	m.position = position3 ? ld3(position3) : mk(0, 0, 0);
	s->meshes.push_back(m);
	s->soup_valid = s->array_valid = s->spheres_valid = false;
	s->a_v0.clear(); s->a_v1.clear(); s->a_v2.clear(); s->a_n0.clear(); s->a_n1.clear(); s->a_n2.clear();
	s->a_e1.clear(); s->a_e2.clear(); s->a_t0.clear(); s->a_t1.clear(); s->a_t2.clear(); s->a_mat.clear();
	s->spheres.clear();
	return (int)s->meshes.size() - 1;
}

int orc_add_light(orc_scene* s, const float* p, const float* c, float energy)
{
	Light l;
	l.position = ld3(p); l.color = ld3(c); l.energy = energy;
	s->lights.push_back(l);
	return (int)s->lights.size() - 1;
}

int orc_num_tris(orc_scene* s)
{
	if (!s->soup_valid) build_soup(s);
	return (int)s->tris.size();
}

// [pinned via orc_process_rays: out_dir is compared bit for bit]
void orc_reflect_raygen(int n, const float* pos, const float* nrm, float* out_org, float* out_dir)
{
	for (int i = 0; i < n; ++i) {
		V3 P = ld3(pos + 3 * i), N = ld3(nrm + 3 * i);
		V3 d = sub(P, mul(N, 2 * dot(P, N)));
		st3(out_dir + 3 * i, normalized(d));
		st3(out_org + 3 * i, P);
	}
}

// [pinned indirectly through orc_process_rays; known-answer groups in tests/test_oracle_aabb.py]
int orc_ray_aabb(const float* o, const float* d, const float* mn, const float* mx)
{
	return ray_aabb(ld3(o), ld3(d), ld3(mn), ld3(mx)) ? 1 : 0;
}

void orc_get_soup(orc_scene* s, float* v0e1e2, float* bounds6)
{
	if (!s->soup_valid) build_soup(s);
	for (size_t i = 0; i < s->tris.size(); ++i) {
		st3(v0e1e2 + 9 * i, s->tris[i].v0);
		st3(v0e1e2 + 9 * i + 3, s->tris[i].e1);
		st3(v0e1e2 + 9 * i + 6, s->tris[i].e2);
	}
	if (bounds6) { st3(bounds6, s->bmin); st3(bounds6 + 3, s->bmax); }
}

// [pinned: tidx/t/u/v/dir bit-exact, records exact for identity/translation model matrices]
int orc_process_rays(orc_scene* s, int count, const float* positions, const float* normals,
                     uint32_t* out_tidx, float* out_t, float* out_u, float* out_v, float* out_dir,
                     orc_ray* out_records, int* out_offset_material, int* out_offset_end,
                     int* n_materials)
{
	build_soup(s);  // the reference rebuilds every call
	Dacrt d;
	d.s = s;
	d.tris = s->dtris;
	d.rays.resize(count);
	d.res.resize(count);
	std::vector<float> org(3 * (size_t)count), dir(3 * (size_t)count);
	orc_reflect_raygen(count, positions, normals, org.data(), dir.data());
	for (int i = 0; i < count; ++i) {
		d.rays[i].o = ld3(&org[3 * i]);
		d.rays[i].d = ld3(&dir[3 * i]);
		d.rays[i].t = std::numeric_limits<float>::infinity();
		d.rays[i].id = (unsigned)i;
		memset(&d.res[i], 0, sizeof(DACRes));
		d.res[i].tidx = 0xFFFFFFFFu;
	}
	if (!s->tris.empty()) d.trace(0, count, 0, (int)s->tris.size(), s->bmin, s->bmax);

	std::map<int, std::vector<orc_ray> > groups;
	for (int i = 0; i < count; ++i) {
		out_tidx[i] = d.res[i].tidx;
		out_t[i] = d.res[i].t; out_u[i] = d.res[i].u; out_v[i] = d.res[i].v;
		st3(out_dir + 3 * i, ld3(&dir[3 * i]));
		if (d.res[i].tidx == 0xFFFFFFFFu) continue;
		float u = d.res[i].u, v = d.res[i].v;
		const Tri& t = s->tris[d.res[i].tidx];
		orc_ray r;
		memset(&r, 0, sizeof(r));
		st3(r.origin, d.res[i].o);
		st3(r.direction, d.res[i].d);
		st3(r.position, bary(u, v, t.v0, t.v1, t.v2));
		st3(r.normal, normalized(bary(u, v, t.n0, t.n1, t.n2)));
		float w = (1 - u - v);
		r.texcoord[0] = (w * t.t0.x + u * t.t1.x) + v * t.t2.x;
		r.texcoord[1] = (w * t.t0.y + u * t.t1.y) + v * t.t2.y;
		groups[t.material].push_back(r);
	}
	int offset = 0, nm = 0, nres = 0;
	for (std::map<int, std::vector<orc_ray> >::iterator it = groups.begin(); it != groups.end(); ++it) {
		offset += (int)it->second.size();
		if (out_records)
			memcpy(out_records + nres, it->second.data(), it->second.size() * sizeof(orc_ray));
		nres += (int)it->second.size();
		out_offset_material[nm] = it->first;
		out_offset_end[nm] = offset;
		++nm;
	}
	*n_materials = nm;
	return nres;
}

// [pinned against processRays' per-ray results except exact-t ties, see tests]
void orc_soup_closest(orc_scene* s, int n, const float* org, const float* dir, float eps_det,
                      float eps_t, float t_max, uint32_t* out_tidx, float* out_t, float* out_u,
                      float* out_v)
{
	if (!s->soup_valid) build_soup(s);
	const size_t nt = s->tris.size();
	for (int i = 0; i < n; ++i) {
		V3 O = ld3(org + 3 * i), D = ld3(dir + 3 * i);
		float best = t_max, bu = 0, bv = 0;
		uint32_t bi = 0xFFFFFFFFu;
		for (size_t k = 0; k < nt; ++k) {
			const Tri& tri = s->tris[k];
			V3 P = cross(D, tri.e2);
			float det = dot(P, tri.e1);
			if (det < eps_det) continue;
			V3 T = sub(O, tri.v0);
			float u = dot(T, P);
			if (u < 0 || u > det) continue;
			V3 Q = cross(T, tri.e1);
			float v = dot(D, Q);
			if (v < 0 || u + v > det) continue;
			float t = dot(tri.e2, Q);
			float inv_det = 1.0f / det;
			t *= inv_det; u *= inv_det; v *= inv_det;
			if (t < eps_t) continue;
			if (t < best) { best = t; bu = u; bv = v; bi = (uint32_t)k; }
		}
		out_tidx[i] = bi;
		out_t[i] = (bi == 0xFFFFFFFFu) ? std::numeric_limits<float>::infinity() : best;
		out_u[i] = bu; out_v[i] = bv;
	}
}

// [pinned bit for bit]
void orc_array_intersect(orc_scene* s, int n, const float* rays6, uint8_t* hit, orc_ray* records,
                         int* material, int* out_index, float* out_t)
{
	ensure_array(s);
	for (int i = 0; i < n; ++i) {
		orc_ray r;
		memset(&r, 0, sizeof(r));
		memcpy(r.origin, rays6 + 6 * i, 12);
		memcpy(r.direction, rays6 + 6 * i + 3, 12);
		int m = -1;
		float t = 0;
		int idx = array_intersect(s, &r, &m, &t);
		hit[i] = idx >= 0;
		if (material) material[i] = idx >= 0 ? m : -1;
		if (out_index) out_index[i] = idx;
		if (out_t) out_t[i] = idx >= 0 ? t : std::numeric_limits<float>::infinity();
		if (records) records[i] = r;
	}
}

// [pinned bit for bit]
void orc_array_occlude(orc_scene* s, int n, const float* rays6, uint8_t* occluded)
{
	ensure_array(s);
	for (int i = 0; i < n; ++i)
		occluded[i] = array_occlude(s, ld3(rays6 + 6 * i), ld3(rays6 + 6 * i + 3)) ? 1 : 0;
}

// [pinned bit for bit, including the srand(420) colour sequence]
void orc_spheres_intersect(orc_scene* s, int n, const float* rays6, uint8_t* hit,
                           orc_ray* records, float* diffuse, int* out_index)
{
	ensure_spheres(s);
	for (int i = 0; i < n; ++i) {
		orc_ray r;
		memset(&r, 0, sizeof(r));
		memcpy(r.origin, rays6 + 6 * i, 12);
		memcpy(r.direction, rays6 + 6 * i + 3, 12);
		int si = spheres_intersect(s, &r);
		hit[i] = si >= 0;
		if (diffuse)
			for (int k = 0; k < 3; ++k) diffuse[3 * i + k] = si >= 0 ? s->spheres[si].diffuse[k] : 0.0f;
		if (out_index) out_index[i] = si;
		if (records) records[i] = r;
	}
}

void orc_spheres_occlude(orc_scene* s, int n, const float* rays6, uint8_t* occluded)
{
	ensure_spheres(s);
	for (int i = 0; i < n; ++i)
		occluded[i] = spheres_occlude(s, ld3(rays6 + 6 * i), ld3(rays6 + 6 * i + 3)) ? 1 : 0;
}

// [pinned bit for bit]  data/camera.cpp:19-44
void orc_camera_rays(float fov, int w, int h, float* out_dir)
{
	float tanfov = tan(fov);
	float aspect = float(w) / h;
	for (int y = 0; y < h; ++y)
		for (int x = 0; x < w; ++x) {
			int i = w * y + x;
			float dirX = 2 * ((x + 0.5) / w) - 1;
			float dirY = 2 * ((y + 0.5) / h) - 1;
			dirX *= aspect;
			out_dir[3 * i + 0] = dirX * tanfov;
			out_dir[3 * i + 1] = dirY * tanfov;
			out_dir[3 * i + 2] = -1;
		}
}

// [unpinned: GLSL + GL texture storage]  prepass_fs.cpp:11-13, ras_rasterizer.cpp:293,301
void orc_gbuffer_encode(int n, const float* normal3, const float* position3, const float* alpha,
                        float* nrm_rgba, float* pos_rgb)
{
	for (int i = 0; i < n; ++i) {
		V3 nn = normalized(ld3(normal3 + 3 * i));
		for (int k = 0; k < 3; ++k) {
			float c = nn[k] * 0.5f + 0.5f;
			c = std::min(std::max(c, 0.0f), 1.0f);
			float q = nearbyintf(c * 65535.0f);          // RGBA16 UNORM store
			nrm_rgba[4 * i + k] = q / 65535.0f;          // glReadPixels(GL_FLOAT)
			pos_rgb[3 * i + k] = half_round(position3[3 * i + k]);  // RGB16F store
		}
		nrm_rgba[4 * i + 3] = alpha[i];
	}
}

// [unpinned: needs a GL context in the reference]  ras_rasterizer.cpp:509-536,556-561
int orc_gbuffer_select(int n, const float* nrm_rgba, const float* pos_rgb, int interlace,
                       int phase, float* out_pos, float* out_nrm, int* out_index)
{
	int count = 0;
	for (int i = 0; i < n; ++i) {
		if (interlace > 1 && (i % interlace) != phase) continue;
		if (nrm_rgba[4 * i + 3] > 0.0f) {
			for (int k = 0; k < 3; ++k) {
				out_nrm[3 * count + k] = (float)(nrm_rgba[4 * i + k] * 2.0 - 1.0);
				out_pos[3 * count + k] = pos_rgb[3 * i + k];
			}
			if (out_index) out_index[count] = i;
			++count;
		}
	}
	return count;
}

// [unpinned]  frame spec: DESIGN.md "Frame semantics"
void orc_render_texels(orc_scene* s, const orc_shade_config* cfg, int n, const float* pos,
                       const float* nrm, const int* material_id, uint8_t* rgba, float* rgb_float,
                       int64_t* rays_issued)
{
	orc_render_texels_uv(s, cfg, n, pos, nrm, NULL, material_id, rgba, rgb_float, rays_issued);
}

// The same with the texels' texture coordinates (n x 2, NULL = (0,0): the record's zeroed default).
void orc_render_texels_uv(orc_scene* s, const orc_shade_config* cfg, int n, const float* pos,
                          const float* nrm, const float* uv, const int* material_id, uint8_t* rgba,
                          float* rgb_float, int64_t* rays_issued)
{
	if (cfg->geometry == 0) ensure_array(s); else ensure_spheres(s);
	ShadeCtx c = {s, cfg, 0};
	for (int i = 0; i < n; ++i) {
		if (material_id[i] < 0) {
			rgba[4 * i] = rgba[4 * i + 1] = rgba[4 * i + 2] = rgba[4 * i + 3] = 0;
			if (rgb_float) rgb_float[3 * i] = rgb_float[3 * i + 1] = rgb_float[3 * i + 2] = 0;
			continue;
		}
		orc_ray r;
		memset(&r, 0, sizeof(r));
		V3 P = ld3(pos + 3 * i);
		st3(r.position, P);
		st3(r.normal, ld3(nrm + 3 * i));
		st3(r.direction, normalized(P));  // eye at the origin (prepass_vs.cpp:12-13)
		if (uv) { r.texcoord[0] = uv[2 * i]; r.texcoord[1] = uv[2 * i + 1]; }
		V3 col = mk(0, 0, 0);
		// ids past the user table address the per-sphere materials (sphere geometry only)
		HitMat hm = {material_id[i], -1};
		Material m0;
		if (material_id[i] < (int)s->materials.size()) m0 = s->materials[material_id[i]];
		else { hm.tri_material = -1; hm.sphere = material_id[i] - (int)s->materials.size(); m0 = material_of(s, hm); }
		shade(c, &r, m0, false, hm, &col, 0);
		if (rgb_float) st3(rgb_float + 3 * i, col);
		// rt_raytracer.cpp:175-178
		rgba[4 * i + 0] = (unsigned char)(std::min((int)col.x, 255));
		rgba[4 * i + 1] = (unsigned char)(std::min((int)col.y, 255));
		rgba[4 * i + 2] = (unsigned char)(std::min((int)col.z, 255));
		rgba[4 * i + 3] = 255;
	}
	if (rays_issued) *rays_issued = c.rays;
}

// [unpinned: GLSL]  mesh_fs.cpp:26-49 with reflection = 0 (raypass target itself)
void orc_shade_records(orc_scene* s, int n, const orc_ray* records, const int* material_id,
                       float* rgb)
{
	for (int i = 0; i < n; ++i) {
		const Material& m = s->materials[material_id[i]];
		V3 albedo = divs(m.diffuse, 255.0f), scol = divs(m.specular, 255.0f);
		if (m.texture >= 0)  // material_textured == 1 (mesh_fs.cpp:29-30)
			albedo = texture_gl_nearest(s->textures[m.texture], records[i].texcoord[0], records[i].texcoord[1]);
		V3 pos = ld3(records[i].position);
		V3 N = normalized(ld3(records[i].normal));
		V3 E = normalized(neg(pos));
		V3 direct = mk(0, 0, 0);
		size_t nl = std::min<size_t>(s->lights.size(), 8);
		for (size_t l = 0; l < nl; ++l) {
			V3 L = sub(s->lights[l].position, pos);
			V3 energy = mul(s->lights[l].color, 1.0f / 255);
			float len = norm(L);
			energy = mul(energy, 50.0f / (50.0f + len * len));
			L = normalized(L);
			V3 H = normalized(add(L, E));
			float lambert = std::min(std::max(dot(N, L), 0.0f), 1.0f);
			float phong = powf(std::min(std::max(dot(N, H), 0.0f), 1.0f), m.shininess);
			for (int k = 0; k < 3; ++k)
				direct[k] = energy[k] * albedo[k] * lambert + energy[k] * scol[k] * phong + direct[k];
		}
		float refl = m.reflectivity;
		for (int k = 0; k < 3; ++k) rgb[3 * i + k] = refl * 0.0f + (1.0f - refl) * direct[k];
	}
}

// [pinned with rng=0 against the compiled reference under the same seed]
int orc_photon_generate(orc_scene* s, int count, int geometry, int rng_mode, uint64_t seed)
{
	if (geometry == 0) ensure_array(s); else ensure_spheres(s);
	s->photons.clear();
	const float max_dist = 12;
	const float rmax_2 = RAND_MAX / 2.0;
	Rng rng;
	rng.mode = rng_mode;
	rng.seed_lo = (uint32_t)seed; rng.seed_hi = (uint32_t)(seed >> 32);
	rng.index = 0; rng.counter = 0;
	if (rng_mode == 0) srand((unsigned)seed);
	uint32_t photon_index = 0;
	for (size_t i = 0; i < s->lights.size(); ++i) {
		const Light& light = s->lights[i];
		V3 energy = divs(mul(light.color, 5.0f), (float)count);  // 5*color/count
		for (int j = 0; j < count;) {
			if (rng_mode == 1 && rng.counter == 0) rng.index = photon_index;
			V3 dir;
			dir.x = rng.next() / rmax_2 - 1;
			dir.y = rng.next() / rmax_2 - 1;
			dir.z = -rng.next() / (float)RAND_MAX;
			if (dot(dir, dir) > 1) continue;
			++j;
			orc_ray ray;
			memset(&ray, 0, sizeof(ray));
			st3(ray.origin, light.position);
			st3(ray.direction, dir);
			emit_photon(s, geometry, rng, &ray, energy, max_dist, 1 << 20);
			++photon_index;
			rng.counter = 0;
		}
	}
	return (int)s->photons.size();
}

// The counter-RNG sequence restricted to global photon indices [first, first+n) (index =
// light * count + j): what one rank of the sharded GPU pass emits.  Same loop body as above.
int orc_photon_generate_range(orc_scene* s, int count, int geometry, uint64_t seed, int64_t first, int64_t n)
{
	if (geometry == 0) ensure_array(s); else ensure_spheres(s);
	s->photons.clear();
	const float max_dist = 12;
	const float rmax_2 = RAND_MAX / 2.0;
	Rng rng;
	rng.mode = 1;
	rng.seed_lo = (uint32_t)seed; rng.seed_hi = (uint32_t)(seed >> 32);
	const int64_t total = (int64_t)s->lights.size() * count;
	for (int64_t g = std::max<int64_t>(first, 0); g < std::min(first + n, total); ++g) {
		const Light& light = s->lights[(size_t)(g / count)];
		V3 energy = divs(mul(light.color, 5.0f), (float)count);
		rng.index = (uint32_t)g; rng.counter = 0;
		V3 dir;
		do {
			dir.x = rng.next() / rmax_2 - 1;
			dir.y = rng.next() / rmax_2 - 1;
			dir.z = -rng.next() / (float)RAND_MAX;
		} while (dot(dir, dir) > 1);
		orc_ray ray;
		memset(&ray, 0, sizeof(ray));
		st3(ray.origin, light.position);
		st3(ray.direction, dir);
		emit_photon(s, geometry, rng, &ray, energy, max_dist, 1 << 20);
	}
	return (int)s->photons.size();
}

int orc_photon_count(orc_scene* s) { return (int)s->photons.size(); }

void orc_photon_get(orc_scene* s, float* out9)
{
	for (size_t i = 0; i < s->photons.size(); ++i) {
		st3(out9 + 9 * i, s->photons[i].position);
		st3(out9 + 9 * i + 3, s->photons[i].energy);
		st3(out9 + 9 * i + 6, s->photons[i].direction);
	}
}

void orc_photon_set(orc_scene* s, int n, const float* in9)
{
	s->photons.resize(n);
	for (int i = 0; i < n; ++i) {
		s->photons[i].position = ld3(in9 + 9 * i);
		s->photons[i].energy = ld3(in9 + 9 * i + 3);
		s->photons[i].direction = ld3(in9 + 9 * i + 6);
		s->photons[i].distance2 = 0;
	}
}

// [pinned against the compiled reference fed the identical photon array]
void orc_radiance_estimate(orc_scene* s, int npts, const float* pos, const float* nrm,
                           unsigned count, float radius, float* out)
{
	for (int i = 0; i < npts; ++i)
		st3(out + 3 * i, radiance_estimate(s, ld3(pos + 3 * i), ld3(nrm + 3 * i), count, radius));
}

// [pinned]  rt_photon_map.cpp:126-142
void orc_photon_lookup(orc_scene* s, int npts, const float* pos, float radius, float* out)
{
	float r2 = radius * radius;
	for (int i = 0; i < npts; ++i) {
		V3 p = ld3(pos + 3 * i), result = mk(0, 0, 0);
		for (size_t k = 0; k < s->photons.size(); ++k) {
			V3 d = sub(s->photons[k].position, p);
			if (dot(d, d) < r2) result = add(result, s->photons[k].energy);
		}
		st3(out + 3 * i, result);
	}
}

}  // extern "C"
