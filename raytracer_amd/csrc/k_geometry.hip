// raytracer_amd/csrc/k_geometry.hip -- the per-frame world-space triangle soup
// (rt_raytracer.cpp:457-503) as one HIP kernel, plus upload of spheres / materials / lights.
//
// HBM layout produced here (all arrays indexed by the GLOBAL triangle id = mesh order, then face
// order, exactly the index the reference's dac_results.tidx carries, rt_raytracer.cpp:465-485):
//   tri_pos   48 B  (v0, material) (v1) (v2)          hit-record interpolation
//   tri_attr  64 B  n0 n1 n2 t0 t1 t2                 hit-record interpolation
//   tri_isect 48 B  (v0, prim) (e1) (e2)              the M-T operands (permuted to leaf order by the LBVH)
//   tri_box   32 B  (min, _) (max, _)                 DACTri boxes, consumed by the LBVH build
#include "rt_ctx.h"

namespace {

struct MeshXf {
	float model[16];   // column-major
	float normal[16];  // inverse-transpose, column-major
	int material;
	int use;           // apply the matrices?
	int pad[2];
};

// Matrix4f * (v, 1) with Eigen's column-accumulation order (see oracle/rt_oracle.cpp header).
__device__ __forceinline__ f3 xform(const float* m, f3 v)
{
	f3 r;
	float a;
	a = m[0] * v.x; a = m[4] * v.y + a; a = m[8] * v.z + a; a = m[12] * 1.0f + a; r.x = a;
	a = m[1] * v.x; a = m[5] * v.y + a; a = m[9] * v.z + a; a = m[13] * 1.0f + a; r.y = a;
	a = m[2] * v.x; a = m[6] * v.y + a; a = m[10] * v.z + a; a = m[14] * 1.0f + a; r.z = a;
	return r;
}

__device__ __forceinline__ uint32_t f2ord(float f)
{
	uint32_t u = __float_as_uint(f);
	return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(256) void k_soup(const float* __restrict__ verts, const float* __restrict__ normals,
                                              const float* __restrict__ texcoords,
                                              const uint32_t* __restrict__ faces,
                                              const uint32_t* __restrict__ mesh_of_tri,
                                              const MeshXf* __restrict__ xf, int num_tris,
                                              TriPos* __restrict__ tri_pos, TriAttr* __restrict__ tri_attr,
                                              TriIsect* __restrict__ tri_isect, float4* __restrict__ tri_box,
                                              uint32_t* __restrict__ bounds_ord)
{
	__shared__ uint32_t sb[6];
	if (threadIdx.x < 3) sb[threadIdx.x] = 0xFFFFFFFFu;
	else if (threadIdx.x < 6) sb[threadIdx.x] = 0u;
	__syncthreads();
	int t = blockIdx.x * 256 + threadIdx.x;
	if (t < num_tris) {
		const MeshXf& m = xf[mesh_of_tri[t]];
		uint32_t i0 = faces[3 * (size_t)t], i1 = faces[3 * (size_t)t + 1], i2 = faces[3 * (size_t)t + 2];
		f3 v0 = ld3(verts + 3 * (size_t)i0), v1 = ld3(verts + 3 * (size_t)i1), v2 = ld3(verts + 3 * (size_t)i2);
		f3 n0 = ld3(normals + 3 * (size_t)i0), n1 = ld3(normals + 3 * (size_t)i1), n2 = ld3(normals + 3 * (size_t)i2);
		if (m.use) {
			v0 = xform(m.model, v0); v1 = xform(m.model, v1); v2 = xform(m.model, v2);
			n0 = xform(m.normal, n0); n1 = xform(m.normal, n1); n2 = xform(m.normal, n2);
		}
		f3 e1 = v1 - v0, e2 = v2 - v0;
		TriPos tp;
		tp.p0 = make_float4(v0.x, v0.y, v0.z, __int_as_float(m.material));
		tp.p1 = make_float4(v1.x, v1.y, v1.z, 0.0f);
		tp.p2 = make_float4(v2.x, v2.y, v2.z, 0.0f);
		tri_pos[t] = tp;
		TriAttr ta;
		st3(ta.n0, n0); st3(ta.n1, n1); st3(ta.n2, n2);
		ta.t0[0] = texcoords[2 * (size_t)i0]; ta.t0[1] = texcoords[2 * (size_t)i0 + 1];
		ta.t1[0] = texcoords[2 * (size_t)i1]; ta.t1[1] = texcoords[2 * (size_t)i1 + 1];
		ta.t2[0] = texcoords[2 * (size_t)i2]; ta.t2[1] = texcoords[2 * (size_t)i2 + 1];
		ta.pad = 0.0f;
		tri_attr[t] = ta;
		TriIsect ti;
		ti.a = make_float4(v0.x, v0.y, v0.z, __uint_as_float((uint32_t)t));
		ti.b = make_float4(e1.x, e1.y, e1.z, 0.0f);
		ti.c = make_float4(e2.x, e2.y, e2.z, 0.0f);
		tri_isect[t] = ti;
		float mnx = fminf(fminf(v0.x, v1.x), v2.x), mny = fminf(fminf(v0.y, v1.y), v2.y), mnz = fminf(fminf(v0.z, v1.z), v2.z);
		float mxx = fmaxf(fmaxf(v0.x, v1.x), v2.x), mxy = fmaxf(fmaxf(v0.y, v1.y), v2.y), mxz = fmaxf(fmaxf(v0.z, v1.z), v2.z);
		tri_box[2 * (size_t)t] = make_float4(mnx, mny, mnz, 0.0f);
		tri_box[2 * (size_t)t + 1] = make_float4(mxx, mxy, mxz, 0.0f);
		atomicMin(&sb[0], f2ord(mnx)); atomicMin(&sb[1], f2ord(mny)); atomicMin(&sb[2], f2ord(mnz));
		atomicMax(&sb[3], f2ord(mxx)); atomicMax(&sb[4], f2ord(mxy)); atomicMax(&sb[5], f2ord(mxz));
	}
	__syncthreads();
	if (threadIdx.x < 3) atomicMin(&bounds_ord[threadIdx.x], sb[threadIdx.x]);
	else if (threadIdx.x < 6) atomicMax(&bounds_ord[threadIdx.x], sb[threadIdx.x]);
}

float ord2f(uint32_t o)
{
	uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
	float f;
	memcpy(&f, &u, 4);
	return f;
}

// inverse-transpose of a column-major 4x4 in double precision (cofactor expansion).  The
// reference uses Eigen's SSE inverse (rt_raytracer.cpp:468); agreement is ~1e-7 relative and
// exact for identity / pure translation.  Identical to the oracle's restatement.
void inverse_transpose(const float* m, float* out)
{
	double a[4][4], inv[4][4], cof[4][4];
	for (int c = 0; c < 4; ++c)
		for (int r = 0; r < 4; ++r) a[r][c] = m[c * 4 + r];
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
	double det = 0;
	for (int c = 0; c < 4; ++c) det += a[0][c] * cof[0][c];
	for (int r = 0; r < 4; ++r)
		for (int c = 0; c < 4; ++c) inv[r][c] = cof[c][r] / det;
	for (int c = 0; c < 4; ++c)
		for (int r = 0; r < 4; ++r) out[c * 4 + r] = (float)inv[c][r];
}

}  // namespace

// ---- materials / lights / textures / spheres -> device tables -------------------------------------
static int upload_tables(rt_hip_ctx* ctx)
{
	hipStream_t st = ctx->stream;
	// user materials first; for sphere geometry one material per sphere follows (the reference's
	// `new Material` with only diffuse_color assigned, rt_accel_spheres.cpp:22-24; every other
	// field is indeterminate there and defined as zero here)
	const size_t nuser = ctx->materials.size();
	const size_t nsph = ctx->cfg.geometry == RT_HIP_SPHERES ? ctx->spheres.size() : 0;
	std::vector<MaterialRec> mr(nuser + nsph);
	memset(mr.data(), 0, mr.size() * sizeof(MaterialRec));
	for (size_t i = 0; i < nsph; ++i)
		for (int k = 0; k < 3; ++k) mr[nuser + i].diffuse[k] = ctx->spheres[i].diffuse[k];
	for (size_t i = 0; i < nuser; ++i) {
		const rt_material& m = ctx->materials[i];
		for (int k = 0; k < 3; ++k) { mr[i].diffuse[k] = m.diffuse_color[k]; mr[i].specular[k] = m.specular_color[k]; }
		mr[i].shininess = m.shininess; mr[i].reflectivity = m.reflectivity; mr[i].alpha = m.alpha; mr[i].ior = m.ior;
		mr[i].pad[0] = mr[i].pad[1] = 0;
	}
	RT_CHECK(ctx->d_materials.reserve(std::max<size_t>(mr.size(), 1) * sizeof(MaterialRec)));
	if (!mr.empty())
		RT_HIP_CHECK(hipMemcpyAsync(ctx->d_materials.p, mr.data(), mr.size() * sizeof(MaterialRec), hipMemcpyHostToDevice, st));
	// textures: one record per material, texels of all textures in one pool
	ctx->any_texture = false;
	for (size_t i = 0; i < ctx->textures.size() && i < nuser; ++i) ctx->any_texture |= ctx->textures[i].w > 0;
	if (ctx->any_texture) {
		std::vector<TexRec> tr(mr.size());
		memset(tr.data(), 0, tr.size() * sizeof(TexRec));
		std::vector<uint32_t> pool;
		for (size_t i = 0; i < ctx->textures.size() && i < nuser; ++i) {
			const HostTexture& t = ctx->textures[i];
			if (t.w <= 0) continue;
			tr[i].offset = (uint32_t)pool.size(); tr[i].w = t.w; tr[i].h = t.h;
			pool.insert(pool.end(), t.texels.begin(), t.texels.end());
		}
		RT_CHECK(ctx->d_tex.reserve(tr.size() * sizeof(TexRec)));
		RT_CHECK(ctx->d_texels.reserve(pool.size() * 4));
		RT_HIP_CHECK(hipMemcpy(ctx->d_tex.p, tr.data(), tr.size() * sizeof(TexRec), hipMemcpyHostToDevice));
		RT_HIP_CHECK(hipMemcpy(ctx->d_texels.p, pool.data(), pool.size() * 4, hipMemcpyHostToDevice));
	}
	std::vector<LightRec> lr(ctx->lights.size());
	for (size_t i = 0; i < lr.size(); ++i) {
		for (int k = 0; k < 3; ++k) { lr[i].position[k] = ctx->lights[i].position[k]; lr[i].color[k] = ctx->lights[i].color[k]; }
		lr[i].energy = ctx->lights[i].energy; lr[i].pad = 0;
	}
	RT_CHECK(ctx->d_lights.reserve(std::max<size_t>(lr.size(), 1) * sizeof(LightRec)));
	if (!lr.empty())
		RT_HIP_CHECK(hipMemcpyAsync(ctx->d_lights.p, lr.data(), lr.size() * sizeof(LightRec), hipMemcpyHostToDevice, st));
	std::vector<SphereRec> sr(ctx->spheres.size());
	for (size_t i = 0; i < sr.size(); ++i) {
		sr[i].cx = ctx->spheres[i].c[0]; sr[i].cy = ctx->spheres[i].c[1]; sr[i].cz = ctx->spheres[i].c[2];
		sr[i].r = ctx->spheres[i].r;
	}
	RT_CHECK(ctx->d_spheres.reserve(std::max<size_t>(sr.size(), 1) * sizeof(SphereRec)));
	if (!sr.empty())
		RT_HIP_CHECK(hipMemcpyAsync(ctx->d_spheres.p, sr.data(), sr.size() * sizeof(SphereRec), hipMemcpyHostToDevice, st));
	RT_HIP_CHECK(hipStreamSynchronize(st));  // the staging vectors die here
	return RT_HIP_OK;
}

// ---- concatenated mesh arrays (only when meshes were added) -----------------------------------------
static int upload_meshes(rt_hip_ctx* ctx, size_t nv, size_t nt)
{
	hipStream_t st = ctx->stream;
	std::vector<float> hv(nv * 3), hn(nv * 3), ht(nv * 2);
	std::vector<uint32_t> hf(nt * 3), hm(nt);
	for (size_t mi = 0; mi < ctx->meshes.size(); ++mi) {
		const HostMesh& m = ctx->meshes[mi];
		size_t mv = m.verts.size() / 3, mf = m.faces.size() / 3;
		if (mv) {
			memcpy(&hv[m.vert_offset * 3], m.verts.data(), mv * 12);
			memcpy(&hn[m.vert_offset * 3], m.normals.data(), mv * 12);
			memcpy(&ht[m.vert_offset * 2], m.texcoords.data(), mv * 8);
		}
		for (size_t f = 0; f < mf * 3; ++f) hf[(size_t)m.first_tri * 3 + f] = m.faces[f] + (uint32_t)m.vert_offset;
		for (size_t f = 0; f < mf; ++f) hm[m.first_tri + f] = (uint32_t)mi;
	}
	RT_CHECK(ctx->d_verts.reserve(hv.size() * 4));
	RT_CHECK(ctx->d_normals.reserve(hn.size() * 4));
	RT_CHECK(ctx->d_texcoords.reserve(ht.size() * 4));
	RT_CHECK(ctx->d_faces.reserve(hf.size() * 4));
	RT_CHECK(ctx->d_mesh_of_tri.reserve(hm.size() * 4));
	RT_HIP_CHECK(hipMemcpyAsync(ctx->d_verts.p, hv.data(), hv.size() * 4, hipMemcpyHostToDevice, st));
	RT_HIP_CHECK(hipMemcpyAsync(ctx->d_normals.p, hn.data(), hn.size() * 4, hipMemcpyHostToDevice, st));
	RT_HIP_CHECK(hipMemcpyAsync(ctx->d_texcoords.p, ht.data(), ht.size() * 4, hipMemcpyHostToDevice, st));
	RT_HIP_CHECK(hipMemcpyAsync(ctx->d_faces.p, hf.data(), hf.size() * 4, hipMemcpyHostToDevice, st));
	RT_HIP_CHECK(hipMemcpyAsync(ctx->d_mesh_of_tri.p, hm.data(), hm.size() * 4, hipMemcpyHostToDevice, st));
	RT_HIP_CHECK(hipStreamSynchronize(st));  // the staging vectors die here
	return RT_HIP_OK;
}

// ---- per-mesh matrices (every update that touches the geometry; a few hundred bytes) -------------------
static int upload_xf(rt_hip_ctx* ctx)
{
	std::vector<MeshXf> hx(ctx->meshes.size());
	for (size_t mi = 0; mi < ctx->meshes.size(); ++mi) {
		const HostMesh& m = ctx->meshes[mi];
		memcpy(hx[mi].model, m.model, sizeof(m.model));
		inverse_transpose(m.model, hx[mi].normal);
		hx[mi].material = m.material;
		hx[mi].use = ctx->cfg.use_model_mat ? 1 : 0;
		hx[mi].pad[0] = hx[mi].pad[1] = 0;
	}
	RT_CHECK(ctx->d_mesh_xf.reserve(std::max<size_t>(hx.size(), 1) * sizeof(MeshXf)));
	if (!hx.empty()) RT_HIP_CHECK(hipMemcpy(ctx->d_mesh_xf.p, hx.data(), hx.size() * sizeof(MeshXf), hipMemcpyHostToDevice));
	return RT_HIP_OK;
}

// The world-space soup (rt_raytracer.cpp:457-503) into tri_pos / tri_attr / tri_box and `isect_out` (original
// order), and the scene bounds back on the host.
static int run_soup(rt_hip_ctx* ctx, TriIsect* isect_out)
{
	hipStream_t st = ctx->stream;
	const size_t nt = (size_t)ctx->num_tris;
	RT_CHECK(ctx->d_bounds.reserve(64));
	uint32_t* h = (uint32_t*)ctx->h_small + 768;  // pinned
	const uint32_t init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
	memcpy(h, init, sizeof(init));
	RT_HIP_CHECK(hipMemcpyAsync(ctx->d_bounds.p, h, sizeof(init), hipMemcpyHostToDevice, st));
	hipLaunchKernelGGL(k_soup, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, ctx->d_verts.as<float>(),
	                   ctx->d_normals.as<float>(), ctx->d_texcoords.as<float>(), ctx->d_faces.as<uint32_t>(),
	                   ctx->d_mesh_of_tri.as<uint32_t>(), ctx->d_mesh_xf.as<MeshXf>(), (int)nt,
	                   ctx->d_tri_pos.as<TriPos>(), ctx->d_tri_attr.as<TriAttr>(), isect_out,
	                   ctx->d_tri_box.as<float4>(), ctx->d_bounds.as<uint32_t>());
	RT_HIP_CHECK(hipGetLastError());
	RT_HIP_CHECK(hipMemcpyAsync(h + 8, ctx->d_bounds.p, sizeof(init), hipMemcpyDeviceToHost, st));
	RT_HIP_CHECK(hipStreamSynchronize(st));
	for (int k = 0; k < 6; ++k) ctx->bounds[k] = ord2f(h[8 + k]);
	return RT_HIP_OK;
}

static int finish_geometry_timing(rt_hip_ctx* ctx, int kind)
{
	RT_HIP_CHECK(hipEventRecord(ctx->ev1, ctx->stream));
	RT_HIP_CHECK(hipEventSynchronize(ctx->ev1));
	float ms = 0;
	RT_HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
	ctx->stats.build_ms = ms;
	ctx->stats.num_tris = ctx->num_tris;
	ctx->stats.num_nodes = ctx->num_nodes;
	ctx->stats.bvh_depth = ctx->bvh_depth;
	ctx->stats.last_update = kind;
	ctx->stats.bvh_area_cost = ctx->use_bvh ? ctx->bvh_cost_now : 0.0;
	ctx->stats.bvh_area_cost_built = ctx->use_bvh ? ctx->bvh_cost_built : 0.0;
	return RT_HIP_OK;
}

int rt_build_geometry(rt_hip_ctx* ctx)
{
	hipStream_t st = ctx->stream;
	if (ctx->dirty_tables || ctx->dirty_topology) RT_CHECK(upload_tables(ctx));
	if (!ctx->dirty_topology && !ctx->dirty_xf) { ctx->stats.last_update = RT_HIP_UPDATE_TABLES; return RT_HIP_OK; }

	const bool want_bvh_kind = ctx->cfg.accel == RT_HIP_ACCEL_LBVH || ctx->cfg.accel == RT_HIP_ACCEL_PLOC ||
	                           ctx->cfg.accel == RT_HIP_ACCEL_SAH || ctx->cfg.accel == RT_HIP_ACCEL_AUTO;
	// ---- refit: only model matrices changed -------------------------------------------------------
	if (!ctx->dirty_topology && ctx->num_tris > 0 && ctx->tuning.refit_max_growth > 0.0f) {
		RT_CHECK(upload_xf(ctx));
		RT_HIP_CHECK(hipEventRecord(ctx->ev0, st));
		if (!ctx->use_bvh && !(want_bvh_kind && ctx->num_nodes > 0)) {
			// brute-force scenes: the soup is all there is
			RT_CHECK(run_soup(ctx, ctx->d_tri_isect.as<TriIsect>()));
			return finish_geometry_timing(ctx, RT_HIP_UPDATE_REFIT);
		}
		if (ctx->use_bvh) {
			RT_CHECK(ctx->d_in_a.reserve((size_t)ctx->num_tris * sizeof(TriIsect)));
			RT_CHECK(run_soup(ctx, ctx->d_in_a.as<TriIsect>()));
			float cost = 0.0f;
			RT_CHECK(rt_refit_lbvh(ctx, ctx->d_in_a.as<TriIsect>(), &cost));
			ctx->bvh_cost_now = cost;
			if (cost <= ctx->tuning.refit_max_growth * ctx->bvh_cost_built) return finish_geometry_timing(ctx, RT_HIP_UPDATE_REFIT);
			// the tree no longer fits the geometry: fall through to a build from scratch (matrices are uploaded already)
		}
	}

	// ---- everything from scratch --------------------------------------------------------------------
	size_t nv = 0, nt = 0;
	for (auto& m : ctx->meshes) { m.vert_offset = nv; m.first_tri = (uint32_t)nt; nv += m.verts.size() / 3; nt += m.faces.size() / 3; }
	if (nt > 0x7FFFFFF0ull) { rt_set_error("too many triangles (%zu)", nt); return RT_HIP_ERR_LIMIT; }
	ctx->num_tris = (int)nt;
	ctx->num_nodes = 0;
	ctx->use_bvh = 0;
	ctx->bvh_depth = 0;
	if (nt == 0) { ctx->stats.last_update = RT_HIP_UPDATE_REBUILD; ctx->stats.num_tris = 0; ctx->stats.num_nodes = 0; ctx->stats.bvh_depth = 0; return RT_HIP_OK; }
	if (ctx->dirty_topology) RT_CHECK(upload_meshes(ctx, nv, nt));
	RT_CHECK(upload_xf(ctx));
	RT_CHECK(ctx->d_tri_pos.reserve(nt * sizeof(TriPos)));
	RT_CHECK(ctx->d_tri_attr.reserve(nt * sizeof(TriAttr)));
	RT_CHECK(ctx->d_tri_isect.reserve(nt * sizeof(TriIsect)));
	RT_CHECK(ctx->d_tri_box.reserve(nt * 2 * sizeof(float4)));

	RT_HIP_CHECK(hipEventRecord(ctx->ev0, st));
	RT_CHECK(run_soup(ctx, ctx->d_tri_isect.as<TriIsect>()));
	bool want_bvh = ctx->cfg.accel == RT_HIP_ACCEL_LBVH || ctx->cfg.accel == RT_HIP_ACCEL_PLOC ||
	                ctx->cfg.accel == RT_HIP_ACCEL_SAH ||
	                (ctx->cfg.accel == RT_HIP_ACCEL_AUTO && nt >= 512);
	if (want_bvh && nt >= 2) RT_CHECK(rt_build_lbvh(ctx));
	return finish_geometry_timing(ctx, RT_HIP_UPDATE_REBUILD);
}
