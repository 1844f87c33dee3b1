// raytracer_amd/host/rt_accel_hip.cpp -- see rt_accel_hip.h.  Plain C++ (g++), links librt_hip.so.
#include "rt_accel_hip.h"

#include <stdio.h>
#include <string.h>

#include "../../include/rt_hip.h"

static_assert(sizeof(Ray) == sizeof(rt_ray), "reference Ray and rt_ray are the same 64 bytes");
static_assert(sizeof(Material) == sizeof(rt_material), "reference Material and rt_material are the same 48 bytes");
static_assert(sizeof(Light) == sizeof(rt_light), "reference Light and rt_light are the same 28 bytes");

namespace {
void upload_materials(rt_hip_ctx* ctx, const std::vector<Material*>& mats)
{
	std::vector<rt_material> table(mats.size());
	for (size_t i = 0; i < mats.size(); ++i) {
		memcpy(&table[i], mats[i], sizeof(rt_material));  // identical layout; Texture* rides in `texture`
	}
	rt_hip_set_materials(ctx, table.empty() ? 0 : &table[0], (int)table.size());
	// Material::texture (data/data.h:25): the decoded image goes to the GPU with the table
	for (size_t i = 0; i < mats.size(); ++i) {
		const Texture* t = mats[i]->texture;
		if (t && t->pixels && t->width && t->height)
			rt_hip_set_texture(ctx, (int)i, t->pixels, (int)t->width, (int)t->height);
	}
}
int add_mesh(rt_hip_ctx* ctx, Mesh* mesh, int material_id)
{
	return rt_hip_add_mesh(ctx, (const float*)mesh->verts, (const float*)mesh->normals, (const float*)mesh->texcoords,
	                       (int)mesh->num_verts, (const uint32_t*)mesh->faces, (int)mesh->num_faces, mesh->model_mat.data(),
	                       material_id, &mesh->position[0]);
}
}  // namespace

// ---- AccelHip --------------------------------------------------------------------------------
AccelHip::AccelHip(int semantics_, int device) : ctx(0), semantics(semantics_), dirty(true)
{
	rt_hip_config cfg;
	rt_hip_array_config(&cfg);
	cfg.device = device;
	cfg.geometry = semantics == 1 ? RT_HIP_SPHERES : RT_HIP_TRIANGLES;
	if (rt_hip_create(&cfg, &ctx) != RT_HIP_OK) {
		// no silent CPU path: the object stays unusable and every query reports a miss + error text
		fprintf(stderr, "AccelHip: %s\n", rt_hip_last_error());
		ctx = 0;
	}
}

AccelHip::~AccelHip()
{
	if (ctx) rt_hip_destroy(ctx);
	for (size_t i = 0; i < sphere_materials.size(); ++i) delete sphere_materials[i];
}

const char* AccelHip::lastError() const { return rt_hip_last_error(); }

int AccelHip::materialId(Material* m)
{
	std::map<Material*, int>::iterator it = material_ids.find(m);
	if (it != material_ids.end()) return it->second;
	int id = (int)materials.size();
	materials.push_back(m);
	material_ids[m] = id;
	return id;
}

void AccelHip::addMesh(Mesh* mesh)
{
	if (!ctx) return;
	int id = mesh->material ? materialId(mesh->material) : -1;
	if (semantics == 0 && id < 0) {
		// AccelArray stores the NULL and would hand it back on a hit; give such meshes a slot so the
		// device table stays dense -- intersect() maps it back to NULL.
		id = (int)materials.size();
		materials.push_back(0);
	}
	if (add_mesh(ctx, mesh, id) < 0) fprintf(stderr, "AccelHip::addMesh: %s\n", rt_hip_last_error());
	if (semantics == 1) {
		// the reference allocates one Material per sphere with a rand()%256 colour (rt_accel_spheres.cpp:22-24)
		Material* m = new Material();
		memset(m, 0, sizeof(Material));
		sphere_materials.push_back(m);
	}
	dirty = true;
}

void AccelHip::sync()
{
	if (!ctx || !dirty) return;
	static Material zero_material;
	std::vector<Material*> table(materials);
	for (size_t i = 0; i < table.size(); ++i)
		if (!table[i]) table[i] = &zero_material;
	upload_materials(ctx, table);
	if (rt_hip_update(ctx) != RT_HIP_OK) fprintf(stderr, "AccelHip::update: %s\n", rt_hip_last_error());
	if (semantics == 1 && !sphere_materials.empty()) {
		std::vector<float> rgb(sphere_materials.size() * 3);
		int n = rt_hip_sphere_colors(ctx, &rgb[0], (int)sphere_materials.size());
		for (int i = 0; i < n; ++i)
			for (int k = 0; k < 3; ++k) sphere_materials[i]->diffuse_color[k] = rgb[3 * i + k];
	}
	dirty = false;
}

void AccelHip::update() { sync(); }

void AccelHip::intersectBatch(Ray* rays, int n, unsigned char* hit, Material** out_materials)
{
	if (!ctx) { memset(hit, 0, n); return; }
	sync();
	std::vector<uint32_t> ids(n);
	if (rt_hip_intersect(ctx, (rt_ray*)rays, n, hit, n ? &ids[0] : 0, 0) != RT_HIP_OK) {
		fprintf(stderr, "AccelHip::intersect: %s\n", rt_hip_last_error());
		memset(hit, 0, n);
		return;
	}
	if (out_materials)
		for (int i = 0; i < n; ++i)
			if (hit[i]) out_materials[i] = semantics == 1 ? sphere_materials[ids[i]] : materials[ids[i]];
}

void AccelHip::occludeBatch(const Ray* rays, int n, unsigned char* occluded)
{
	if (!ctx) { memset(occluded, 0, n); return; }
	sync();
	if (rt_hip_occlude(ctx, (const rt_ray*)rays, n, occluded) != RT_HIP_OK) {
		fprintf(stderr, "AccelHip::occlude: %s\n", rt_hip_last_error());
		memset(occluded, 0, n);
	}
}

bool AccelHip::intersect(Ray* ray, Material** material)
{
	unsigned char hit = 0;
	Material* m = 0;
	intersectBatch(ray, 1, &hit, &m);
	// the reference writes through `material` unconditionally on a hit (rt_accel_array.cpp:131);
	// a NULL out-pointer (shade()'s shadow query, rt_raytracer.cpp:114) crashes there and is tolerated here
	if (hit && material) *material = m;
	return hit != 0;
}

bool AccelHip::occlude(Ray* ray)
{
	unsigned char occ = 0;
	occludeBatch(ray, 1, &occ);
	return occ != 0;
}

// ---- PhotonMapHip ------------------------------------------------------------------------------
PhotonMapHip::PhotonMapHip(unsigned long long seed_) : accel(0), seed(seed_), stored(0) {}

void PhotonMapHip::generate(const Scene* scene, IAccel* meshes, int count)
{
	stored = 0;
	accel = dynamic_cast<AccelHip*>(meshes);
	if (!accel || !accel->ctx) {
		fprintf(stderr, "PhotonMapHip::generate: the IAccel must be a usable AccelHip (the photons are traced on the GPU)\n");
		accel = 0;
		return;
	}
	rt_hip_ctx* ctx = accel->ctx;
	accel->sync();                                     // meshes + materials
	std::vector<rt_light> lights(scene->lights.size());
	for (size_t i = 0; i < lights.size(); ++i) memcpy(&lights[i], scene->lights[i], sizeof(rt_light));  // identical layout
	rt_hip_config cfg;
	rt_hip_array_config(&cfg);                         // AccelHip's semantics; only the shading / photon fields are taken
	cfg.seed = seed;                                   // max_dist 12, absorb 0.6: the reference's constants (rt_photon_map.cpp:92,45)
	cfg.use_photon_map = 1;
	long long n = 0;
	if (rt_hip_set_config(ctx, &cfg) != RT_HIP_OK || rt_hip_set_lights(ctx, lights.empty() ? 0 : &lights[0], (int)lights.size()) != RT_HIP_OK ||
	    rt_hip_update(ctx) != RT_HIP_OK || rt_hip_photon_generate(ctx, count, (int64_t*)&n) != RT_HIP_OK) {
		fprintf(stderr, "PhotonMapHip::generate: %s\n", rt_hip_last_error());
		return;
	}
	stored = n;
}

void PhotonMapHip::radianceEstimateBatch(const RtVec3* positions, const RtVec3* normals, int n, unsigned int count, float radius, RtVec3* out)
{
	if (n <= 0) return;
	if (!accel || rt_hip_radiance_estimate(accel->ctx, (const float*)positions, (const float*)normals, n, (int)count, radius, (float*)out) != RT_HIP_OK) {
		if (accel) fprintf(stderr, "PhotonMapHip::radiance_estimate: %s\n", rt_hip_last_error());
		memset(out, 0, sizeof(RtVec3) * (size_t)n);
	}
}

void PhotonMapHip::lookupBatch(const RtVec3* positions, int n, float radius, RtVec3* out)
{
	if (n <= 0) return;
	if (!accel || rt_hip_photon_lookup(accel->ctx, (const float*)positions, n, radius, (float*)out) != RT_HIP_OK) {
		if (accel) fprintf(stderr, "PhotonMapHip::lookup: %s\n", rt_hip_last_error());
		memset(out, 0, sizeof(RtVec3) * (size_t)n);
	}
}

RtVec3 PhotonMapHip::radiance_estimate(RtVec3 position, RtVec3, RtVec3 normal, unsigned int count, float radius)
{
	RtVec3 out;
	radianceEstimateBatch(&position, &normal, 1, count, radius, &out);
	return out;
}

RtVec3 PhotonMapHip::lookup(RtVec3 position, float radius)
{
	RtVec3 out;
	lookupBatch(&position, 1, radius, &out);
	return out;
}

long long PhotonMapHip::photons(float* out9, long long capacity)
{
	long long n = 0;
	if (!accel || rt_hip_photon_get(accel->ctx, out9, capacity, (int64_t*)&n) != RT_HIP_OK) return 0;
	return capacity == 0 ? 0 : n;
}

// ---- RayTracerHip ------------------------------------------------------------------------------
RayTracerHip::RayTracerHip(int device) : ctx(0), dirty(true)
{
	rt_hip_config cfg;
	rt_hip_default_config(&cfg);
	cfg.device = device;
	if (rt_hip_create(&cfg, &ctx) != RT_HIP_OK) {
		fprintf(stderr, "RayTracerHip: %s\n", rt_hip_last_error());
		ctx = 0;
	}
}

RayTracerHip::~RayTracerHip()
{
	if (ctx) rt_hip_destroy(ctx);
}

const char* RayTracerHip::lastError() const { return rt_hip_last_error(); }

void RayTracerHip::addMesh(Mesh* mesh)
{
	meshes.push_back(mesh);
	if (!ctx) return;
	Material* m = mesh->material;
	std::map<Material*, int>::iterator it = material_ids.find(m);
	int id;
	if (it == material_ids.end()) {
		id = (int)materials.size();
		materials.push_back(m);
		material_ids[m] = id;
	} else {
		id = it->second;
	}
	if (add_mesh(ctx, mesh, id) < 0) fprintf(stderr, "RayTracerHip::addMesh: %s\n", rt_hip_last_error());
	dirty = true;
}

void RayTracerHip::processRays(const Camera&, int count, RtVec3* positions, RtVec3* normals, Ray** results,
                               ResultOffset** result_offsets, int* out_result_count, int* out_material_count)
{
	*out_result_count = 0;
	*out_material_count = 0;
	*results = 0;
	*result_offsets = 0;
	if (!ctx) return;
	// The reference re-reads every model_mat each frame (rt_raytracer.cpp:467) and rebuilds its soup; here the
	// matrices are pushed and rt_hip_update does only what changed: nothing, or a refit of the kept BVH.
	for (size_t i = 0; i < meshes.size(); ++i) rt_hip_set_model_mat(ctx, (int)i, meshes[i]->model_mat.data());
	if (dirty) {
		// processRays' output depends on WHICH Material* a triangle carries, never on the material's contents
		// (rt_raytracer.cpp:510-539), so the table only goes up when meshes were added
		static Material zero_material;
		std::vector<Material*> table(materials);
		for (size_t i = 0; i < table.size(); ++i)
			if (!table[i]) table[i] = &zero_material;
		upload_materials(ctx, table);
		dirty = false;
	}
	if (rt_hip_update(ctx) != RT_HIP_OK) { fprintf(stderr, "RayTracerHip: %s\n", rt_hip_last_error()); return; }
	const rt_ray* res = 0;
	const rt_result_offset* off = 0;
	int nres = 0, nmat = 0;
	if (rt_hip_process_gbuffer(ctx, (const float*)positions, (const float*)normals, count, &res, &off, &nres, &nmat) != RT_HIP_OK) {
		fprintf(stderr, "RayTracerHip::processRays: %s\n", rt_hip_last_error());
		return;
	}
	offset_vec.resize(nmat);
	for (int i = 0; i < nmat; ++i) offset_vec[i] = ResultOffset(materials[off[i].material], off[i].end);
	*results = (Ray*)res;  // ctx-owned pinned memory, valid until the next call (as the reference's members are)
	*result_offsets = offset_vec.empty() ? 0 : &offset_vec[0];
	*out_result_count = nres;
	*out_material_count = nmat;
}
