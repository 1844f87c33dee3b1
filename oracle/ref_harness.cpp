// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
//
// Drives the *compiled reference* (Kupoman/raytracer, sources compiled where they lie under
// /root/reference by oracle/Makefile -> oracle/_ref/libref_oracle.so) through a flat C ABI so
// that Python (ctypes) can (1) generate the golden vectors under tests/golden/ and (2) pin the
// CPU restatement in oracle/rt_oracle.cpp bit for bit.  Nothing here restates an algorithm: every
// number that leaves this file was computed by the reference's own object code.
//
// Reference entry points driven (all paths relative to /root/reference):
//   RayTracer::addMesh / processRays          raytracer/rt_raytracer.cpp:69, :428
//   dac_results / dac_rays (extern globals)    raytracer/rt_raytracer.cpp:36, :48
//   AccelArray::addMesh/intersect/occlude      raytracer/rt_accel_array.cpp:17, :75, :45
//   AccelSpheres::addMesh/intersect/occlude    raytracer/rt_accel_spheres.cpp:27, :58, :38
//   PhotonMap::generate/radiance_estimate/lookup  raytracer/rt_photon_map.cpp:86, :149, :126
//   Camera::getScreenRays                      data/camera.cpp:19
//
// No reference constructor/destructor that lives in an uncompilable file (data/data.cpp,
// data/scene.cpp) is re-defined here: Mesh and Scene objects are carved out of zeroed storage
// (both are plain aggregates of PODs / std::vectors, and an all-zero libstdc++ std::vector is a
// valid empty vector), so no stand-in for reference code is linked.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <vector>

#include "data/data.h"
#include "data/scene.h"
#include "data/camera.h"
#include "raytracer/rt_iaccel.h"
#include "raytracer/rt_accel_array.h"
#include "raytracer/rt_accel_spheres.h"
#include "raytracer/rt_raytracer.h"
#define private public  // PhotonMap::photons is the only way to read/feed photons
#include "raytracer/rt_photon_map.h"
#undef private

// Layout of the reference's file-scope globals (rt_raytracer.cpp:30-48), re-declared to read them.
struct DACRay {
	Eigen::Vector3f origin;
	Eigen::Vector3f direction;
	float t;
	unsigned int id;
};
struct DACRayResult {
	float u;
	float v;
	unsigned int tidx;
	Eigen::Vector3f origin;
	Eigen::Vector3f direction;
	float t;
};
extern std::vector<DACRay> dac_rays;
extern std::vector<DACRayResult> dac_results;

// rt_photon_map.cpp:22-30 (class is local to that TU; layout re-declared to read the pointers).
class Photon {
public:
	Eigen::Vector3f position;
	Eigen::Vector3f energy;
	Eigen::Vector3f direction;
	float distance2;
	short flag;
};

static_assert(sizeof(Ray) == 64, "reference Ray is 64 bytes");
static_assert(sizeof(Material) == 48, "reference Material is 48 bytes");
static_assert(sizeof(DACRayResult) == 40, "DACRayResult is 40 bytes");
static_assert(sizeof(Photon) == 44, "Photon is 44 bytes");

namespace {
struct State {
	std::vector<Material*> materials;
	std::vector<Mesh*> meshes;
	std::vector<Light*> lights;
	RayTracer* tracer = nullptr;
	AccelArray* array = nullptr;
	AccelSpheres* spheres = nullptr;
	PhotonMap* pmap = nullptr;
	Scene* scene = nullptr;
};
State g;

int material_index(Material* m)
{
	for (size_t i = 0; i < g.materials.size(); ++i)
		if (g.materials[i] == m) return (int)i;
	return -1;
}
}  // namespace

extern "C" {

// Leaks on purpose (test process; the reference leaks too).
void ref_reset(void)
{
	g = State();
}

int ref_sizeof_ray(void) { return (int)sizeof(Ray); }
int ref_sizeof_material(void) { return (int)sizeof(Material); }
int ref_sizeof_light(void) { return (int)sizeof(Light); }
int ref_sizeof_tri(void) { return (int)sizeof(Tri); }

int ref_add_material(const float* diffuse, const float* specular, float shininess,
                     float reflectivity, float alpha, float ior)
{
	Material* m = (Material*)calloc(1, sizeof(Material));
	m->diffuse_color = Eigen::Vector3f(diffuse[0], diffuse[1], diffuse[2]);
	m->specular_color = Eigen::Vector3f(specular[0], specular[1], specular[2]);
	m->shininess = shininess;
	m->reflectivity = reflectivity;
	m->alpha = alpha;
	m->ior = ior;
	m->texture = NULL;
	g.materials.push_back(m);
	return (int)g.materials.size() - 1;
}

// Geometry arrays are copied; model_mat is column-major (Eigen's default storage).
int ref_add_mesh(const float* verts, const float* normals, const float* texcoords, int nverts,
                 const unsigned int* faces, int nfaces, const float* model_mat16,
                 int material_id, const float* position3)
{
	void* mem = NULL;
	if (posix_memalign(&mem, 16, sizeof(Mesh))) return -1;
	memset(mem, 0, sizeof(Mesh));
	Mesh* mesh = (Mesh*)mem;  // no Mesh::Mesh(): it lives in the uncompilable data/data.cpp
	mesh->num_verts = nverts;
	mesh->verts = new Eigen::Vector3f[nverts];
	mesh->normals = new Eigen::Vector3f[nverts];
	mesh->texcoords = new Eigen::Vector2f[nverts];
	memcpy(mesh->verts, verts, sizeof(float) * 3 * nverts);
	memcpy(mesh->normals, normals, sizeof(float) * 3 * nverts);
	memcpy(mesh->texcoords, texcoords, sizeof(float) * 2 * nverts);
	mesh->num_faces = nfaces;
	mesh->faces = new Face[nfaces];
	memcpy(mesh->faces, faces, sizeof(unsigned int) * 3 * nfaces);
	for (int c = 0; c < 4; ++c)
		for (int r = 0; r < 4; ++r) mesh->model_mat(r, c) = model_mat16[c * 4 + r];
	mesh->material = (material_id >= 0) ? g.materials[material_id] : NULL;
	if (position3) mesh->position = Eigen::Vector3f(position3[0], position3[1], position3[2]);
	mesh->radius = 1.0f;
	g.meshes.push_back(mesh);
	return (int)g.meshes.size() - 1;
}

int ref_add_light(const float* position, const float* color, float energy)
{
	Light* l = (Light*)calloc(1, sizeof(Light));
	l->position = Eigen::Vector3f(position[0], position[1], position[2]);
	l->color = Eigen::Vector3f(color[0], color[1], color[2]);
	l->energy = energy;
	g.lights.push_back(l);
	return (int)g.lights.size() - 1;
}

// ---- RayTracer::processRays (rt_raytracer.cpp:428) ------------------------------------------
// positions/normals: count x 3 floats.  Per-ray outputs (indexed by input ray id) come from the
// reference's dac_results global; the grouped hit records are copied out as raw 64-byte Rays.
// Returns the number of hit records; *n_materials receives the number of ResultOffset entries.
int ref_process_rays(int count, const float* positions, const float* normals,
                     unsigned int* out_tidx, float* out_t, float* out_u, float* out_v,
                     float* out_dir /* count x 3 */, void* out_records /* count x 64 B */,
                     int* out_offset_material /* <= #materials */, int* out_offset_end,
                     int* n_materials)
{
	if (!g.tracer) {
		g.tracer = new RayTracer();
		for (size_t i = 0; i < g.meshes.size(); ++i) g.tracer->addMesh(g.meshes[i]);
	}
	std::vector<Eigen::Vector3f> pos(count), nrm(count);
	memcpy(pos.data(), positions, sizeof(float) * 3 * count);
	memcpy(nrm.data(), normals, sizeof(float) * 3 * count);
	Ray* results = NULL;
	ResultOffset* offsets = NULL;
	int nres = 0, nmat = 0;
	void* cam_mem = calloc(1, sizeof(Camera));  // processRays never touches the camera (:447)
	g.tracer->processRays(*(Camera*)cam_mem, count, pos.data(), nrm.data(), &results, &offsets,
	                      &nres, &nmat);
	free(cam_mem);
	for (int i = 0; i < count; ++i) {
		out_tidx[i] = dac_results[i].tidx;
		out_t[i] = dac_results[i].t;
		out_u[i] = dac_results[i].u;
		out_v[i] = dac_results[i].v;
	}
	// dac_rays is permuted in place by trace(); un-permute directions by ray id.
	for (int i = 0; i < count; ++i) {
		const DACRay& r = dac_rays[i];
		out_dir[3 * r.id + 0] = r.direction[0];
		out_dir[3 * r.id + 1] = r.direction[1];
		out_dir[3 * r.id + 2] = r.direction[2];
	}
	if (nres > 0) memcpy(out_records, results, (size_t)nres * sizeof(Ray));
	for (int i = 0; i < nmat; ++i) {
		out_offset_material[i] = material_index(offsets[i].first);
		out_offset_end[i] = offsets[i].second;
	}
	*n_materials = nmat;
	return nres;
}

// ---- AccelArray (rt_accel_array.cpp) ---------------------------------------------------------
static void ensure_array()
{
	if (!g.array) {
		g.array = new AccelArray();
		for (size_t i = 0; i < g.meshes.size(); ++i) g.array->addMesh(g.meshes[i]);
		g.array->update();
	}
}

// rays: n x 6 floats (origin, direction).  records: n x 64 B, each the Ray after the call
// (untouched on a miss, exactly as the reference leaves it).  hit: n bytes.  material: n ints.
void ref_array_intersect(int n, const float* rays, unsigned char* hit, void* records,
                         int* material)
{
	ensure_array();
	Ray* out = (Ray*)records;
	for (int i = 0; i < n; ++i) {
		Ray r;
		memset(&r, 0, sizeof(r));
		r.origin = Eigen::Vector3f(rays[6 * i + 0], rays[6 * i + 1], rays[6 * i + 2]);
		r.direction = Eigen::Vector3f(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
		Material* m = NULL;
		bool h = g.array->intersect(&r, &m);
		hit[i] = h ? 1 : 0;
		material[i] = h ? material_index(m) : -1;
		memcpy(&out[i], &r, sizeof(Ray));
	}
}

void ref_array_occlude(int n, const float* rays, unsigned char* occluded)
{
	ensure_array();
	for (int i = 0; i < n; ++i) {
		Ray r;
		memset(&r, 0, sizeof(r));
		r.origin = Eigen::Vector3f(rays[6 * i + 0], rays[6 * i + 1], rays[6 * i + 2]);
		r.direction = Eigen::Vector3f(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
		occluded[i] = g.array->occlude(&r) ? 1 : 0;
	}
}

// ---- AccelSpheres (rt_accel_spheres.cpp) -----------------------------------------------------
static void ensure_spheres()
{
	if (!g.spheres) {
		g.spheres = new AccelSpheres();  // srand(420)
		for (size_t i = 0; i < g.meshes.size(); ++i) g.spheres->addMesh(g.meshes[i]);
	}
}

// diffuse: n x 3 floats -- the diffuse colour of the Material* the reference returned (the
// rand()%256 sequence under srand(420)), zeros on a miss.
void ref_spheres_intersect(int n, const float* rays, unsigned char* hit, void* records,
                           float* diffuse)
{
	ensure_spheres();
	Ray* out = (Ray*)records;
	for (int i = 0; i < n; ++i) {
		Ray r;
		memset(&r, 0, sizeof(r));
		r.origin = Eigen::Vector3f(rays[6 * i + 0], rays[6 * i + 1], rays[6 * i + 2]);
		r.direction = Eigen::Vector3f(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
		Material* m = NULL;
		bool h = g.spheres->intersect(&r, &m);
		hit[i] = h ? 1 : 0;
		for (int k = 0; k < 3; ++k) diffuse[3 * i + k] = h ? m->diffuse_color[k] : 0.0f;
		memcpy(&out[i], &r, sizeof(Ray));
	}
}

void ref_spheres_occlude(int n, const float* rays, unsigned char* occluded)
{
	ensure_spheres();
	for (int i = 0; i < n; ++i) {
		Ray r;
		memset(&r, 0, sizeof(r));
		r.origin = Eigen::Vector3f(rays[6 * i + 0], rays[6 * i + 1], rays[6 * i + 2]);
		r.direction = Eigen::Vector3f(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
		occluded[i] = g.spheres->occlude(&r) ? 1 : 0;
	}
}

// ---- PhotonMap (rt_photon_map.cpp) -----------------------------------------------------------
// generate() seeds libc with srand(time(NULL)) (:94).  The harness does not interpose time():
// it brackets the call with time() and only accepts a run that stayed inside one second, so the
// seed the reference used is known and returned in *seed_used.  accel: 0 = AccelArray, 1 = spheres.
// Returns the number of stored photons, or -1 if 20 attempts all straddled a second boundary.
int ref_photon_generate(int count, int accel, long long* seed_used)
{
	if (!g.scene) {
		g.scene = (Scene*)calloc(1, sizeof(Scene));  // no Scene::Scene(): data/scene.cpp is uncompilable
		for (size_t i = 0; i < g.lights.size(); ++i) g.scene->lights.push_back(g.lights[i]);
	}
	IAccel* acc;
	if (accel == 0) { ensure_array(); acc = g.array; }
	else { ensure_spheres(); acc = g.spheres; }
	for (int attempt = 0; attempt < 20; ++attempt) {
		PhotonMap* pm = new PhotonMap();
		FILE* saved = NULL; (void)saved;
		time_t t0 = time(NULL);
		fflush(stdout);
		pm->generate(g.scene, acc, count);  // prints count + energy to stdout (:116,123)
		time_t t1 = time(NULL);
		if (t0 == t1) {
			g.pmap = pm;
			*seed_used = (long long)t0;
			return (int)pm->photons.size();
		}
	}
	return -1;
}

int ref_photon_count(void) { return g.pmap ? (int)g.pmap->photons.size() : 0; }

// out: n x 9 floats (position, energy, direction)
void ref_photon_get(float* out)
{
	if (!g.pmap) return;
	for (size_t i = 0; i < g.pmap->photons.size(); ++i) {
		const Photon* p = (const Photon*)g.pmap->photons[i];
		for (int k = 0; k < 3; ++k) {
			out[9 * i + k] = p->position[k];
			out[9 * i + 3 + k] = p->energy[k];
			out[9 * i + 6 + k] = p->direction[k];
		}
	}
}

// Feed an explicit photon set (pins the gather independently of the emission RNG).
void ref_photon_set(int n, const float* in)
{
	g.pmap = new PhotonMap();
	for (int i = 0; i < n; ++i) {
		Photon* p = new Photon();
		for (int k = 0; k < 3; ++k) {
			p->position[k] = in[9 * i + k];
			p->energy[k] = in[9 * i + 3 + k];
			p->direction[k] = in[9 * i + 6 + k];
		}
		p->distance2 = 0;
		p->flag = 0;
		g.pmap->photons.push_back((class Photon*)p);
	}
}

// pos/nrm: npts x 3; out: npts x 3.  ray_dir is unused by the reference (:149) -> zeros.
void ref_radiance_estimate(int npts, const float* pos, const float* nrm, unsigned int count,
                           float radius, float* out)
{
	for (int i = 0; i < npts; ++i) {
		Eigen::Vector3f r = g.pmap->radiance_estimate(
		    Eigen::Vector3f(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]), Eigen::Vector3f(0, 0, 0),
		    Eigen::Vector3f(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]), count, radius);
		out[3 * i] = r[0]; out[3 * i + 1] = r[1]; out[3 * i + 2] = r[2];
	}
}

void ref_photon_lookup(int npts, const float* pos, float radius, float* out)
{
	for (int i = 0; i < npts; ++i) {
		Eigen::Vector3f r = g.pmap->lookup(
		    Eigen::Vector3f(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]), radius);
		out[3 * i] = r[0]; out[3 * i + 1] = r[1]; out[3 * i + 2] = r[2];
	}
}

// ---- Camera::getScreenRays (data/camera.cpp:19) ----------------------------------------------
// out_dir: w*h x 3 floats, row 0 = bottom (y ascending).
void ref_camera_rays(float fov, int w, int h, float* out_dir)
{
	Camera cam(fov, w, h);
	Ray* rays = cam.getScreenRays();
	for (int i = 0; i < w * h; ++i)
		for (int k = 0; k < 3; ++k) out_dir[3 * i + k] = rays[i].direction[k];
}

}  // extern "C"

// ---- drop-in demonstration (libref_dropin.so only) ----------------------------------------------
// The reference's own object code (PhotonMap::generate, rt_photon_map.cpp:86) driving the HIP back
// end through the IAccel interface: AccelHip (raytracer_amd/host/rt_accel_hip.cpp, compiled here
// against the reference's headers) takes AccelArray's place.  Built by `make -C oracle dropin`.
#ifdef WITH_ACCEL_HIP
#include "rt_accel_hip.h"

namespace {
AccelHip* g_hip = nullptr;
RayTracerHip* g_rthip = nullptr;
}

extern "C" {

int dropin_photon_generate(int count, long long* seed_used)
{
	if (!g.scene) {
		g.scene = (Scene*)calloc(1, sizeof(Scene));
		for (size_t i = 0; i < g.lights.size(); ++i) g.scene->lights.push_back(g.lights[i]);
	}
	if (!g_hip) {
		g_hip = new AccelHip(0, 0);
		if (!g_hip->ok()) return -2;
		for (size_t i = 0; i < g.meshes.size(); ++i) g_hip->addMesh(g.meshes[i]);
		g_hip->update();
	}
	for (int attempt = 0; attempt < 20; ++attempt) {
		PhotonMap* pm = new PhotonMap();
		time_t t0 = time(NULL);
		pm->generate(g.scene, g_hip, count);
		time_t t1 = time(NULL);
		if (t0 == t1) {
			g.pmap = pm;
			*seed_used = (long long)t0;
			return (int)pm->photons.size();
		}
	}
	return -1;
}

// The reference-shaped photon sequence with the GPU photon map in PhotonMap's place: generate once
// (rt_raytracer.cpp:165-167), then radiance_estimate per shading point (rt_raytracer.cpp:155-157) and lookup.
// The first `single` points go through the single-point methods with the reference's signatures, the rest
// through the batch forms.  Returns the number of photons stored (copied to photons_out, 9 floats each).
long long dropin_photon_map_hip(int count, unsigned long long seed, int npts, const float* pos, const float* nrm,
                                unsigned int k, float radius, int single, float* out_est, float* out_lookup,
                                float* photons_out, long long max_photons)
{
	if (!g.scene) {
		g.scene = (Scene*)calloc(1, sizeof(Scene));
		for (size_t i = 0; i < g.lights.size(); ++i) g.scene->lights.push_back(g.lights[i]);
	}
	if (!g_hip) {
		g_hip = new AccelHip(0, 0);
		if (!g_hip->ok()) return -2;
		for (size_t i = 0; i < g.meshes.size(); ++i) g_hip->addMesh(g.meshes[i]);
	}
	PhotonMapHip* pm = new PhotonMapHip(seed);
	pm->generate(g.scene, g_hip, count);
	const Eigen::Vector3f* P = (const Eigen::Vector3f*)pos;
	const Eigen::Vector3f* N = (const Eigen::Vector3f*)nrm;
	Eigen::Vector3f* E = (Eigen::Vector3f*)out_est;
	Eigen::Vector3f* L = (Eigen::Vector3f*)out_lookup;
	if (single > npts) single = npts;
	for (int i = 0; i < single; ++i) {
		E[i] = pm->radiance_estimate(P[i], Eigen::Vector3f(0, 0, -1), N[i], k, radius);
		L[i] = pm->lookup(P[i], radius);
	}
	pm->radianceEstimateBatch(P + single, N + single, npts - single, k, radius, E + single);
	pm->lookupBatch(P + single, npts - single, radius, L + single);
	long long n = pm->photonCount();
	if (n > max_photons) return -3;
	if (n > 0 && pm->photons(photons_out, max_photons) != n) return -4;
	delete pm;
	return n;
}

int dropin_process_rays(int count, const float* positions, const float* normals, void* out_records,
                        int* out_offset_material, int* out_offset_end, int* n_materials)
{
	if (!g_rthip) {
		g_rthip = new RayTracerHip(0);
		for (size_t i = 0; i < g.meshes.size(); ++i) g_rthip->addMesh(g.meshes[i]);
	}
	std::vector<Eigen::Vector3f> pos(count), nrm(count);
	memcpy(pos.data(), positions, sizeof(float) * 3 * count);
	memcpy(nrm.data(), normals, sizeof(float) * 3 * count);
	Ray* results = NULL;
	ResultOffset* offsets = NULL;
	int nres = 0, nmat = 0;
	void* cam_mem = calloc(1, sizeof(Camera));
	g_rthip->processRays(*(Camera*)cam_mem, count, pos.data(), nrm.data(), &results, &offsets, &nres, &nmat);
	free(cam_mem);
	if (nres > 0) memcpy(out_records, results, (size_t)nres * sizeof(Ray));
	for (int i = 0; i < nmat; ++i) {
		out_offset_material[i] = material_index(offsets[i].first);
		out_offset_end[i] = offsets[i].second;
	}
	*n_materials = nmat;
	return nres;
}

void dropin_reset(void)
{
	delete g_hip; g_hip = nullptr;
	delete g_rthip; g_rthip = nullptr;
}

}  // extern "C"
#endif
