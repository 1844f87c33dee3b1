// raytracer_amd/host/rt_accel_hip.h -- the reference-side plug-in: `AccelHip` is a new IAccel
// (raytracer/rt_iaccel.h:4-11) next to AccelArray / AccelSpheres, and `RayTracerHip` offers
// RayTracer's two hot-path methods (raytracer/rt_raytracer.h:55-58) with identical signatures.
// `PhotonMapHip` offers PhotonMap's three methods (raytracer/rt_photon_map.h:13-16) over the GPU photon pass.
// All are thin: every call forwards to the C ABI of include/rt_hip.h (librt_hip.so).
#pragma once

#ifdef RT_HIP_IN_REFERENCE_TREE
#include "data/data.h"
#include "data/scene.h"
#include "raytracer/rt_iaccel.h"
typedef Eigen::Vector3f RtVec3;
#else
#include "rt_ref_mirror.h"
typedef Vec3f RtVec3;
#endif

#include <map>
#include <vector>

struct rt_hip_ctx;

class AccelHip : public IAccel
{
public:
	// semantics: 0 = AccelArray (object space, 1e-5 thresholds, t < 1e6), 1 = AccelSpheres
	explicit AccelHip(int semantics = 0, int device = 0);
	~AccelHip();
	virtual void addMesh(class Mesh* mesh);
	virtual void update();
	virtual bool occlude(class Ray* ray);
	virtual bool intersect(class Ray* ray, struct Material** material);
	// batch forms of the two queries (what the GPU is for; the virtuals are batches of one)
	void intersectBatch(Ray* rays, int n, unsigned char* hit, Material** materials);
	void occludeBatch(const Ray* rays, int n, unsigned char* occluded);
	const char* lastError() const;
	bool ok() const { return ctx != 0; }

private:
	friend class PhotonMapHip;
	AccelHip(const AccelHip&);
	AccelHip& operator=(const AccelHip&);
	int materialId(Material* m);
	void sync();
	rt_hip_ctx* ctx;
	int semantics;
	bool dirty;
	std::vector<Material*> materials;      // id -> Material* (borrowed, owned by Scene)
	std::map<Material*, int> material_ids;
	std::vector<Material*> sphere_materials;  // AccelSpheres hands out its own Material per sphere
};

// PhotonMap (raytracer/rt_photon_map.h:6-17) on the GPU: same three public methods, same argument meaning.
//   generate           the emission loops of rt_photon_map.cpp:86-124 / :32-69 as one kernel over all photon
//                      indices (counter RNG keyed by (seed, index) instead of srand(time(NULL)) + serial rand(), so the
//                      set is reproducible and shardable), then the hashed grid.  `meshes` must be an AccelHip: the
//                      geometry the photons are traced against lives in its context, and so does the map.
//   radiance_estimate  rt_photon_map.cpp:149-208 incl. the NaN of an empty neighbourhood; ray_dir is unused there too
//   lookup             rt_photon_map.cpp:126-142
// The single-point methods are batches of one (API compatibility, slow by design); the *Batch forms are what a
// GPU caller uses (shade() would call them once per level, rt_raytracer.cpp:155-157).
class PhotonMapHip
{
public:
	explicit PhotonMapHip(unsigned long long seed = 42);
	void generate(const class Scene* scene, class IAccel* meshes, int count);
	RtVec3 lookup(RtVec3 position, float radius);
	RtVec3 radiance_estimate(RtVec3 position, RtVec3 ray_dir, RtVec3 normal, unsigned int count, float radius);
	void lookupBatch(const RtVec3* positions, int n, float radius, RtVec3* out);
	void radianceEstimateBatch(const RtVec3* positions, const RtVec3* normals, int n, unsigned int count, float radius, RtVec3* out);
	long long photonCount() const { return stored; }
	// PhotonMap::photons as 9 floats each (position, energy, direction), photon-index order; returns how many were written
	long long photons(float* out9, long long capacity);

private:
	PhotonMapHip(const PhotonMapHip&);
	PhotonMapHip& operator=(const PhotonMapHip&);
	AccelHip* accel;      // borrowed: the context that holds the geometry and the map
	unsigned long long seed;
	long long stored;
};

class RayTracerHip
{
public:
	explicit RayTracerHip(int device = 0);
	~RayTracerHip();
	void addMesh(struct Mesh* mesh);                                     // rt_raytracer.cpp:69
	void processRays(const class Camera& camera, int count, RtVec3* positions, RtVec3* normals, Ray** results,
	                 ResultOffset** result_offsets, int* out_result_count, int* out_material_count);  // :428
	const char* lastError() const;

private:
	RayTracerHip(const RayTracerHip&);
	RayTracerHip& operator=(const RayTracerHip&);
	rt_hip_ctx* ctx;
	bool dirty;
	std::vector<Mesh*> meshes;
	std::vector<Material*> materials;
	std::map<Material*, int> material_ids;
	std::vector<ResultOffset> offset_vec;   // Material* restored from the dense ids
};
