// raytracer_amd/host/rt_ref_mirror.h -- layout mirror of the reference's hot-path types for
// building the host shim OUTSIDE the reference tree (no Eigen there).  Inside the reference tree
// define RT_HIP_IN_REFERENCE_TREE and the shim includes the reference's own headers instead
// (data/data.h, raytracer/rt_iaccel.h); the two sets are byte-compatible by construction and the
// static_asserts in rt_accel_hip.cpp hold for both.
#pragma once
#include <utility>
#include <vector>

struct Vec3f { float v[3]; float& operator[](int i) { return v[i]; } const float& operator[](int i) const { return v[i]; } };
struct Vec2f { float v[2]; };
struct alignas(16) Mat4f { float m[16]; const float* data() const { return m; } };  // column-major, as Eigen::Matrix4f

class Texture {                 // data/data.h:7-16 (members only)
	struct FIBITMAP* bitmap;
public:
	unsigned char* pixels;      // RGBA8, FreeImage_GetBits row order (row 0 = bottom), data/data.cpp:17-26
	unsigned int width, height;
};
struct Material {               // data/data.h:18-26
	Vec3f diffuse_color, specular_color;
	float shininess, reflectivity, alpha, ior;
	Texture* texture;
};
typedef std::pair<struct Material*, int> ResultOffset;  // data/data.h:28
struct Ray {                    // data/data.h:34-42
	Vec3f origin, direction, position, normal;
	Vec2f texcoord;
	float pad[2];
};
struct Light { Vec3f position, color; float energy; };   // data/data.h:44-48
struct Face { unsigned int v[3]; };                      // data/data.h:50-52
class Mesh {                    // data/data.h:54-69
public:
	Vec3f position;
	float radius;
	unsigned int num_verts;
	Vec3f* verts;
	Vec3f* normals;
	Vec2f* texcoords;
	Mat4f model_mat;
	unsigned int num_faces;
	Face* faces;
	Material* material;
};
class Camera;
class Scene {                   // data/scene.h:6-21 (members only)
public:
	std::vector<Light*> lights;
	std::vector<Mesh*> meshes;
	std::vector<Material*> materials;
	std::vector<Texture*> textures;
	class Camera* camera;
	class RayTracer* raytracer;
	class Rasterizer* rasterizer;
};
class IAccel {                  // raytracer/rt_iaccel.h:4-11
public:
	virtual void addMesh(class Mesh* mesh) = 0;
	virtual void update() = 0;
	virtual bool occlude(class Ray*) = 0;
	virtual bool intersect(class Ray*, struct Material**) = 0;
};
