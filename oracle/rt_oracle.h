/* oracle/rt_oracle.h -- C ABI of the CPU restatement (TEST INFRASTRUCTURE, see rt_oracle.cpp). */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

/* 64-byte hit record, byte-identical to the reference's struct Ray (data/data.h:34-42). */
typedef struct {
	float origin[3];
	float direction[3];
	float position[3];
	float normal[3];
	float texcoord[2];
	float pad[2];
} orc_ray;

/* Config of the Gen-1 shading restatement (rt_raytracer.cpp:50-61 ctor constants). */
typedef struct {
	int bounces;           /* 2 */
	int do_shadows;        /* 0 */
	int use_photon_map;    /* 0: photon_map == NULL */
	int photon_estimate;   /* 100 */
	float photon_radius;   /* 1.0 */
	int geometry;          /* 0 = triangles with AccelArray semantics, 1 = spheres */
	float shadow_t_max;    /* 0 = literal (any hit shadows); > 0: only hits with t < shadow_t_max (triangles) */
} orc_shade_config;

orc_scene* orc_scene_create(void);
void orc_scene_destroy(orc_scene* s);
int orc_add_material(orc_scene* s, const float* diffuse3, const float* specular3, float shininess,
                     float reflectivity, float alpha, float ior);
/* Material::texture (data/data.h:25): decoded RGBA8 image, row 0 = bottom scanline; NULL removes. */
int orc_set_texture(orc_scene* s, int material_id, const uint8_t* rgba, int width, int height);
int orc_add_mesh(orc_scene* s, const float* verts, const float* normals, const float* texcoords,
                 int nverts, const uint32_t* faces, int nfaces, const float* model_mat16,
                 int material_id, const float* position3);
int orc_add_light(orc_scene* s, const float* position3, const float* color3, float energy);
int orc_num_tris(orc_scene* s);

/* a7: reflection ray generation, rt_raytracer.cpp:444-455 */
void orc_reflect_raygen(int n, const float* pos, const float* nrm, float* out_org, float* out_dir);

/* a10: Woo ray/AABB, rt_raytracer.cpp:194-248 */
int orc_ray_aabb(const float* origin3, const float* dir3, const float* min3, const float* max3);

/* a8: world-space soup (v0,e1,e2 per tri: ntris x 9 floats), rt_raytracer.cpp:457-503 */
void orc_get_soup(orc_scene* s, float* v0e1e2, float* bounds6);

/* a7-a12: RayTracer::processRays restated (DACRT), rt_raytracer.cpp:428-540.
 * Groups are emitted in ascending material id (the reference uses Material* address order). */
int orc_process_rays(orc_scene* s, int count, const float* positions, const float* normals,
                     uint32_t* out_tidx, float* out_t, float* out_u, float* out_v, float* out_dir,
                     orc_ray* out_records, int* out_offset_material, int* out_offset_end,
                     int* n_materials);

/* a11 as an exhaustive closest-hit over the world soup (naiveTrace over everything); ties keep
 * the lowest triangle index.  eps_det / eps_t are float thresholds ("x < eps" rejects). */
void orc_soup_closest(orc_scene* s, int n, const float* org, const float* dir, float eps_det,
                      float eps_t, float t_max, uint32_t* out_tidx, float* out_t, float* out_u,
                      float* out_v);

/* a13-a15: AccelArray restated (object space, EPSILON 1e-5 as double, min_t 1e6). */
void orc_array_intersect(orc_scene* s, int n, const float* rays6, uint8_t* hit, orc_ray* records,
                         int* material, int* out_index, float* out_t);
void orc_array_occlude(orc_scene* s, int n, const float* rays6, uint8_t* occluded);

/* a16-a18: AccelSpheres restated. diffuse = the rand()%256 colours under srand(420). */
void orc_spheres_intersect(orc_scene* s, int n, const float* rays6, uint8_t* hit,
                           orc_ray* records, float* diffuse, int* out_index);
void orc_spheres_occlude(orc_scene* s, int n, const float* rays6, uint8_t* occluded);

/* a26: Camera::getScreenRays, data/camera.cpp:19-44 */
void orc_camera_rays(float fov, int w, int h, float* out_dir);

/* a21: G-buffer encode (prepass_fs.cpp:11-13 + GL UNORM16 / FP16 storage) and select/decode
 * (ras_rasterizer.cpp:509-536,556-561).  nrm_rgba: n x 4, pos_rgb: n x 3 (as read back). */
void orc_gbuffer_encode(int n, const float* normal3, const float* position3, const float* alpha,
                        float* nrm_rgba, float* pos_rgb);
int orc_gbuffer_select(int n, const float* nrm_rgba, const float* pos_rgb, int interlace,
                       int phase, float* out_pos, float* out_nrm, int* out_index);

/* a19: Gen-1 shade() at G-buffer texels (frame spec in DESIGN.md).  material_id: n ints (-1 = no
 * surface).  rgba: n x 4 bytes.  rays_issued (optional): secondary rays traced. */
void orc_render_texels(orc_scene* s, const orc_shade_config* cfg, int n, const float* pos,
                       const float* nrm, const int* material_id, uint8_t* rgba, float* rgb_float,
                       int64_t* rays_issued);

/* The same with per-texel texture coordinates (n x 2 floats, NULL = zeros): Texture::lookup
 * (data/data.cpp:35-57) replaces the diffuse colour of textured materials (rt_raytracer.cpp:93-95). */
void orc_render_texels_uv(orc_scene* s, const orc_shade_config* cfg, int n, const float* pos,
                          const float* nrm, const float* uv, const int* material_id, uint8_t* rgba,
                          float* rgb_float, int64_t* rays_issued);

/* a20: live shading equation (mesh_fs.cpp:26-49) for hit records; rgb out n x 3 floats (0..1). */
void orc_shade_records(orc_scene* s, int n, const orc_ray* records, const int* material_id,
                       float* rgb);

/* a22-a25: photon map.  rng: 0 = libc srand(seed)/rand() (the reference's stream),
 * 1 = counter RNG keyed by (seed, photon index) used by the GPU pass.  accel as geometry above. */
int orc_photon_generate(orc_scene* s, int count, int geometry, int rng, uint64_t seed);
int orc_photon_generate_range(orc_scene* s, int count, int geometry, uint64_t seed, int64_t first, int64_t n);
int orc_photon_count(orc_scene* s);
void orc_photon_get(orc_scene* s, float* out9);
void orc_photon_set(orc_scene* s, int n, const float* in9);
void orc_radiance_estimate(orc_scene* s, int npts, const float* pos, const float* nrm,
                           unsigned count, float radius, float* out);
void orc_photon_lookup(orc_scene* s, int npts, const float* pos, float radius, float* out);

#ifdef __cplusplus
}
#endif
#endif
