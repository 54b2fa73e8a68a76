/* CPU oracle for the LBVH-build + ray-trace path: public C interface.
 *
 * TEST INFRASTRUCTURE ONLY. This is a scalar (OpenMP over the GPU thread index)
 * restatement of the reference's HLSL compute kernels and of the host code that
 * sequences them. Only tests, the smoke check and the CPU-baseline leg of the
 * benchmark may load it; the product library never links or calls it.
 *
 * Parity status: PARITY UNPINNED by reference tests. The reference ships no
 * tests, golden vectors or CPU implementation, and its HLSL/D3D12 path cannot be
 * compiled on Linux; the only pins are known-answer values derived from the
 * shader sources (xorshift32 sequence, per-thread seed formula, LeftShift3 /
 * Morton interleave, node-count closed forms) and the structural invariant
 * "sorted array == stable sort by code".
 *
 * Every struct below has the byte layout of the reference's buffer element or
 * constant buffer of the same name.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* shader/Raytracer.hlsli:25-32, src/RaytracerMesh.h:24-31 (48 bytes) */
typedef struct { float px, py, pz, u, v, nx, ny, nz, tx, ty, tz; uint32_t mat; } orc_vertex;
/* shader/Raytracer.hlsli:15-21 (32 bytes) */
typedef struct { float ox, oy, oz, dx, dy, dz, tmin, tmax; } orc_ray;
/* shader/Raytracer.hlsli:35-40 (32 bytes) */
typedef struct { float mn[3]; float mx[3]; uint32_t link[2]; } orc_aabb;
typedef struct { uint32_t x, y, z, w; } orc_uint4;
typedef struct { float x, y, z, w; } orc_float4;
/* shader/PerRayShading.hlsli:27-40, src/RaytracerMesh.h:33-46 (64 bytes) */
typedef struct {
    float albedo[3]; float roughness; float f0[3]; float metallic; float emissive[3];
    uint32_t tex_albedo, tex_roughness, tex_f0, tex_metallic, tex_emissive;
} orc_material;
/* shader/PerRayShading.hlsli:42-48, src/TextureAtlas.h:14-20 */
typedef struct { uint32_t offset, width, height, row_pitch; } orc_texture_id;

/* src/RaytracerPipeline.h:36-45 (160 bytes). The two matrices hold the 16 floats
 * exactly as the host stores them (XMStoreFloat4x4 of inverse(transpose(M))). */
typedef struct {
    float inverse_view[16]; float inverse_projection[16];
    int32_t screen[2]; float aa_spread; float dof_spread;
    uint32_t max_rays_per_pixel; uint32_t rng_seed[3];
} orc_raygen_info;
/* src/RaytracerPipeline.h:94-100 (48 bytes) */
typedef struct { float scene_min[4]; float scene_max[4]; uint32_t num_primitives; uint32_t pad[3]; } orc_morton_info;
/* src/RaytracerPipeline.h:101-105 */
typedef struct { uint32_t num_elements; uint32_t pass_index; } orc_sort_info;
/* src/RaytracerPipeline.h:148-154 */
typedef struct { uint32_t num_children, previous_index, current_index, pad; } orc_bvh_info;
/* src/RaytracerPipeline.h:192-199 (32 bytes) */
typedef struct { int32_t screen[2]; uint32_t num_indices, num_rays, max_rays_per_pixel; uint32_t rng_seed[3]; } orc_trace_info;
/* src/RaytracerPipeline.h:241-246 */
typedef struct { int32_t screen[2]; uint32_t max_rays_per_pixel; uint32_t num_samples; } orc_image_info;

/* per-dispatch traversal statistics (sums over all rays of one trace dispatch) */
typedef struct { uint64_t nodes_visited, tris_tested, rays_hit, rays, max_stack; } orc_trace_stats;

/* ---- known-answer helpers ---- */
uint32_t orc_xorshift(uint32_t seed);                 /* shader/Random.hlsli:3-9 */
uint32_t orc_thread_seed(uint32_t thread_id);         /* shader/CS_TraceRays.hlsl:255-257 */
uint32_t orc_left_shift3(uint32_t v);                 /* shader/CS_GenerateMortonCodes.hlsl:30-38 */
uint32_t orc_morton(uint32_t x, uint32_t y, uint32_t z);
float orc_random_float(uint32_t* seed);               /* shader/Random.hlsli:63-66 */
void orc_sincos(float x, float* s, float* c);

/* ---- one function per HLSL main; one loop iteration per GPU thread ---- */
void orc_generate_morton_codes(const orc_morton_info* info, const uint32_t* indices, const orc_vertex* vertices,
                               orc_uint4* morton, orc_uint4* freq);
void orc_sort_count(const orc_sort_info* info, const orc_uint4* morton, orc_uint4* temp, orc_uint4* freq);
void orc_sort_prefix_sum(const orc_sort_info* info, orc_uint4* freq);
void orc_sort_sort(const orc_sort_info* info, orc_uint4* morton, const orc_uint4* temp, orc_uint4* freq);
void orc_bvh_build_leaves(const orc_bvh_info* info, const uint32_t* indices, const orc_vertex* vertices,
                          const orc_uint4* morton, orc_aabb* bvh);
void orc_bvh_build(const orc_bvh_info* info, orc_aabb* bvh);
void orc_camera_ray_generation(const orc_raygen_info* info, orc_ray* rays, orc_ray* old_rays, uint32_t* ray_pixels);
/* true_division != 0 evaluates the slab test with IEEE division by the direction
 * as the HLSL is written; 0 multiplies by one IEEE reciprocal per axis per ray. */
void orc_trace_rays(const orc_trace_info* info, int use_bvh, int true_division,
                    const uint32_t* indices, const orc_vertex* vertices,
                    const orc_material* materials, const orc_texture_id* texture_ids,
                    const orc_uint4* atlas, uint32_t atlas_elements,
                    orc_ray* rays, orc_ray* old_rays, const uint32_t* ray_pixels,
                    orc_float4* scattered, orc_float4* emitted, const orc_aabb* bvh,
                    orc_trace_stats* stats);
void orc_generate_final_image(const orc_image_info* info, const orc_float4* emitted, orc_float4* result,
                              orc_float4* output);

/* ---- host-side sequencing (src/RaytracerPipeline.cpp) ---- */
/* SortPrimitives::Sort: Morton codes then 4 x (count, prefix sum, scatter). as_written != 0
 * runs the serial-per-digit scatter; 0 replaces each pass by std::stable_sort on the digit. */
void orc_sort_primitives(const float scene_min[3], const float scene_max[3], uint32_t num_primitives,
                         const uint32_t* indices, const orc_vertex* vertices,
                         orc_uint4* morton, orc_uint4* temp, orc_uint4* freq, int as_written);
/* BuildBVH::Build level schedule. Returns the number of nodes written (root at 0), or 0 if
 * num_primitives < 3 (the reference's loop never terminates there). */
uint32_t orc_build_bvh(uint32_t num_primitives, const uint32_t* indices, const orc_vertex* vertices,
                       const orc_uint4* morton, orc_aabb* bvh);
uint32_t orc_bvh_node_count(uint32_t num_primitives);
/* CameraRayGen::Initialize matrix derivation (src/RaytracerPipeline.cpp:81-89) */
void orc_camera_matrices(const float position[3], const float focus[3], const float up[3],
                         float fov, float aspect, float near_z, float far_z,
                         float inverse_view[16], float inverse_projection[16]);
/* centroid scene bounds as LoadMeshFromFile computes them (src/RaytracerMesh.cpp:342-364) */
void orc_scene_aabb(uint64_t index_count, const uint32_t* indices, const orc_vertex* vertices,
                    float scene_min[3], float scene_max[3]);
/* Synthetic dynamic workload (not in the reference): triangle i moves as a whole by
 * 0.25 * sin(2 pi * frac(frame / 32 + 0.618 * i)) along each axis. CPU twin of the product's kernel. */
void orc_synthetic_animate(uint32_t num_primitives, uint32_t frame, const orc_vertex* base, orc_vertex* out);
/* CalculateTangents (src/RaytracerMesh.cpp:99-155) */
void orc_calculate_tangents(uint64_t index_count, const uint32_t* indices, orc_vertex* vertices);
/* CalculateNormals (src/RaytracerMesh.cpp:59-97) */
void orc_calculate_normals(uint64_t index_count, const uint32_t* indices, orc_vertex* vertices);
/* std::mt19937 draw sequence used for the host seeds (src/RaytracerPipeline.cpp:113-115,695-697) */
void orc_mt19937_draw(uint32_t seed, uint32_t skip, uint32_t count, uint32_t* out);
float orc_roughness_from_mtl(float shininess, float pr);   /* src/RaytracerMesh.cpp:290-299 */

/* ---- whole pipeline (RaytracerPipeline::Initialize / Render) ---- */
typedef struct {
    int32_t width, height;
    uint32_t max_rays_per_pixel, max_ray_depth;
    float aa_spread, dof_spread;
    int32_t use_bvh;
    float fov, near_z, far_z;
    float cam_position[3], cam_focus[3], cam_up[3];
    uint32_t seed_raygen, seed_trace;   /* mt19937 seeds (the reference uses random_device) */
    int32_t true_division;
    int32_t rebuild_every_cycle;        /* 0: build once as the reference does; 1: rebuild at every sample cycle; 2: build once, then
                                           leaves and levels again over the kept order at every sample cycle (refit) */
} orc_settings;

typedef struct orc_pipeline orc_pipeline;
orc_pipeline* orc_pipeline_create(const orc_settings* s, uint64_t index_count, const uint32_t* indices,
                                  uint64_t vertex_count, const orc_vertex* vertices,
                                  uint64_t material_count, const orc_material* materials,
                                  uint32_t texture_count, const orc_texture_id* texture_ids,
                                  const orc_uint4* atlas, uint32_t atlas_elements,
                                  const float scene_min[3], const float scene_max[3]);
/* replaces the vertex data (same count) and recomputes the scene bounds; 0 on a count mismatch */
int orc_pipeline_set_vertices(orc_pipeline* p, uint64_t vertex_count, const orc_vertex* vertices);
int orc_pipeline_render(orc_pipeline* p);              /* one Render() call = one bounce */
const orc_float4* orc_pipeline_output(orc_pipeline* p); /* tone-mapped image, width*height */
const orc_float4* orc_pipeline_result(orc_pipeline* p); /* accumulated ResultBuffer */
const orc_aabb* orc_pipeline_bvh(orc_pipeline* p);
const orc_uint4* orc_pipeline_morton(orc_pipeline* p);
const orc_ray* orc_pipeline_rays(orc_pipeline* p);
const orc_float4* orc_pipeline_emitted(orc_pipeline* p);
const orc_float4* orc_pipeline_scattered(orc_pipeline* p);
void orc_pipeline_stats(orc_pipeline* p, orc_trace_stats* last, orc_trace_stats* total);
void orc_pipeline_destroy(orc_pipeline* p);

int orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
