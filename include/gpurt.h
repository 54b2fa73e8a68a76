/* gpurt -- C-ABI of the MI355X (gfx950) LBVH-build + ray-trace path.
 *
 * This is the boundary a maintainer of the reference binds to in place of its
 * Direct3D 12 layers: device + command submission (src/GPUDevice.h, src/GPUScheduler.h),
 * compute pipeline states (src/PipelineState.h) and GPU resources
 * (src/ShaderResources.h). One launcher per compute shader takes the shader's
 * constant buffer as a POD struct (byte-identical to the reference's) and the
 * buffers the shader binds, in register order. Launches are asynchronous on the
 * device's stream; stream order replaces the reference's UAV barriers.
 *
 * Conventions: every function returns 0 on success or a non-zero error code
 * (RT_ERR_* below, or 0x1000 + hipError_t for a failed HIP call); nothing throws
 * or aborts. Handles are opaque. Host memory stays owned by the caller, device
 * memory by the buffer handle. All calls for one device must come from one
 * thread at a time (the reference drives its pipeline from one render thread,
 * src/Main.cpp:18-25); that thread need not be the creator.
 */
#ifndef GPURT_H
#define GPURT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK 0
#define RT_ERR_INVALID_ARGUMENT 1   /* null handle, size mismatch, count out of range */
#define RT_ERR_TOO_FEW_PRIMITIVES 2 /* fewer than 3 triangles: the reference's level loop never ends */
#define RT_ERR_BUFFER_TOO_SMALL 3
#define RT_ERR_OUT_OF_MEMORY 4
#define RT_ERR_LIMIT_EXCEEDED 5     /* beyond a fixed limit: of an entry point that allocates nothing, of the second hierarchy */
#define RT_ERR_RCCL_UNAVAILABLE 6   /* librccl.so.1 could not be loaded (multi-GPU entry points only) */
#define RT_ERR_HIP_BASE 0x1000      /* 0x1000 + hipError_t */
#define RT_ERR_RCCL_BASE 0x2000     /* 0x2000 + ncclResult_t */

typedef struct rt_device rt_device;
typedef struct rt_buffer rt_buffer;
typedef struct rt_event rt_event;
typedef struct rt_accel rt_accel;
typedef struct rt_comm rt_comm;

/* ---- element layouts (shader/Raytracer.hlsli:15-40, src/RaytracerMesh.h:14-46,
 *      shader/PerRayShading.hlsli:27-48, src/TextureAtlas.h:14-20) ---- */
typedef struct { float position[3]; float uv[2]; float normal[3]; float tangent[3]; uint32_t material_id; } rt_vertex; /* 48 B */
typedef struct { float origin[3]; float direction[3]; float t_min; float t_max; } rt_ray;                               /* 32 B */
typedef struct { float min[3]; float max[3]; uint32_t link[2]; } rt_aabb;                                               /* 32 B */
typedef struct {
    float albedo[3]; float roughness; float f0_color[3]; float metallic; float emissive[3];
    uint32_t albedo_texture, roughness_texture, f0_texture, metallic_texture, emissive_texture;
} rt_pbr_material;                                                                                                       /* 64 B */
typedef struct { uint32_t offset, width, height, row_pitch; } rt_texture_id;                                            /* 16 B */

/* ---- constant buffers (src/RaytracerPipeline.h:36-45, 94-105, 148-154, 192-199, 241-246) ---- */
typedef struct {
    float inverse_view[16];        /* as XMStoreFloat4x4 leaves it: inverse(transpose(view)) row-major */
    float inverse_projection[16];
    int32_t screen_size[2];
    float aa_sample_spread;
    float dof_sample_spread;
    uint32_t max_rays_per_pixel;
    uint32_t rng_seed[3];
} rt_camera_ray_gen_info;                                                                                                /* 160 B */
typedef struct { float scene_min[4]; float scene_max[4]; uint32_t num_primitives; uint32_t padding[3]; } rt_morton_code_info; /* 48 B */
typedef struct { uint32_t num_elements; uint32_t sort_pass_index; } rt_sort_info;                                       /* 8 B */
typedef struct { uint32_t num_children, previous_index, current_index, padding; } rt_bvh_info;                          /* 16 B */
typedef struct { int32_t screen_dimensions[2]; uint32_t num_indices, num_rays, max_rays_per_pixel; uint32_t rng_seed[3]; } rt_trace_rays_info; /* 32 B */
typedef struct { int32_t screen_dimensions[2]; uint32_t max_rays_per_pixel; uint32_t num_samples; } rt_generate_final_image_info;             /* 16 B */

/* Rows of the image whose rays a device holds (multi-GPU sharding; no counterpart in the reference, which
 * renders the whole image on one device): rows first_row, first_row + row_stride, ..., num_rows of them,
 * stored one after the other. {0, height, 1} is the whole image; a NULL pointer means the same. A stride
 * equal to the number of devices deals the rows out in turn, which balances the load better than bands. */
typedef struct { uint32_t first_row, num_rows, row_stride; } rt_image_rows;

/* ---- device, stream, events: replaces DX12Device (src/GPUDevice.h:17-38) and
 *      GPUScheduler::Record/Execute/Flush (src/GPUScheduler.cpp:130-166) ---- */
int rt_device_create(int gpu_index, rt_device** out);
int rt_device_destroy(rt_device* dev);
int rt_device_sync(rt_device* dev);                      /* GPUScheduler::Flush */
void* rt_device_stream(rt_device* dev);                  /* the hipStream_t launches go to */
int rt_device_set_stream(rt_device* dev, void* hip_stream); /* adopt a caller-owned stream (not destroyed here) */
int rt_device_index(rt_device* dev);
int rt_device_count(int* count);                         /* GPUs this process sees (the reference takes the first adapter, src/GPUDevice.cpp:40-91) */
const char* rt_error_string(int code);

/* Which traversal kernel rt_trace_rays / rt_trace_rays_accel launch (no counterpart in the reference).
 * QUAD, WAVE and SIMPLE walk the reference's hierarchy in the reference's order and return the reference's closest
 * hit by construction: QUAD two levels per step on 64-byte records, WAVE one level per step (also used for
 * hierarchies of fewer than three levels and while counters are on), SIMPLE one thread per ray through traversal
 * and shading. SPATIAL (default; rt_trace_rays_accel only, elsewhere it means QUAD) finds the same closest hit on a
 * second hierarchy that rt_accel_build makes from non-overlapping Morton codes, several times faster; rays for which
 * the reference's order could decide are walked again by QUAD. It can differ from the reference only in hits that are
 * numerically false: a near-degenerate triangle whose hit test accepts a ray that passes beside the triangle's own
 * (slightly widened) box, or far enough in front of it - see the kernel's description; comparisons of the two walks
 * over random scenes put that below one ray in 10^9. Whether rt_accel_build makes the second hierarchy is a property of
 * the accel (rt_accel_set_second_hierarchy, on by default); an accel built without it is traced by QUAD. Triangles whose
 * box is a large part of the scene's (walls, a floor: up to 64 per build) are left out of that second hierarchy, in which
 * they would make every node above them as large as themselves, and are tested from a list before a ray's walk starts.
 * Instrumentation and test-only switches of these kernels are declared in gpurt_debug.h. The environment variable GPURT_TRACE_KERNEL = spatial | quad | wave | simple sets the initial choice of a new device. */
#define RT_TRACE_KERNEL_QUAD 0
#define RT_TRACE_KERNEL_WAVE 1
#define RT_TRACE_KERNEL_SIMPLE 2
#define RT_TRACE_KERNEL_SPATIAL 3
int rt_device_set_trace_kernel(rt_device* dev, int kernel);
/* Dropping missed rays between bounces (no counterpart call in the reference; its author's notes on re-binning rays are
 * at shader/CS_Intersection.hlsl:20-37). In the reference a ray that misses is left untouched (shader/CS_TraceRays.hlsl:248,
 * 290-291) and is traced again, unchanged, by every later bounce of the sample cycle - to miss again. While enabled, the
 * shading step marks such rays and later trace calls leave them out (their place in the ray order goes to the end and
 * the traversal launches over the live rays only) until new camera rays are generated or a hierarchy is built on this
 * device. For a scene that does not change inside a sample cycle every buffer stays bit-identical to the default mode;
 * the difference is work, not results. The marks follow the device's own calls: after changing the ray buffers by other
 * means (rt_buffer_upload), call this function again to clear them. Default: off, the reference's re-tracing. */
int rt_device_set_drop_missed_rays(rt_device* dev, int enable);
/* The slab test as the reference writes it (no counterpart call in the reference; it selects the reference's own
 * arithmetic): shader/CS_TraceRays.hlsl:83-84 divides by the ray direction for every box, the kernels above multiply by
 * one reciprocal per ray and axis, which rounds differently in the last place and lets a few rays per million enter or
 * miss a box the other form does not. While enabled, trace calls with a hierarchy run the one-level walk (WAVE, or
 * SIMPLE if that is selected) with the division: the reference's result as written, at a fraction of the speed. */
int rt_device_set_slab_division(rt_device* dev, int enable);
/* Tuning of the persistent traversal kernels (no counterpart in the reference; no setting changes a result). A new
 * device starts from the defaults, or from the environment variables named beside the fields, read once by
 * rt_device_create. 0 = the library's default where stated. */
typedef struct {
    uint32_t leaf_threshold;    /* GPURT_LEAF_THRESHOLD: lanes with a leaf waiting that start a triangle round; 0 = 32 (SPATIAL, where it is
                                   meant of 64 lanes and scaled to the lanes of the wave that have a ray) / 16 */
    uint32_t refill_threshold;  /* GPURT_REFILL_THRESHOLD: free lanes that start a refill; 0 = 8 (SPATIAL) / 16 */
    uint32_t blocks_per_cu;     /* GPURT_BLOCKS_PER_CU: persistent workgroups per compute unit; default 4 */
    uint32_t sort_rays;         /* GPURT_SORT_RAYS: 1 = order the rays of a launch by estimated walk length, 0 = as they come,
                                   2 (default) = order them for the walks in the reference's order, not for the default kernel (its walks are short and alike: measured, the order
                                   costs what it gains) */
    uint32_t deal_shift;        /* GPURT_DEAL_SHIFT: log2 of the block of ray-order positions dealt to one queue; 6 .. 20, default 6 */
    uint32_t cost_bits;         /* GPURT_COST_BITS: bits of the walk-length class in the ray key; 6 .. 12, default 6 */
    uint32_t cost_depth;        /* GPURT_COST_DEPTH: tree depth probed for the walk-length estimate; 0 = default (no probe on SPATIAL) */
    uint32_t cost_stride;       /* GPURT_COST_STRIDE: every n-th node pair of that depth is probed; 0 = default */
    uint32_t cost_chord;        /* GPURT_COST_CHORD: 1 = estimate from the ray's way through the trunk's box, 0 = probe, 2 = default */
    uint32_t key_passes;        /* GPURT_KEY_PASSES: 8-bit sorting passes over the ray keys, 1 .. 3; 0 = default */
    uint32_t stuck_threshold;   /* GPURT_STUCK_THRESHOLD: SPATIAL only - lanes whose walk is over but whose leaves still wait that
                                   start a triangle round before leaf_threshold is reached; 0 = 4 */
    uint32_t retire_us;         /* GPURT_RETIRE_US: SPATIAL only - the end of a launch: the persistent workgroups stop taking rays in
                                   four turns, a quarter of them each time the ray queues are another retire_us microseconds (at
                                   the pace so far) from empty, so that the last rays are walked by one wave per SIMD instead of
                                   by four that share its issue slots; 0 = all stay to the end, default 40, at most 1000 */
} rt_trace_tuning;
int rt_device_get_trace_tuning(rt_device* dev, rt_trace_tuning* out);
int rt_device_set_trace_tuning(rt_device* dev, const rt_trace_tuning* tuning);   /* out-of-range values are clamped */

int rt_event_create(rt_device* dev, rt_event** out);
int rt_event_record(rt_event* ev);                       /* on the device's stream */
int rt_event_elapsed_ms(rt_event* start, rt_event* stop, float* ms); /* synchronises on stop */
int rt_event_destroy(rt_event* ev);

/* ---- buffers: replaces StructuredBuffer / RWStructuredBuffer / UploadBuffer /
 *      ReadbackBuffer (src/ShaderResources.h). New buffers are zero-filled, as the
 *      reference's committed resources are (src/ShaderResources.h:2297); its result and
 *      light buffers rely on that. ---- */
int rt_buffer_alloc(rt_device* dev, size_t bytes, rt_buffer** out);
int rt_buffer_wrap(rt_device* dev, void* device_ptr, size_t bytes, rt_buffer** out); /* non-owning view of caller memory */
int rt_buffer_upload(rt_buffer* buf, const void* src, size_t bytes, size_t dst_offset);
int rt_buffer_readback(rt_buffer* buf, void* dst, size_t bytes, size_t src_offset);  /* synchronises */
int rt_buffer_clear(rt_buffer* buf);
int rt_buffer_free(rt_buffer* buf);
void* rt_buffer_ptr(rt_buffer* buf);
size_t rt_buffer_size(rt_buffer* buf);

/* ---- one launcher per compute shader, reference buffer layouts ---- */

/* shader/CS_GenerateMortonCodes.hlsl (dispatch: src/RaytracerPipeline.cpp:284-303).
 * indices: uint32[3N]; vertices: rt_vertex[]; morton: uint4[N] <- (code, 3*tri, 0, 0);
 * code_frequencies: uint4[256], .x zeroed. */
int rt_generate_morton_codes(rt_device* dev, const rt_morton_code_info* info, const rt_buffer* indices,
                             const rt_buffer* vertices, rt_buffer* morton_codes, rt_buffer* code_frequencies);
/* shader/CS_SortCount.hlsl (src/RaytracerPipeline.cpp:313-337): temp[i].xy = morton[i].xy;
 * code_frequencies[d].x += number of keys whose digit (key >> 8*pass) & 0xff equals d. */
int rt_sort_count(rt_device* dev, const rt_sort_info* info, const rt_buffer* morton_codes, rt_buffer* temp_morton_codes,
                  rt_buffer* code_frequencies);
/* shader/CS_SortPrefixSum.hlsl (src/RaytracerPipeline.cpp:339-354): .y = exclusive prefix sum of .x */
int rt_sort_prefix_sum(rt_device* dev, const rt_sort_info* info, rt_buffer* code_frequencies);
/* shader/CS_SortSort.hlsl (src/RaytracerPipeline.cpp:356-379): morton[].xy = temp[].xy stably partitioned
 * by the pass's digit, starting at code_frequencies[d].y; .x of every counter is zeroed.
 * Must follow rt_sort_count and rt_sort_prefix_sum of the same pass on the same buffers (the
 * per-tile digit counts of the count step are kept in the device's workspace). */
int rt_sort_sort(rt_device* dev, const rt_sort_info* info, rt_buffer* morton_codes, const rt_buffer* temp_morton_codes,
                 rt_buffer* code_frequencies);
/* shader/CS_BVHBuildLeaves.hlsl (src/RaytracerPipeline.cpp:483-503): info->num_children = N.
 * Writes nodes 1 .. ceil(N/2). */
int rt_bvh_build_leaves(rt_device* dev, const rt_bvh_info* info, const rt_buffer* indices, const rt_buffer* vertices,
                        const rt_buffer* morton_codes, rt_buffer* bvh);
/* shader/CS_BVHBuild.hlsl (src/RaytracerPipeline.cpp:507-559): one level. */
int rt_bvh_build(rt_device* dev, const rt_bvh_info* info, rt_buffer* bvh);
/* shader/CS_CameraRayGeneration.hlsl (src/RaytracerPipeline.cpp:105-150). ray_pixels: uint32 per ray
 * ((x << 16) | y), the packing of the reference's uint4 array read as a flat uint array.
 * rows restricts generation to the image rows this device holds (NULL: the whole image); pixel
 * coordinates and random seeds stay those of the whole image. */
int rt_camera_ray_generation(rt_device* dev, const rt_camera_ray_gen_info* info, rt_buffer* rays, rt_buffer* old_rays,
                             rt_buffer* ray_pixels, const rt_image_rows* rows);
/* shader/CS_TraceRays.hlsl + PerRayShading.hlsli + Random.hlsli (src/RaytracerPipeline.cpp:687-729).
 * use_bvh mirrors RT_USE_BVH (src/Settings.h:13). The hierarchy must have the layout the build
 * launchers above emit for info->num_indices / 3 primitives. atlas: uint4[atlas_elements].
 * rows names the image rows the per-ray buffers hold (NULL: the whole image, info->num_rays =
 * width * height * rays per pixel; otherwise info->num_rays = rows->num_rows * width * rays per pixel).
 * The per-thread random seed uses the ray's index in the whole image, so that any split of the rows
 * reproduces the single-device image. */
int rt_trace_rays(rt_device* dev, const rt_trace_rays_info* info, int use_bvh, const rt_buffer* indices,
                  const rt_buffer* vertices, const rt_buffer* materials, const rt_buffer* texture_ids,
                  const rt_buffer* texture_atlas, uint32_t atlas_elements, rt_buffer* rays, rt_buffer* old_rays,
                  const rt_buffer* ray_pixels, rt_buffer* scattered_light, rt_buffer* emitted_light,
                  const rt_buffer* bvh, const rt_image_rows* rows);
/* shader/CS_GenerateFinalImage.hlsl (src/RaytracerPipeline.cpp:809-854). result: float4[W*H] running
 * mean; output: float4[W*H] tone-mapped (the reference's RGBA16F texture, kept in binary32);
 * output_half: optional uint16[4*W*H] binary16 copy, may be NULL. info->screen_dimensions[1] is
 * the number of rows held by the buffers. */
int rt_generate_final_image(rt_device* dev, const rt_generate_final_image_info* info, const rt_buffer* emitted_light,
                            rt_buffer* result, rt_buffer* output, rt_buffer* output_half);

/* ---- fused per-frame LBVH build (no counterpart dispatch in the reference: it replaces the
 *      1 + 12 + ~log2(N) dependent dispatches of SortPrimitives::Sort and BuildBVH::Build,
 *      src/RaytracerPipeline.cpp:276-383, 475-562, with identical results) ---- */
uint32_t rt_bvh_node_count(uint32_t num_primitives);     /* nodes the hierarchy uses; 0 if < 3 primitives */
/* Allocates every workspace once: packed triangles, sort keys, the private copies of the reference's hierarchy and,
 * up to 2^26 leaf slots (134 M triangles; up to 67 M the hierarchy has twice as many slots as leaves), the library's own
 * hierarchy for RT_TRACE_KERNEL_SPATIAL - about 1 KB per triangle in all. */
int rt_accel_create(rt_device* dev, uint32_t max_primitives, rt_accel** out);
int rt_accel_destroy(rt_accel* accel);
/* Whether builds of this accel also make the library's own hierarchy (what RT_TRACE_KERNEL_SPATIAL walks; on by
 * default where it fits), and the finest Morton cells its leaf slots may be aligned to: cell_bits in 0, 3, 6, .. 21
 * (0 = leaves by rank only; anything above 21 means 21, the default). Takes effect at the next build; an accel whose last build did not make it is traced by
 * RT_TRACE_KERNEL_QUAD. Enabling it on an accel that is too large for it (more than 2^26 leaf slots, see rt_accel_create)
 * returns RT_ERR_LIMIT_EXCEEDED and leaves it off: such an accel works, traced by the much slower RT_TRACE_KERNEL_QUAD, and
 * a caller that asks learns of it here rather than from the frame time. */
int rt_accel_set_second_hierarchy(rt_accel* accel, int enable, uint32_t cell_bits);
/* Gathers Indices -> Vertices -> positions into the accel's packed triangle array
 * (3 x float4 per triangle, original order). Call after the mesh changes. */
int rt_accel_pack(rt_accel* accel, uint32_t num_primitives, const rt_buffer* indices, const rt_buffer* vertices);
/* Centroid bounds of a mesh: the scene AABB the reference's loader computes on the host
 * (src/RaytracerMesh.cpp:342-364: min / max over ((p1 + p2) + p3) * (1/3) of every triangle), for meshes
 * that change on the device. bounds receives float[8] = (min.xyz, 0, max.xyz, 0), the head of
 * rt_morton_code_info. Positions must be finite (the host loop's SSE min / max depend on the order of
 * the triangles once a NaN is involved; this reduction does not keep that order). Zero triangles give
 * +-FLT_MAX as the host loop does. */
int rt_scene_bounds(rt_device* dev, uint32_t num_primitives, const rt_buffer* indices, const rt_buffer* vertices,
                    rt_buffer* bounds);
/* rt_scene_bounds for callers that cannot allocate: code under HIP graph capture, tight per-frame loops, harnesses
 * that forbid hipMalloc. No counterpart in the reference. Same arguments, same result bit for bit, but the reduction's
 * scratch (per-workgroup partial bounds, 24 KiB) is a static device array that every GPU the library runs on holds
 * until the process ends, so the call never allocates or frees (rt_scene_bounds allocates its scratch on a device's
 * first call). Limits: num_primitives <= RT_SCENE_BOUNDS_STATIC_MAX_PRIMITIVES and a GPU index below 64, else
 * RT_ERR_LIMIT_EXCEEDED. All calls on one GPU share that GPU's array: calls that overlap on one GPU (two streams, two
 * device handles of the same GPU) race; calls on one stream are ordered and need no clearing in between. */
#define RT_SCENE_BOUNDS_STATIC_MAX_PRIMITIVES 0x55555555u   /* 3 * count must fit the 32-bit index arithmetic */
int rt_scene_bounds_static(rt_device* dev, uint32_t num_primitives, const rt_buffer* indices, const rt_buffer* vertices,
                           rt_buffer* bounds);
/* rt_accel_pack and rt_scene_bounds in one pass over the mesh. */
int rt_accel_pack_bounds(rt_accel* accel, uint32_t num_primitives, const rt_buffer* indices, const rt_buffer* vertices,
                         rt_buffer* bounds);
/* Morton codes, stable sort, leaves and all levels. bvh receives rt_bvh_node_count(N) nodes, bit-identical
 * to the per-shader launchers; morton_codes (optional, uint4[N]) receives the sorted (code, 3*tri, 0, 0). */
int rt_accel_build(rt_accel* accel, const float scene_min[3], const float scene_max[3], rt_buffer* bvh,
                   rt_buffer* morton_codes);
/* rt_accel_build with the scene bounds read from device memory (rt_scene_bounds' layout): a per-frame
 * rebuild of a moving mesh without a round trip to the host. */
int rt_accel_build_device_bounds(rt_accel* accel, const rt_buffer* bounds, rt_buffer* bvh, rt_buffer* morton_codes);
/* Refit (no counterpart in the reference, which sorts again on every build): leaves and levels once more over the
 * vertices packed since (rt_accel_pack / rt_accel_pack_bounds of the same mesh after it has moved), in the sorted order
 * of the accel's last rt_accel_build - i.e. CS_BVHBuildLeaves and CS_BVHBuild run over the Morton records of that build.
 * bvh receives the nodes (same count, same links, new boxes); the library's own hierarchy, if the build made one, is
 * refitted as well. Skips the Morton codes, both sorts and the cell layout; the hierarchy loosens as triangles move
 * away from where they were sorted, so rebuild now and then. RT_ERR_INVALID_ARGUMENT without a build to refit. */
int rt_accel_refit(rt_accel* accel, rt_buffer* bvh);
/* Workload helper (no counterpart in the reference, whose scenes are static): the synthetic dynamic scene.
 * Triangle i of base_vertices (vertices 3i .. 3i + 2, 48-byte rt_vertex) is written to vertices moved as a
 * whole by d = 0.25 * sin(2 pi * frac(frame / 32 + 0.618 * i)) along each axis; the other vertex fields are
 * copied. */
int rt_synthetic_animate(rt_device* dev, uint32_t num_primitives, uint32_t frame, const rt_buffer* base_vertices,
                         rt_buffer* vertices);
/* rt_trace_rays reading the accel's Morton-ordered packed triangles instead of Indices -> Vertices.
 * State kept between calls on a device (none of it enters a result): with the SPATIAL kernel a call also leaves the
 * sort keys of the rays it scattered for the next call - used there if that call traces the same ray buffer with the
 * same accel and settings, made afresh otherwise; rt_camera_ray_generation and builds discard them. Keys only decide the
 * order rays are traced in. */
int rt_trace_rays_accel(rt_device* dev, const rt_trace_rays_info* info, const rt_accel* accel, const rt_buffer* indices,
                        const rt_buffer* vertices, const rt_buffer* materials, const rt_buffer* texture_ids,
                        const rt_buffer* texture_atlas, uint32_t atlas_elements, rt_buffer* rays, rt_buffer* old_rays,
                        const rt_buffer* ray_pixels, rt_buffer* scattered_light, rt_buffer* emitted_light,
                        const rt_buffer* bvh, const rt_image_rows* rows);

/* ---- multi-GPU (no counterpart in the reference, which renders on one device): one process or host thread per GPU,
 *      the image rows dealt to the devices in turn, scene and hierarchy replicated and rebuilt per device, and ONE
 *      exchange per finished frame - the rows gathered to a root device over RCCL (xGMI inside a node). RCCL is loaded
 *      at run time by the first of these calls; nothing else in the library needs it. ---- */
/* The rows rank `rank` of `world` holds: rank, rank + world, ... ({first_row, num_rows, row_stride = world}; pass it to
 * rt_camera_ray_generation / rt_trace_rays). A rank beyond the image height holds no row. */
int rt_rows_for_rank(uint32_t height, uint32_t world, uint32_t rank, rt_image_rows* rows);
/* gathered: float4 per pixel, the ranks' rows one rank after the other (what a gather of the per-rank row buffers
 * yields); image: the same rows in image order. Not in place. */
int rt_rows_interleave(rt_device* dev, const rt_buffer* gathered, uint32_t width, uint32_t height, uint32_t world, rt_buffer* image);
#define RT_COMM_ID_BYTES 128
/* An id for a new group of devices (ncclGetUniqueId): made once, by any one rank, and handed to every rank by the
 * caller's own means (shared memory between threads, a file, the launcher's store). */
int rt_comm_unique_id(void* unique_id /* RT_COMM_ID_BYTES */);
/* Joins the group as rank `rank` of `world` with the device's GPU (ncclCommInitRank): collective, returns once every
 * rank has called. One GPU can be one rank of a group only. */
int rt_comm_init(rt_device* dev, int rank, int world, const void* unique_id, rt_comm** out);
int rt_comm_destroy(rt_comm* comm);
int rt_comm_rank(const rt_comm* comm);
int rt_comm_world(const rt_comm* comm);
/* Collective, queued on every rank's device stream: each rank's finished rows (float4 per pixel, the rows of
 * rt_rows_for_rank one after the other - e.g. the output of rt_generate_final_image) travel to rank `root`, whose
 * `image` (float4[width * height]; ignored elsewhere, may be NULL) receives them in image order. The root keeps a
 * staging buffer of the image's size, allocated on the first call. */
int rt_gather_rows(rt_comm* comm, const rt_buffer* rows, uint32_t width, uint32_t height, int root, rt_buffer* image);

#ifdef __cplusplus
}
#endif
#endif
