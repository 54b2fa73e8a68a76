// Camera-ray generation and final-image accumulation for gfx950.
//
// Semantics follow the reference's shader/CS_CameraRayGeneration.hlsl and CS_GenerateFinalImage.hlsl
// (dispatched by src/RaytracerPipeline.cpp:105-150 and 809-854), including their quirks: the
// ray-generation seed depends on pixel x only, both image axes are flipped and no half-pixel offset
// is applied, TMax is the far point's distance from the world origin, and the running mean starts at
// sample count 2. Both kernels are bandwidth-bound streaming passes (68 bytes written per ray;
// 16 * rays-per-pixel + 32 (+8) bytes per pixel).

#include "rt_internal.h"
#include "rt_math.h"

namespace {

using namespace rtm;

struct RaygenParams {
    float ivp[16];            // mul(InverseProjection, InverseView), row-major [i][j]
    float cam_add[3];         // mul(InverseView, (0,0,0,1)).xyz
    int32_t width, height;
    float aa, dof;
    uint32_t rpp;
    uint32_t seed[3];
    uint32_t first_row, num_rows, row_stride;      // the image rows held by the buffers
};

__device__ __forceinline__ float dot4(float a0, float a1, float a2, float a3, float b0, float b1, float b2, float b3) {
    return ((a0 * b0 + a1 * b1) + a2 * b2) + a3 * b3;
}

// shader/CS_CameraRayGeneration.hlsl:56-101; one thread per pixel, 64x4 pixel workgroups
__global__ __launch_bounds__(256) void k_raygen(RaygenParams p, float4* __restrict__ rays, float4* __restrict__ old_rays,
                                                uint32_t* __restrict__ ray_pixels) {
    const uint32_t x = blockIdx.x * 64u + threadIdx.x;
    const uint32_t row = blockIdx.y * 4u + threadIdx.y;
    if (x >= (uint32_t)p.width || row >= p.num_rows) return;
    const uint32_t y = p.first_row + row * p.row_stride;
    uint32_t s0 = x * (x + 17u);
    xorshift(s0);
    uint32_t sx = s0 + p.seed[0], sy = s0 + p.seed[1];
    const float isx = 1.0f / (float)p.width, isy = 1.0f / (float)p.height;
    for (uint32_t i = 0; i < p.rpp; i++) {
        float scx = (float)x * isx, scy = (float)y * isy;
        float ndx = -2.0f * scx + 1.0f, ndy = -2.0f * scy + 1.0f;
        float r1x = rnd(sx), r1y = rnd(sy);
        float nnx = ndx + ((r1x - 0.5f) * isx) * p.dof;
        float nny = ndy + ((r1y - 0.5f) * isy) * p.dof;
        float r2x = rnd(sx), r2y = rnd(sy);
        float fnx = ndx + ((r2x - 0.5f) * isx) * p.aa;
        float fny = ndy + ((r2y - 0.5f) * isy) * p.aa;
        float np[4], fp[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            np[j] = dot4(nnx, nny, 0.0f, 1.0f, p.ivp[j], p.ivp[4 + j], p.ivp[8 + j], p.ivp[12 + j]);
            fp[j] = dot4(fnx, fny, 1.0f, 1.0f, p.ivp[j], p.ivp[4 + j], p.ivp[8 + j], p.ivp[12 + j]);
        }
        f3 nearp = mk(np[0] / np[3], np[1] / np[3], np[2] / np[3]);
        f3 farp = mk(fp[0] / fp[3], fp[1] / fp[3], fp[2] / fp[3]);
        f3 dir = normalize3(sub(farp, nearp));
        f3 org = add(nearp, mk(p.cam_add[0], p.cam_add[1], p.cam_add[2]));
        const size_t fi = ((size_t)row * (uint32_t)p.width + x) * p.rpp + i;
        rays[2 * fi] = make_float4(org.x, org.y, org.z, dir.x);
        rays[2 * fi + 1] = make_float4(dir.y, dir.z, 0.0f, sqrtf(dot3(farp, farp)));
        old_rays[2 * fi] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        old_rays[2 * fi + 1] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(i));
        ray_pixels[fi] = (x << 16) | (y & 0xffffu);
    }
}

__device__ __forceinline__ uint16_t f32_to_f16_rne(float f) {
    _Float16 h = (_Float16)f;
    return __builtin_bit_cast(uint16_t, h);
}

// shader/CS_GenerateFinalImage.hlsl:28-60; one thread per pixel
__global__ __launch_bounds__(256) void k_final_image(uint32_t pixels, uint32_t rpp, uint32_t num_samples,
                                                     const float4* __restrict__ emitted, float4* __restrict__ result,
                                                     float4* __restrict__ output, uint2* __restrict__ output_half) {
    const uint32_t pidx = blockIdx.x * 256u + threadIdx.x;
    if (pidx >= pixels) return;
    float4 total = result[pidx];
    float nc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (uint32_t i = 0; i < rpp; i++) {
        float4 e = emitted[(size_t)pidx * rpp + i];
        nc[0] += e.x; nc[1] += e.y; nc[2] += e.z; nc[3] += e.w;
    }
    const uint32_t ns = num_samples & 0x7fffffffu;
    const float inv = 1.0f / (float)ns;
    const float keep = (float)(ns - 1u) * inv;
    float tc[4] = {total.x, total.y, total.z, total.w}, dc[4], oc[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        nc[k] = nc[k] / (float)rpp;
        dc[k] = keep * tc[k] + inv * nc[k];
        oc[k] = dc[k] / (1.0f + dc[k]);
    }
    if (num_samples & 0x80000000u) total = make_float4(dc[0], dc[1], dc[2], dc[3]);
    output[pidx] = make_float4(oc[0], oc[1], oc[2], oc[3]);
    if (output_half) {
        uint32_t lo = (uint32_t)f32_to_f16_rne(oc[0]) | ((uint32_t)f32_to_f16_rne(oc[1]) << 16);
        uint32_t hi = (uint32_t)f32_to_f16_rne(oc[2]) | ((uint32_t)f32_to_f16_rne(oc[3]) << 16);
        output_half[pidx] = make_uint2(lo, hi);
    }
    result[pidx] = total;
}

inline float h_dot4(const float a[4], const float b[4]) { return ((a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]) + a[3] * b[3]; }

}  // namespace

extern "C" {

int rt_camera_ray_generation(rt_device* dev, const rt_camera_ray_gen_info* info, rt_buffer* rays, rt_buffer* old_rays,
                             rt_buffer* ray_pixels, const rt_image_rows* rows) {
    if (!dev || !info || !rays || !old_rays || !ray_pixels) return RT_ERR_INVALID_ARGUMENT;
    const int32_t W = info->screen_size[0], H = info->screen_size[1];
    if (W <= 0 || H <= 0 || W > 65535 || H > 65536) return RT_ERR_INVALID_ARGUMENT;      // pixel x,y are packed in 16 bits each
    const uint32_t first_row = rows ? rows->first_row : 0u, num_rows = rows ? rows->num_rows : (uint32_t)H;
    const uint32_t row_stride = rows ? rows->row_stride : 1u;
    if (!rt_rows_fit(first_row, num_rows, row_stride, (uint32_t)H)) return RT_ERR_INVALID_ARGUMENT;
    const size_t nr = (size_t)num_rows * W * info->max_rays_per_pixel;
    if (nr > 0xffffffffull) return RT_ERR_INVALID_ARGUMENT;
    if (rays->bytes < nr * 32u || old_rays->bytes < nr * 32u || ray_pixels->bytes < nr * 4u) return RT_ERR_BUFFER_TOO_SMALL;
    if (!nr) return RT_OK;
    int rc = rt_set_device(dev);
    if (rc) return rc;

    RaygenParams p;
    // constant-buffer matrices are read column-major: M[r][c] = mem[4c + r]
    float IV[4][4], IP[4][4];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            IV[r][c] = info->inverse_view[4 * c + r];
            IP[r][c] = info->inverse_projection[4 * c + r];
        }
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            const float col[4] = {IV[0][j], IV[1][j], IV[2][j], IV[3][j]};
            p.ivp[4 * i + j] = h_dot4(IP[i], col);
        }
    const float unit_w[4] = {0.0f, 0.0f, 0.0f, 1.0f};
    for (int r = 0; r < 3; r++) p.cam_add[r] = h_dot4(IV[r], unit_w);
    p.width = W; p.height = H;
    p.aa = info->aa_sample_spread; p.dof = info->dof_sample_spread;
    p.rpp = info->max_rays_per_pixel;
    for (int k = 0; k < 3; k++) p.seed[k] = info->rng_seed[k];
    p.first_row = first_row; p.num_rows = num_rows; p.row_stride = row_stride;
    const dim3 grid(((uint32_t)W + 63u) / 64u, (num_rows + 3u) / 4u), block(64, 4);
    dev->missed_valid = false;          // new rays: no ray has missed yet (drop-missed-rays mode),
    dev->keys_ready = false;            // and the keys the last shading left behind are not theirs
    hipLaunchKernelGGL(k_raygen, grid, block, 0, dev->stream, p, (float4*)rays->ptr, (float4*)old_rays->ptr,
                       (uint32_t*)ray_pixels->ptr);
    return rt_launch_status();
}

int rt_generate_final_image(rt_device* dev, const rt_generate_final_image_info* info, const rt_buffer* emitted_light,
                            rt_buffer* result, rt_buffer* output, rt_buffer* output_half) {
    if (!dev || !info || !emitted_light || !result || !output) return RT_ERR_INVALID_ARGUMENT;
    if (info->screen_dimensions[0] <= 0 || info->screen_dimensions[1] < 0 || !info->max_rays_per_pixel ||
        !(info->num_samples & 0x7fffffffu))
        return RT_ERR_INVALID_ARGUMENT;
    const size_t pixels = (size_t)info->screen_dimensions[0] * (size_t)info->screen_dimensions[1];
    if (pixels > 0xffffffffull) return RT_ERR_INVALID_ARGUMENT;
    if (emitted_light->bytes < pixels * info->max_rays_per_pixel * 16u || result->bytes < pixels * 16u ||
        output->bytes < pixels * 16u || (output_half && output_half->bytes < pixels * 8u))
        return RT_ERR_BUFFER_TOO_SMALL;
    if (!pixels) return RT_OK;
    int rc = rt_set_device(dev);
    if (rc) return rc;
    hipLaunchKernelGGL(k_final_image, dim3(((uint32_t)pixels + 255u) / 256u), dim3(256), 0, dev->stream, (uint32_t)pixels,
                       info->max_rays_per_pixel, info->num_samples, (const float4*)emitted_light->ptr, (float4*)result->ptr,
                       (float4*)output->ptr, output_half ? (uint2*)output_half->ptr : nullptr);
    return rt_launch_status();
}

}  // extern "C"
