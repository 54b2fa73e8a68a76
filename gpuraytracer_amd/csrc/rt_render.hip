// Camera-ray generation, BVH traversal + per-ray shading, and final-image accumulation for gfx950.
//
// Semantics follow the reference's compute shaders
//   shader/CS_CameraRayGeneration.hlsl, CS_TraceRays.hlsl, PerRayShading.hlsli, Random.hlsli,
//   CS_GenerateFinalImage.hlsl
// (dispatched by src/RaytracerPipeline.cpp:105-150, 687-729, 809-854), including their quirks: the
// ray-generation seed depends on pixel x only, a missed ray is left untouched, the interpolated
// normal is not renormalised, texture offsets are byte counts used as texel indices.
//
// What is new is the traversal: the reference keeps a 48-entry stack of node indices in private
// memory. The hierarchy it builds is a balanced implicit tree whose shape depends only on the
// primitive count (children of node i of level k are nodes 2i, 2i+1 of level k-1), so the same
// left-first depth-first order is walked here with two registers (level, index) and no stack:
// "pop" is "while this node is a right child or has no sibling, go to the parent; then step to
// the sibling". The order of box and triangle tests, and therefore the closest hit and its
// tie-breaking, is exactly the reference's.
//
// Arithmetic conventions (one rounding per written operation, -ffp-contract=off):
//   dot(a,b) = (a.x*b.x + a.y*b.y) + a.z*b.z; normalize(v) = v / sqrt(dot(v,v)); pow(x,5) =
//   ((x*x)*(x*x))*x; lerp(a,b,t) = a + t*(b-a); sin/cos = range reduction + 11/10-degree minimax
//   polynomials; the slab test multiplies by one IEEE reciprocal of each direction component per
//   ray instead of dividing per box.

#include "rt_internal.h"

namespace {

struct f3 { float x, y, z; };

__device__ __forceinline__ f3 mk(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 add(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 sub(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 mul(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ f3 scale(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 divs(f3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ f3 cross3(f3 a, f3 b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ f3 normalize3(f3 a) { return divs(a, sqrtf(dot3(a, a))); }
__device__ __forceinline__ float saturatef(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
__device__ __forceinline__ float pow5(float x) { float x2 = x * x; return (x2 * x2) * x; }
__device__ __forceinline__ uint32_t f2u(float x) {
    if (!(x > 0.0f)) return 0u;
    if (x >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)x;
}

// shader/Random.hlsli:3-9, 63-66
__device__ __forceinline__ uint32_t xorshift(uint32_t& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }
__device__ __forceinline__ float rnd(uint32_t& s) { return saturatef((float)xorshift(s) * 2.3283064e-10f); }

__device__ __forceinline__ void sincos_poly(float x, float* s, float* c) {
    float q = 0.15915494f * x;
    q = (x >= 0.0f) ? (float)(int)(q + 0.5f) : (float)(int)(q - 0.5f);
    float y = x - 6.2831855f * q;
    float sign;
    if (y > 1.5707964f) { y = 3.1415927f - y; sign = -1.0f; }
    else if (y < -1.5707964f) { y = -3.1415927f - y; sign = -1.0f; }
    else sign = 1.0f;
    float y2 = y * y;
    *s = (((((-2.3889859e-08f * y2 + 2.7525562e-06f) * y2 - 0.00019840874f) * y2 + 0.0083333310f) * y2 - 0.16666667f) * y2 + 1.0f) * y;
    float p = ((((-2.6051615e-07f * y2 + 2.4760495e-05f) * y2 - 0.0013888378f) * y2 + 0.041666638f) * y2 - 0.5f) * y2 + 1.0f;
    *c = sign * p;
}

// ------------------------------------------------------------------ ray generation ----

struct RaygenParams {
    float ivp[16];            // mul(InverseProjection, InverseView), row-major [i][j]
    float cam_add[3];         // mul(InverseView, (0,0,0,1)).xyz
    int32_t width, height;
    float aa, dof;
    uint32_t rpp;
    uint32_t seed[3];
    uint32_t first_row, num_rows;
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
    const uint32_t y = p.first_row + row;
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

// ------------------------------------------------------------------------- tracing ----

struct TraceParams {
    int32_t width;
    uint32_t num_indices, num_rays, rpp;
    uint32_t seed[3];
    uint32_t first_ray;
    uint32_t atlas_elems;
    uint32_t num_prims;
    rt_levels lv;
};

struct Hit { float t, u, v, w; };

// shader/CS_TraceRays.hlsl:46-77
__device__ __forceinline__ Hit intersect_tri(f3 o, f3 d, f3 v1, f3 v2, f3 v3) {
    f3 e1 = sub(v2, v1), e2 = sub(v3, v1), tv = sub(o, v1);
    f3 pv = cross3(d, e2), qv = cross3(tv, e1);
    float det = dot3(e1, pv);
    float inv = (fabsf(det) < 1e-6f) ? 0.0f : 1.0f / det;
    float cw = rintf(fabsf(det) * inv);
    float t = dot3(e2, qv) * inv, u = dot3(tv, pv) * inv, v = dot3(d, qv) * inv;
    Hit h;
    if (t < 1e-6f || u < 0.0f || v < 0.0f || u + v > 1.0f) { h.t = -1.0f; h.u = -1.0f; h.v = -1.0f; }
    else { h.t = t; h.u = u; h.v = v; }
    h.w = cw;
    return h;
}

// shader/CS_TraceRays.hlsl:100-113 (the accepted hit remembers the triangle's first index slot)
__device__ __forceinline__ void check_hit(const Hit& h, float tmin, float tmax, uint32_t ci, Hit& best, uint32_t& best_ci) {
    bool use = (tmin <= h.t) && (tmax > h.t) && (h.t < best.t);
    if (use) { best = h; best_ci = ci; }
}

// shader/CS_TraceRays.hlsl:81-97
__device__ __forceinline__ float slab(f3 o, f3 inv_d, float tmin_r, float tmax_r, float4 lo, float4 hi) {
    f3 t1 = mul(sub(mk(lo.x, lo.y, lo.z), o), inv_d);
    f3 t2 = mul(sub(mk(lo.w, hi.x, hi.y), o), inv_d);
    float tmin = fmaxf(fminf(t1.x, t2.x), fmaxf(fminf(t1.y, t2.y), fminf(t1.z, t2.z)));
    float tmax = fminf(fmaxf(t1.x, t2.x), fminf(fmaxf(t1.y, t2.y), fmaxf(t1.z, t2.z)));
    if ((tmax < tmin) || (tmax <= 0.0f) || (tmax < tmin_r) || (tmin > tmax_r)) tmin = 1e30f;
    return tmin;
}

// shader/PerRayShading.hlsli:60-74
__device__ __forceinline__ f3 sample_texture(rt_texture_id id, float u, float v, const uint4* __restrict__ atlas, uint32_t atlas_elems) {
    uint32_t sx = f2u(rintf(u * (float)(id.width - 1u)));
    uint32_t sy = f2u(rintf(v * (float)(id.height - 1u)));
    uint32_t loc = sx + id.row_pitch * sy + id.offset;
    uint32_t e = loc >> 1;
    uint2 px = make_uint2(0u, 0u);                      // loads past the atlas return 0
    if (e < atlas_elems) px = reinterpret_cast<const uint2*>(atlas)[loc];
    f3 c = mk((float)(px.x & 0xffffu), (float)(px.x >> 16), (float)(px.y & 0xffffu));
    c = scale(c, 1.5259022e-5f);
    return mul(c, c);
}

// shader/PerRayShading.hlsli:79-106
__device__ __forceinline__ f3 reflected_color(float rough, f3 f0, float ndl, float ndv, float ndh, float vdh) {
    float alpha = rough * rough;
    float sf = pow5(saturatef(1.0f - vdh));
    f3 F = mk((f0.x + sf) - f0.x * sf, (f0.y + sf) - f0.y * sf, (f0.z + sf) - f0.z * sf);
    float a2 = alpha * alpha;
    float ndh2 = ndh * ndh;
    float sq = (ndh2 * a2 - ndh2) + 1.0f;
    float dden = (3.141592654f * sq) * sq;
    float k = alpha * 0.5f;
    float gx = (ndv - k * ndv) + k;
    float gy = (ndl - k * ndl) + k;
    float den = dden * (gx * gy) + 1e-6f;
    return mk(((0.25f * F.x) * a2) / den, ((0.25f * F.y) * a2) / den, ((0.25f * F.z) * a2) / den);
}

// shader/PerRayShading.hlsli:110-133
__device__ __forceinline__ f3 refracted_color(float rough, float metal, f3 albedo, float ndl, float ndv, float vdh, float ldv) {
    float facing = ldv * 0.5f + 0.5f;
    float prough = facing * (0.9f - 0.4f * facing);
    float rgh = prough * (vdh / ((ndl + ndv) + 1e-6f)) + prough;
    float smooth = (1.05f * (1.0f - pow5(1.0f - ndl))) * (1.0f - pow5(1.0f - ndv));
    float alpha = rough * rough;
    float single = (smooth + alpha * (rgh - smooth)) / 3.141592654f;
    float multi = 0.1159f * alpha;
    f3 diff = mk(albedo.x * (albedo.x * multi + single), albedo.y * (albedo.y * multi + single),
                 albedo.z * (albedo.z * multi + single));
    return mk(diff.x - diff.x * metal, diff.y - diff.y * metal, diff.z - diff.z * metal);
}

// shader/CS_TraceRays.hlsl:129-144
__device__ __forceinline__ float interp(float a1, float a2, float a3, float u, float v) {
    return (a1 * ((1.0f - u) - v) + a2 * u) + a3 * v;
}

__device__ __forceinline__ f3 vpos(const rt_vertex* __restrict__ verts, uint32_t i) {
    const float* p = verts[i].position;
    return mk(p[0], p[1], p[2]);
}

// One thread per ray. USE_BVH mirrors RT_USE_BVH; PACKED reads leaf triangles from the Morton-ordered
// float4 x 3 array (leaf i holds sorted triangles 2i, 2i+1) instead of Indices -> Vertices.
// STATS adds the traversal counters (box tests, triangle tests, hits, rays) to stats[0..3].
template <bool USE_BVH, bool PACKED, bool STATS>
__global__ __launch_bounds__(256) void k_trace(TraceParams p, const uint32_t* __restrict__ indices,
                                               const rt_vertex* __restrict__ verts, const rt_pbr_material* __restrict__ mats,
                                               const rt_texture_id* __restrict__ tex, const uint4* __restrict__ atlas,
                                               float4* __restrict__ rays, float4* __restrict__ old_rays,
                                               const uint32_t* __restrict__ ray_pixels, float4* __restrict__ scattered,
                                               float4* __restrict__ emitted, const float4* __restrict__ nodes,
                                               const float4* __restrict__ sorted_tris, unsigned long long* __restrict__ stats) {
    __shared__ uint32_t s_off[33];
    __shared__ uint32_t s_n[33];
    if (USE_BVH) {
        if (threadIdx.x < 33u) { s_off[threadIdx.x] = p.lv.off[threadIdx.x]; s_n[threadIdx.x] = p.lv.n[threadIdx.x]; }
        __syncthreads();
    }
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    if (tid >= p.num_rays) return;

    const float4 r0 = rays[2 * (size_t)tid], r1 = rays[2 * (size_t)tid + 1];
    const float4 o0 = old_rays[2 * (size_t)tid], o1 = old_rays[2 * (size_t)tid + 1];
    const f3 org = mk(r0.x, r0.y, r0.z), dir = mk(r0.w, r1.x, r1.y);
    const float tmin_r = r1.z, tmax_r = r1.w;

    Hit best;
    best.t = tmax_r; best.u = 0.0f; best.v = 0.0f; best.w = 0.0f;
    uint32_t best_ci = 0u;
    uint32_t n_nodes = 0u, n_tris = 0u;

    if (!USE_BVH) {
        // shader/CS_TraceRays.hlsl:160-175
        for (uint32_t ci = 0; ci < p.num_indices; ci += 3u) {
            Hit h = intersect_tri(org, dir, vpos(verts, indices[ci]), vpos(verts, indices[ci + 1]), vpos(verts, indices[ci + 2]));
            check_hit(h, tmin_r, tmax_r, ci, best, best_ci);
            if (STATS) n_tris++;
        }
    } else {
        // shader/CS_TraceRays.hlsl:179-237, walked without a stack (see the file header)
        const f3 inv_d = mk(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
        const uint32_t root_level = p.lv.count - 1u;
        uint32_t k = root_level, i = 0u, node = 0u;
        bool first = true;
        for (;;) {
            const float4 lo = nodes[2 * (size_t)node], hi = nodes[2 * (size_t)node + 1];
            const float bt = slab(org, inv_d, tmin_r, tmax_r, lo, hi);
            if (STATS) n_nodes++;
            // the trunk is entered on any box hit; every other node also needs to start before the best hit
            bool enter = (bt != 1e30f) && (first || bt < best.t);
            first = false;
            if (enter && k > 0u) {
                k--;
                i = 2u * i;
                node = s_off[k] + i;
                continue;
            }
            if (enter) {
                const uint32_t l0 = __float_as_uint(hi.z), l1 = __float_as_uint(hi.w);
                const uint32_t ci0 = l0 & 0x7fffffffu;
                f3 a, b, c;
                if (PACKED) {
                    const float4* t = sorted_tris + 6 * (size_t)i;
                    float4 ta = t[0], tb = t[1], tc = t[2];
                    a = mk(ta.x, ta.y, ta.z); b = mk(tb.x, tb.y, tb.z); c = mk(tc.x, tc.y, tc.z);
                } else {
                    a = vpos(verts, indices[ci0]); b = vpos(verts, indices[ci0 + 1]); c = vpos(verts, indices[ci0 + 2]);
                }
                check_hit(intersect_tri(org, dir, a, b, c), tmin_r, tmax_r, ci0, best, best_ci);
                if (STATS) n_tris += (l1 != 0xffffffffu) ? 2u : 1u;
                if (l1 != 0xffffffffu) {
                    const uint32_t ci1 = l1 & 0x7fffffffu;
                    if (PACKED) {
                        const float4* t = sorted_tris + 6 * (size_t)i + 3;
                        float4 ta = t[0], tb = t[1], tc = t[2];
                        a = mk(ta.x, ta.y, ta.z); b = mk(tb.x, tb.y, tb.z); c = mk(tc.x, tc.y, tc.z);
                    } else {
                        a = vpos(verts, indices[ci1]); b = vpos(verts, indices[ci1 + 1]); c = vpos(verts, indices[ci1 + 2]);
                    }
                    check_hit(intersect_tri(org, dir, a, b, c), tmin_r, tmax_r, ci1, best, best_ci);
                }
            }
            // pop: climb while this node is a right child or an only child, then step to the sibling
            bool done = false;
            for (;;) {
                if (k == root_level) { done = true; break; }
                if (!(i & 1u) && (i + 1u) < s_n[k]) { i++; node = s_off[k] + i; break; }
                i >>= 1;
                k++;
            }
            if (done) break;
        }
    }
    if (STATS) {
        unsigned long long c0 = n_nodes, c1 = n_tris, c2 = (best.t != tmax_r) ? 1u : 0u, c3 = 1u;
        if (__builtin_amdgcn_ballot_w64(true) == ~0ull) {
            // whole wave present: one atomic per counter per wave
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                c0 += __shfl_xor(c0, d, 64); c1 += __shfl_xor(c1, d, 64); c2 += __shfl_xor(c2, d, 64); c3 += __shfl_xor(c3, d, 64);
            }
            if ((threadIdx.x & 63u) == 0u) {
                atomicAdd(&stats[0], c0); atomicAdd(&stats[1], c1); atomicAdd(&stats[2], c2); atomicAdd(&stats[3], c3);
            }
        } else {
            atomicAdd(&stats[0], c0); atomicAdd(&stats[1], c1); atomicAdd(&stats[2], c2); atomicAdd(&stats[3], c3);
        }
    }

    // shader/CS_TraceRays.hlsl:116-126, 241-246
    const uint32_t pv = ray_pixels[tid];
    const uint32_t px = pv >> 16, py = pv & 0xffffu;
    const uint32_t off = __float_as_uint(o1.w);
    const uint32_t idx = ((uint32_t)p.width * py + px) * p.rpp + off - p.first_ray;
    float4 S = scattered[idx];
    float4 E = emitted[idx];

    if (best.t != tmax_r) {
        const uint32_t ia = indices[best_ci], ib = indices[best_ci + 1], ic = indices[best_ci + 2];
        const rt_vertex A = verts[ia], B = verts[ib], C = verts[ic];
        const uint32_t gid = tid + p.first_ray;
        uint32_t s0 = gid * (gid + 17u);
        xorshift(s0);
        uint32_t sx = s0 + p.seed[0], sy = s0 + p.seed[1];

        const float u = best.u, v = best.v;
        const float tu = interp(A.uv[0], B.uv[0], C.uv[0], u, v), tv = interp(A.uv[1], B.uv[1], C.uv[1], u, v);
        const f3 N = mk(interp(A.normal[0], B.normal[0], C.normal[0], u, v), interp(A.normal[1], B.normal[1], C.normal[1], u, v),
                        interp(A.normal[2], B.normal[2], C.normal[2], u, v));

        // shader/Random.hlsli:100-126
        f3 p1 = normalize3(mk(-N.z, 0.0f, N.x));
        if (N.z == 0.0f && N.x == 0.0f) p1 = mk(1.0f, 0.0f, 0.0f);
        const f3 p2 = normalize3(cross3(N, p1));
        p1 = normalize3(cross3(N, p2));
        const float rn1 = rnd(sx), rn2 = rnd(sy);
        const float hy = sqrtf(1.0f - rn2);
        const float hr = sqrtf(rn2);
        const float phi = 6.2831853f * rn1;
        float sn, cs;
        sincos_poly(phi, &sn, &cs);
        const f3 L = add(add(scale(p1, cs * hr), scale(N, hy)), scale(p2, sn * hr));

        if (dot3(mk(o0.w, o1.x, o1.y), mk(o0.w, o1.x, o1.y)) == 0.0f) {
            S.x = 1.0f; S.y = 1.0f; S.z = 1.0f;
            E.x = 0.0f; E.y = 0.0f; E.z = 0.0f;
        }

        // shader/PerRayShading.hlsli:138-191
        const f3 V = mk(-dir.x, -dir.y, -dir.z);
        const f3 H = normalize3(add(V, L));
        const rt_pbr_material m = mats[A.material_id];
        const f3 ta = sample_texture(tex[m.albedo_texture], tu, tv, atlas, p.atlas_elems);
        const f3 tr = sample_texture(tex[m.roughness_texture], tu, tv, atlas, p.atlas_elems);
        const f3 tf = sample_texture(tex[m.f0_texture], tu, tv, atlas, p.atlas_elems);
        const f3 tm = sample_texture(tex[m.metallic_texture], tu, tv, atlas, p.atlas_elems);
        const f3 te = sample_texture(tex[m.emissive_texture], tu, tv, atlas, p.atlas_elems);
        const f3 albedo = mul(mk(m.albedo[0], m.albedo[1], m.albedo[2]), ta);
        const float rough = m.roughness * tr.x;
        const f3 f0 = mul(mk(m.f0_color[0], m.f0_color[1], m.f0_color[2]), tf);
        const float metal = m.metallic * tm.x;
        const f3 emis = mul(mk(m.emissive[0], m.emissive[1], m.emissive[2]), te);

        const float vdh = saturatef(dot3(V, H)), ndh = saturatef(dot3(N, H)), ndv = saturatef(dot3(N, V));
        const float ndl = saturatef(dot3(N, L)), ldv = saturatef(dot3(L, V));
        const f3 Fr = reflected_color(rough, f0, ndl, ndv, ndh, vdh);
        const f3 Fd = refracted_color(rough, metal, albedo, ndl, ndv, vdh, ldv);
        const f3 Sc = mk(S.x, S.y, S.z), Em = mk(E.x, E.y, E.z);
        const f3 kk = scale(add(Fr, Fd), ndl);
        const f3 newS = mul(Sc, mk(fmaxf(0.0f, kk.x), fmaxf(0.0f, kk.y), fmaxf(0.0f, kk.z)));
        const f3 se = mul(Sc, emis);
        const f3 newE = add(mk(fmaxf(0.0f, se.x), fmaxf(0.0f, se.y), fmaxf(0.0f, se.z)), Em);
        S.x = newS.x; S.y = newS.y; S.z = newS.z;
        E.x = newE.x; E.y = newE.y; E.z = newE.z;

        // new ray from the hit point; the traced ray becomes the "old" ray, keeping the sample slot in TMax
        rays[2 * (size_t)tid] = make_float4(org.x + dir.x * best.t, org.y + dir.y * best.t, org.z + dir.z * best.t, L.x);
        rays[2 * (size_t)tid + 1] = make_float4(L.y, L.z, tmin_r, tmax_r);
        old_rays[2 * (size_t)tid] = r0;
        old_rays[2 * (size_t)tid + 1] = make_float4(r1.x, r1.y, r1.z, o1.w);
    }
    scattered[idx] = S;
    emitted[idx] = E;
}

// ---------------------------------------------------------------------- final image ----

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

int trace_common(rt_device* dev, const rt_trace_rays_info* info, int use_bvh, const rt_accel* accel, const rt_buffer* indices,
                 const rt_buffer* vertices, const rt_buffer* materials, const rt_buffer* texture_ids,
                 const rt_buffer* texture_atlas, uint32_t atlas_elements, rt_buffer* rays, rt_buffer* old_rays,
                 const rt_buffer* ray_pixels, rt_buffer* scattered_light, rt_buffer* emitted_light, const rt_buffer* bvh,
                 uint32_t first_ray) {
    unsigned long long* stats = dev ? dev->trace_stats : nullptr;
    if (!dev || !info || !indices || !vertices || !materials || !texture_ids || !texture_atlas || !rays || !old_rays ||
        !ray_pixels || !scattered_light || !emitted_light)
        return RT_ERR_INVALID_ARGUMENT;
    if (use_bvh && !bvh) return RT_ERR_INVALID_ARGUMENT;
    const uint32_t nr = info->num_rays;
    if (rays->bytes < (size_t)nr * 32u || old_rays->bytes < (size_t)nr * 32u || ray_pixels->bytes < (size_t)nr * 4u ||
        scattered_light->bytes < (size_t)nr * 16u || emitted_light->bytes < (size_t)nr * 16u)
        return RT_ERR_BUFFER_TOO_SMALL;
    if (indices->bytes < (size_t)info->num_indices * 4u || texture_atlas->bytes < (size_t)atlas_elements * 16u)
        return RT_ERR_BUFFER_TOO_SMALL;
    if (!nr) return RT_OK;
    int rc = rt_set_device(dev);
    if (rc) return rc;
    TraceParams p;
    p.width = info->screen_dimensions[0];
    p.num_indices = info->num_indices;
    p.num_rays = nr;
    p.rpp = info->max_rays_per_pixel;
    for (int k = 0; k < 3; k++) p.seed[k] = info->rng_seed[k];
    p.first_ray = first_ray;
    p.atlas_elems = atlas_elements;
    p.num_prims = info->num_indices / 3u;
    p.lv.count = 0;
    if (use_bvh) {
        if (!rt_make_levels(p.num_prims, &p.lv)) return RT_ERR_TOO_FEW_PRIMITIVES;
        if (bvh->bytes < (size_t)rt_bvh_node_count(p.num_prims) * 32u) return RT_ERR_BUFFER_TOO_SMALL;
        if (accel && accel->num_prims != p.num_prims) return RT_ERR_INVALID_ARGUMENT;
    }
    const dim3 grid((nr + 255u) / 256u), block(256);
    const float4* nodes = bvh ? (const float4*)bvh->ptr : nullptr;
#define RT_TRACE_ARGS                                                                                                     \
    p, (const uint32_t*)indices->ptr, (const rt_vertex*)vertices->ptr, (const rt_pbr_material*)materials->ptr,            \
        (const rt_texture_id*)texture_ids->ptr, (const uint4*)texture_atlas->ptr, (float4*)rays->ptr, (float4*)old_rays->ptr, \
        (const uint32_t*)ray_pixels->ptr, (float4*)scattered_light->ptr, (float4*)emitted_light->ptr, nodes
    const float4* st = accel ? (const float4*)accel->sorted_tris : nullptr;
    if (!use_bvh && !stats) hipLaunchKernelGGL((k_trace<false, false, false>), grid, block, 0, dev->stream, RT_TRACE_ARGS, st, stats);
    else if (!use_bvh) hipLaunchKernelGGL((k_trace<false, false, true>), grid, block, 0, dev->stream, RT_TRACE_ARGS, st, stats);
    else if (accel && !stats) hipLaunchKernelGGL((k_trace<true, true, false>), grid, block, 0, dev->stream, RT_TRACE_ARGS, st, stats);
    else if (accel) hipLaunchKernelGGL((k_trace<true, true, true>), grid, block, 0, dev->stream, RT_TRACE_ARGS, st, stats);
    else if (!stats) hipLaunchKernelGGL((k_trace<true, false, false>), grid, block, 0, dev->stream, RT_TRACE_ARGS, st, stats);
    else hipLaunchKernelGGL((k_trace<true, false, true>), grid, block, 0, dev->stream, RT_TRACE_ARGS, st, stats);
#undef RT_TRACE_ARGS
    return rt_launch_status();
}

}  // namespace

extern "C" {

int rt_device_set_trace_stats(rt_device* dev, rt_buffer* stats) {
    if (!dev || (stats && stats->bytes < 32u)) return RT_ERR_INVALID_ARGUMENT;
    dev->trace_stats = stats ? (unsigned long long*)stats->ptr : nullptr;
    return RT_OK;
}

int rt_camera_ray_generation(rt_device* dev, const rt_camera_ray_gen_info* info, rt_buffer* rays, rt_buffer* old_rays,
                             rt_buffer* ray_pixels, uint32_t first_row, uint32_t num_rows) {
    if (!dev || !info || !rays || !old_rays || !ray_pixels) return RT_ERR_INVALID_ARGUMENT;
    const int32_t W = info->screen_size[0], H = info->screen_size[1];
    if (W <= 0 || H <= 0 || W > 65535 || H > 65536) return RT_ERR_INVALID_ARGUMENT;      // pixel x,y are packed in 16 bits each
    if (first_row > (uint32_t)H || num_rows > (uint32_t)H - first_row) return RT_ERR_INVALID_ARGUMENT;
    const size_t nr = (size_t)num_rows * W * info->max_rays_per_pixel;
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
    p.first_row = first_row; p.num_rows = num_rows;
    const dim3 grid(((uint32_t)W + 63u) / 64u, (num_rows + 3u) / 4u), block(64, 4);
    hipLaunchKernelGGL(k_raygen, grid, block, 0, dev->stream, p, (float4*)rays->ptr, (float4*)old_rays->ptr,
                       (uint32_t*)ray_pixels->ptr);
    return rt_launch_status();
}

int rt_trace_rays(rt_device* dev, const rt_trace_rays_info* info, int use_bvh, const rt_buffer* indices,
                  const rt_buffer* vertices, const rt_buffer* materials, const rt_buffer* texture_ids,
                  const rt_buffer* texture_atlas, uint32_t atlas_elements, rt_buffer* rays, rt_buffer* old_rays,
                  const rt_buffer* ray_pixels, rt_buffer* scattered_light, rt_buffer* emitted_light, const rt_buffer* bvh,
                  uint32_t first_ray) {
    return trace_common(dev, info, use_bvh, nullptr, indices, vertices, materials, texture_ids, texture_atlas, atlas_elements,
                        rays, old_rays, ray_pixels, scattered_light, emitted_light, bvh, first_ray);
}

int rt_trace_rays_accel(rt_device* dev, const rt_trace_rays_info* info, const rt_accel* accel, const rt_buffer* indices,
                        const rt_buffer* vertices, const rt_buffer* materials, const rt_buffer* texture_ids,
                        const rt_buffer* texture_atlas, uint32_t atlas_elements, rt_buffer* rays, rt_buffer* old_rays,
                        const rt_buffer* ray_pixels, rt_buffer* scattered_light, rt_buffer* emitted_light, const rt_buffer* bvh,
                        uint32_t first_ray) {
    if (!accel) return RT_ERR_INVALID_ARGUMENT;
    return trace_common(dev, info, 1, accel, indices, vertices, materials, texture_ids, texture_atlas, atlas_elements, rays,
                        old_rays, ray_pixels, scattered_light, emitted_light, bvh, first_ray);
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
