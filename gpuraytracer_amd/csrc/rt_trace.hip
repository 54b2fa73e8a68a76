// BVH traversal + per-ray shading for gfx950.
//
// Semantics follow the reference's shader/CS_TraceRays.hlsl, PerRayShading.hlsli and Random.hlsli
// (dispatched by src/RaytracerPipeline.cpp:687-729), including their quirks: a missed ray is left
// untouched, the interpolated normal is not renormalised, texture offsets are byte counts used as
// texel indices, the seed of a ray is a function of its global index.
//
// What is new is how the work is laid out:
//  * The reference keeps a 48-entry stack of node indices in private memory. The hierarchy it builds is a balanced
//    implicit tree whose shape depends only on the primitive count, so the same left-first depth-first order is
//    walked here without a stack: nodes are numbered as in a complete binary tree (the node in slot h has its
//    children in slots 2h and 2h + 1), and the nodes still to be visited are a few bits per level in a register.
//    The order of the triangle tests, hence the closest hit and its tie-breaking, is exactly the reference's.
//  * The reference runs one thread per ray through traversal and shading. Here traversal is a persistent kernel:
//    waves pull ray indices from atomic queues and refill finished lanes, so a wave never idles behind its longest
//    ray; a lane that reaches a leaf notes it in a first-in first-out list and walks on, and the triangle tests
//    run when enough lanes have one waiting; closest hits go to a 20-byte record and a second, fully convergent
//    kernel shades them.
//  * Three traversal kernels walk the reference's hierarchy in the reference's order and give the same buffers by
//    construction: k_traverse_quad (two levels per step on 64-byte records of widened binary16 boxes, exact
//    re-test at the leaves), k_traverse (one level per step on exact sibling pairs; its counting variant reproduces
//    the reference's box-test count), k_trace (one thread per ray; also the brute-force path without a hierarchy).
//  * The default, k_traverse_spatial, finds the same closest hits on a second hierarchy of the library's own and hands
//    the rays for which the reference's order could decide to k_traverse_quad (see the description at the kernel); a
//    short list of such rays is walked in parts, one subtree per lane (k_traverse_quad<true>, k_listed_merge).
//
// Arithmetic conventions are those of rt_math.h; the slab test multiplies by one IEEE reciprocal of
// each direction component per ray instead of dividing per box.

#include <cstdlib>
#include <cstring>
#include <new>

#include "rt_internal.h"
#include "rt_math.h"

namespace {

using namespace rtm;

struct TraceParams {
    int32_t width;
    uint32_t num_indices, num_rays, rpp;
    uint32_t seed[3];
    uint32_t first_ray;        // index in the whole image of the first ray of the device's first row
    uint32_t row_len;          // rays per image row (width * rays per pixel)
    uint32_t row_stride;       // the device holds every row_stride-th image row, stored one after the other
    uint32_t atlas_elems;
    uint32_t num_prims;
    uint32_t num_materials, num_textures;
    uint32_t leaf_threshold;
    uint32_t queue_chunk;      // slots per queue of the persistent traversal kernel (a multiple of the dealt block)
    uint32_t refill_threshold; // lanes that must be free before a wave of k_traverse_quad takes new rays
    uint32_t deal_shift;       // blocks of 2^deal_shift consecutive positions of the ray order are dealt to the queues in turn
    float beyond;              // k_traverse_spatial enters boxes up to reach * beyond
    uint32_t trunk_children;   // node index (reference layout) of the first of the trunk's two children, for hierarchy lv
    uint32_t stuck_threshold;  // k_traverse_spatial: lanes whose walk is over with leaves still waiting that start a triangle round
    uint32_t walk_hist;        // counting variant of k_traverse_spatial: the counter buffer also has room for steps by depth
    uint32_t retire_ticks;     // k_traverse_spatial: workgroups of class c take no more rays once the queues will be empty within c * retire_ticks (of the 100 MHz clock; 0: off)
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

// shader/CS_TraceRays.hlsl:81-97. DIV: inv_d is the direction itself and the test divides by it, as the shader is
// written; otherwise inv_d holds one reciprocal per axis and the test multiplies.
template <bool DIV = false>
__device__ __forceinline__ float slab(f3 o, f3 inv_d, float tmin_r, float tmax_r, float4 lo, float4 hi) {
    const f3 a = sub(mk(lo.x, lo.y, lo.z), o), b = sub(mk(lo.w, hi.x, hi.y), o);
    f3 t1 = DIV ? mk(a.x / inv_d.x, a.y / inv_d.y, a.z / inv_d.z) : mul(a, inv_d);
    f3 t2 = DIV ? mk(b.x / inv_d.x, b.y / inv_d.y, b.z / inv_d.z) : mul(b, inv_d);
    float tmin = fmaxf(fminf(t1.x, t2.x), fmaxf(fminf(t1.y, t2.y), fminf(t1.z, t2.z)));
    float tmax = fminf(fmaxf(t1.x, t2.x), fminf(fmaxf(t1.y, t2.y), fmaxf(t1.z, t2.z)));
    if ((tmax < tmin) || (tmax <= 0.0f) || (tmax < tmin_r) || (tmin > tmax_r)) tmin = 1e30f;
    return tmin;
}

// shader/PerRayShading.hlsli:60-74. A texture id past the table reads zeros, like an out-of-range
// structured-buffer load.
__device__ __forceinline__ f3 sample_texture(const rt_texture_id* __restrict__ tex, uint32_t num_textures, uint32_t tid, float u,
                                             float v, const uint4* __restrict__ atlas, uint32_t atlas_elems) {
    rt_texture_id id;
    id.offset = 0u; id.width = 0u; id.height = 0u; id.row_pitch = 0u;
    if (tid < num_textures) id = tex[tid];
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

struct ShadeBuffers {
    const uint32_t* indices;
    const rt_vertex* verts;
    const rt_pbr_material* mats;
    const rt_texture_id* tex;
    const uint4* atlas;
    float4* rays;
    float4* old_rays;
    const uint32_t* ray_pixels;
    float4* scattered;
    float4* emitted;
};

// Index in the whole image of the ray held in element tid of this device's buffers, and back: the device holds
// image rows first, first + stride, ... one after the other.
__device__ __forceinline__ uint32_t global_ray(const TraceParams& p, uint32_t tid) {
    if (p.row_stride == 1u) return tid + p.first_ray;
    return tid + p.first_ray + (tid / p.row_len) * (p.row_len * (p.row_stride - 1u));
}
__device__ __forceinline__ uint32_t local_ray(const TraceParams& p, uint32_t gid) {
    const uint32_t g = gid - p.first_ray;
    if (p.row_stride == 1u) return g;
    const uint32_t row = g / p.row_len;
    return (row / p.row_stride) * p.row_len + (g - row * p.row_len);
}

// Everything CS_TraceRays does after the closest hit is known (shader/CS_TraceRays.hlsl:241-292): light
// buffers, shading, the scattered ray. tid is the ray's index in this device's buffers.
// n0, n1: the ray as the call leaves it (the scattered one after a hit, the same one after a miss)
__device__ __forceinline__ void shade_ray(const TraceParams& p, const ShadeBuffers& b, uint32_t tid, float4 r0, float4 r1,
                                          const Hit& best, uint32_t best_ci, float4& n0, float4& n1) {
    n0 = r0; n1 = r1;
    const float4 o0 = b.old_rays[2 * (size_t)tid], o1 = b.old_rays[2 * (size_t)tid + 1];
    const f3 org = mk(r0.x, r0.y, r0.z), dir = mk(r0.w, r1.x, r1.y);
    const float tmin_r = r1.z, tmax_r = r1.w;

    // shader/CS_TraceRays.hlsl:116-126, 241-246
    const uint32_t pv = b.ray_pixels[tid];
    const uint32_t px = pv >> 16, py = pv & 0xffffu;
    const uint32_t off = __float_as_uint(o1.w);
    const uint32_t idx = local_ray(p, ((uint32_t)p.width * py + px) * p.rpp + off);
    // pixel data that does not belong to these rows: an out-of-range structured-buffer access of the reference reads
    // zeros and drops the write (the ray itself is scattered all the same)
    const bool lit = idx < p.num_rays;
    float4 S = make_float4(0.0f, 0.0f, 0.0f, 0.0f), E = S;
    if (lit) { S = b.scattered[idx]; E = b.emitted[idx]; }

    if (best.t != tmax_r) {
        const uint32_t ia = b.indices[best_ci], ib = b.indices[best_ci + 1], ic = b.indices[best_ci + 2];
        const rt_vertex A = b.verts[ia], B = b.verts[ib], C = b.verts[ic];
        const uint32_t gid = global_ray(p, tid);
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

        // shader/PerRayShading.hlsli:138-191. A material id past the table reads zeros.
        const f3 V = mk(-dir.x, -dir.y, -dir.z);
        const f3 H = normalize3(add(V, L));
        rt_pbr_material m;
        if (A.material_id < p.num_materials) m = b.mats[A.material_id];
        else {
            m.albedo[0] = m.albedo[1] = m.albedo[2] = m.roughness = m.f0_color[0] = m.f0_color[1] = m.f0_color[2] = 0.0f;
            m.metallic = m.emissive[0] = m.emissive[1] = m.emissive[2] = 0.0f;
            m.albedo_texture = m.roughness_texture = m.f0_texture = m.metallic_texture = m.emissive_texture = 0u;
        }
        const f3 ta = sample_texture(b.tex, p.num_textures, m.albedo_texture, tu, tv, b.atlas, p.atlas_elems);
        const f3 tr = sample_texture(b.tex, p.num_textures, m.roughness_texture, tu, tv, b.atlas, p.atlas_elems);
        const f3 tf = sample_texture(b.tex, p.num_textures, m.f0_texture, tu, tv, b.atlas, p.atlas_elems);
        const f3 tm = sample_texture(b.tex, p.num_textures, m.metallic_texture, tu, tv, b.atlas, p.atlas_elems);
        const f3 te = sample_texture(b.tex, p.num_textures, m.emissive_texture, tu, tv, b.atlas, p.atlas_elems);
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
        n0 = make_float4(org.x + dir.x * best.t, org.y + dir.y * best.t, org.z + dir.z * best.t, L.x);
        n1 = make_float4(L.y, L.z, tmin_r, tmax_r);
        b.rays[2 * (size_t)tid] = n0;
        b.rays[2 * (size_t)tid + 1] = n1;
        b.old_rays[2 * (size_t)tid] = r0;
        b.old_rays[2 * (size_t)tid + 1] = make_float4(r1.x, r1.y, r1.z, o1.w);
    }
    if (lit) {
        b.scattered[idx] = S;
        b.emitted[idx] = E;
    }
}

// sums four per-thread counters over the workgroup's present lanes into stats[0..3]
__device__ __forceinline__ void add_stats(unsigned long long* stats, unsigned long long c0, unsigned long long c1,
                                          unsigned long long c2, unsigned long long c3) {
    if (__builtin_amdgcn_ballot_w64(true) == ~0ull) {
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

// leaf triangles through the leaf's links: Indices -> Vertices (shader/CS_TraceRays.hlsl:196-214)
__device__ __forceinline__ void test_leaf(f3 org, f3 dir, float tmin_r, float tmax_r, uint32_t l0, uint32_t l1,
                                          const uint32_t* __restrict__ indices, const rt_vertex* __restrict__ verts, Hit& best,
                                          uint32_t& best_ci) {
    const uint32_t ci0 = l0 & 0x7fffffffu;
    check_hit(intersect_tri(org, dir, vpos(verts, indices[ci0]), vpos(verts, indices[ci0 + 1]), vpos(verts, indices[ci0 + 2])), tmin_r,
              tmax_r, ci0, best, best_ci);
    if (l1 != 0xffffffffu) {
        const uint32_t ci1 = l1 & 0x7fffffffu;
        check_hit(intersect_tri(org, dir, vpos(verts, indices[ci1]), vpos(verts, indices[ci1 + 1]), vpos(verts, indices[ci1 + 2])),
                  tmin_r, tmax_r, ci1, best, best_ci);
    }
}

// Left-first depth-first successor of node i of level k once its subtree is finished: climb past every
// right child (the trailing one bits of i), then step to the sibling. The sibling of the last node of a
// level with an odd node count does not exist; its index equals the level's node count and the walk
// skips it without a box test (it is odd, so the next pop climbs on). Returns true when the walk is
// back at the root level, i.e. the traversal is complete.
__device__ __forceinline__ bool pop_node(uint32_t& k, uint32_t& i, uint32_t root_level) {
    const uint32_t c = (uint32_t)__builtin_ctz(~i);
    i = (i >> c) + 1u;
    k += c;
    return k >= root_level;
}

// ---- simple form: one thread per ray through traversal and shading (also the RT_USE_BVH = 0 path) ----
template <bool USE_BVH, bool STATS, bool DIV = false>
__global__ __launch_bounds__(256) void k_trace(TraceParams p, ShadeBuffers b, const float4* __restrict__ nodes,
                                               unsigned long long* __restrict__ stats, uint8_t* __restrict__ missed) {
    __shared__ uint32_t s_off[33];
    __shared__ uint32_t s_n[33];
    if (USE_BVH) {
        if (threadIdx.x < 33u) { s_off[threadIdx.x] = p.lv.off[threadIdx.x]; s_n[threadIdx.x] = p.lv.n[threadIdx.x]; }
        __syncthreads();
    }
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    if (tid >= p.num_rays) return;
    // drop-missed-rays mode: a ray that missed stays as it is until new camera rays come, and would miss again
    if (missed && missed[tid]) return;

    const float4 r0 = b.rays[2 * (size_t)tid], r1 = b.rays[2 * (size_t)tid + 1];
    const f3 org = mk(r0.x, r0.y, r0.z), dir = mk(r0.w, r1.x, r1.y);
    const float tmin_r = r1.z, tmax_r = r1.w;
    Hit best;
    best.t = tmax_r; best.u = 0.0f; best.v = 0.0f; best.w = 0.0f;
    uint32_t best_ci = 0u;
    uint32_t n_nodes = 0u, n_tris = 0u;

    if (!USE_BVH) {
        // shader/CS_TraceRays.hlsl:160-175
        for (uint32_t ci = 0; ci < p.num_indices; ci += 3u) {
            Hit h = intersect_tri(org, dir, vpos(b.verts, b.indices[ci]), vpos(b.verts, b.indices[ci + 1]), vpos(b.verts, b.indices[ci + 2]));
            check_hit(h, tmin_r, tmax_r, ci, best, best_ci);
            if (STATS) n_tris++;
        }
    } else {
        // shader/CS_TraceRays.hlsl:179-237, walked without a stack
        const f3 inv_d = DIV ? dir : mk(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
        const uint32_t root_level = p.lv.count - 1u;
        if (STATS) n_nodes++;
        // the trunk is entered on any box hit; every other node also has to start before the best hit
        if (slab<DIV>(org, inv_d, tmin_r, tmax_r, nodes[0], nodes[1]) != 1e30f) {
            uint32_t k = root_level - 1u, i = 0u;
            for (;;) {
                if (i < s_n[k]) {
                    const size_t node = (size_t)s_off[k] + i;
                    const float4 lo = nodes[2 * node], hi = nodes[2 * node + 1];
                    const float bt = slab<DIV>(org, inv_d, tmin_r, tmax_r, lo, hi);
                    if (STATS) n_nodes++;
                    const bool enter = (bt != 1e30f) && (bt < best.t);
                    if (enter && k > 0u) { k--; i <<= 1; continue; }
                    if (enter) {
                        const uint32_t l0 = __float_as_uint(hi.z), l1 = __float_as_uint(hi.w);
                        test_leaf(org, dir, tmin_r, tmax_r, l0, l1, b.indices, b.verts, best, best_ci);
                        if (STATS) n_tris += (l1 != 0xffffffffu) ? 2u : 1u;
                    }
                }
                if (pop_node(k, i, root_level)) break;
            }
        }
    }
    if (STATS) add_stats(stats, n_nodes, n_tris, (best.t != tmax_r) ? 1u : 0u, 1u);
    if (missed && best.t == tmax_r) missed[tid] = 1u;
    float4 n0, n1;
    shade_ray(p, b, tid, r0, r1, best, best_ci, n0, n1);
}

// ---- wavefront form: persistent traversal kernel + convergent shading kernel ----

constexpr uint32_t kNoRay = 0xffffffffu;

typedef float v2f __attribute__((ext_vector_type(2)));

// shader/CS_TraceRays.hlsl:81-97 for both boxes of an interleaved sibling pair at once (same operations per
// box as slab(); the subtractions and multiplications cover the two boxes with one packed instruction each)
template <bool DIV = false>
__device__ __forceinline__ void slab_pair(f3 o, f3 inv_d, float tmin_r, float tmax_r, float4 q0, float4 q1, float4 q2, float& bt0,
                                          float& bt1) {
    const v2f ox = {o.x, o.x}, oy = {o.y, o.y}, oz = {o.z, o.z};
    const v2f ix = {inv_d.x, inv_d.x}, iy = {inv_d.y, inv_d.y}, iz = {inv_d.z, inv_d.z};
    v2f ax = v2f{q0.x, q0.y} - ox, ay = v2f{q0.z, q0.w} - oy, az = v2f{q1.x, q1.y} - oz;
    v2f bx = v2f{q1.z, q1.w} - ox, by = v2f{q2.x, q2.y} - oy, bz = v2f{q2.z, q2.w} - oz;
    if (DIV) { ax = ax / ix; ay = ay / iy; az = az / iz; bx = bx / ix; by = by / iy; bz = bz / iz; }     // DIV: inv_d is the direction
    else { ax = ax * ix; ay = ay * iy; az = az * iz; bx = bx * ix; by = by * iy; bz = bz * iz; }
    float tn0 = fmaxf(fminf(ax.x, bx.x), fmaxf(fminf(ay.x, by.x), fminf(az.x, bz.x)));
    float tn1 = fmaxf(fminf(ax.y, bx.y), fmaxf(fminf(ay.y, by.y), fminf(az.y, bz.y)));
    const float tx0 = fminf(fmaxf(ax.x, bx.x), fminf(fmaxf(ay.x, by.x), fmaxf(az.x, bz.x)));
    const float tx1 = fminf(fmaxf(ax.y, bx.y), fminf(fmaxf(ay.y, by.y), fmaxf(az.y, bz.y)));
    if ((tx0 < tn0) || (tx0 <= 0.0f) || (tx0 < tmin_r) || (tn0 > tmax_r)) tn0 = 1e30f;
    if ((tx1 < tn1) || (tx1 <= 0.0f) || (tx1 < tmin_r) || (tn1 > tmax_r)) tn1 = 1e30f;
    bt0 = tn0; bt1 = tn1;
}

constexpr uint32_t kFifo = 8u;        // leaves a lane may have waiting for a triangle round

// shader/CS_TraceRays.hlsl:46-77 for the two triangles of a leaf at once: every quantity is a pair (triangle 0,
// triangle 1) and every operation of intersect_tri() is applied to both halves by one packed instruction
// where the hardware has one. The leaf is held interleaved, 80 bytes:
//   (a0.x a1.x a0.y a1.y) (a0.z a1.z b0.x b1.x) (b0.y b1.y b0.z b1.z) (c0.x c1.x c0.y c1.y) (c0.z c1.z ci0 ci1)
// with a, b, c the corners, ci the slot of the triangle's first index; ci1 = 0xffffffff marks a leaf that
// holds one triangle (its second half repeats the first).
__device__ __forceinline__ void intersect_leaf(f3 o, f3 d, float4 l0, float4 l1, float4 l2, float4 l3, float4 l4, Hit& h0, Hit& h1) {
    const v2f ax = {l0.x, l0.y}, ay = {l0.z, l0.w}, az = {l1.x, l1.y};
    const v2f bx = {l1.z, l1.w}, by = {l2.x, l2.y}, bz = {l2.z, l2.w};
    const v2f cx = {l3.x, l3.y}, cy = {l3.z, l3.w}, cz = {l4.x, l4.y};
    const v2f ox = {o.x, o.x}, oy = {o.y, o.y}, oz = {o.z, o.z};
    const v2f dx = {d.x, d.x}, dy = {d.y, d.y}, dz = {d.z, d.z};
    const v2f e1x = bx - ax, e1y = by - ay, e1z = bz - az;
    const v2f e2x = cx - ax, e2y = cy - ay, e2z = cz - az;
    const v2f tx = ox - ax, ty = oy - ay, tz = oz - az;
    const v2f px = dy * e2z - dz * e2y, py = dz * e2x - dx * e2z, pz = dx * e2y - dy * e2x;       // cross(d, e2)
    const v2f qx = ty * e1z - tz * e1y, qy = tz * e1x - tx * e1z, qz = tx * e1y - ty * e1x;       // cross(tv, e1)
    const v2f det = (e1x * px + e1y * py) + e1z * pz;
    const v2f inv = {(fabsf(det.x) < 1e-6f) ? 0.0f : 1.0f / det.x, (fabsf(det.y) < 1e-6f) ? 0.0f : 1.0f / det.y};
    const v2f t = ((e2x * qx + e2y * qy) + e2z * qz) * inv;
    const v2f u = ((tx * px + ty * py) + tz * pz) * inv;
    const v2f v = ((dx * qx + dy * qy) + dz * qz) * inv;
    const v2f uv = u + v;
    h0.w = rintf(fabsf(det.x) * inv.x);
    h1.w = rintf(fabsf(det.y) * inv.y);
    const bool out0 = t.x < 1e-6f || u.x < 0.0f || v.x < 0.0f || uv.x > 1.0f;
    const bool out1 = t.y < 1e-6f || u.y < 0.0f || v.y < 0.0f || uv.y > 1.0f;
    h0.t = out0 ? -1.0f : t.x; h0.u = out0 ? -1.0f : u.x; h0.v = out0 ? -1.0f : v.x;
    h1.t = out1 ? -1.0f : t.y; h1.u = out1 ? -1.0f : u.y; h1.v = out1 ? -1.0f : v.y;
}

// check_hit() without a branch; wanted = the leaf's triangle is to be tested at all
__device__ __forceinline__ void check_hit_select(bool wanted, const Hit& h, float tmin, float tmax, uint32_t ci, Hit& best,
                                                 uint32_t& best_ci) {
    const bool use = wanted && (tmin <= h.t) && (tmax > h.t) && (h.t < best.t);
    best.t = use ? h.t : best.t; best.u = use ? h.u : best.u; best.v = use ? h.v : best.v; best.w = use ? h.w : best.w;
    best_ci = use ? ci : best_ci;
}

// The walk runs on the private copy of the hierarchy: the node in slot h (trunk: 1) has the 64-byte record of
// its two children at pairs[4h .. 4h + 3] and the children in slots 2h and 2h + 1; leaf i is slot
// 2^root_level + i and owns the interleaved triangle pair sorted_tris[5i .. 5i + 4].
//
// One step of a lane handles the two children of the node it has entered: both boxes are tested at once (the
// reference tests the second one when its walk comes back to it, which it always does); of two children that
// are hit, the walk goes into the first and notes the second (one bit per depth), which it enters when it
// comes back. A lane that finds a leaf hit notes (leaf, entry distance) in a first-in first-out list and
// walks on; once enough lanes of the wave have a leaf waiting, each of them tests its oldest one, after
// comparing the noted distance with its closest hit again.
//
// This tests exactly the triangles of the reference's walk (shader/CS_TraceRays.hlsl:179-237), in its order:
// box entry distances never decrease from a node to its descendants and the closest hit only shrinks, so a
// leaf passes its delayed comparison exactly if it and all of its ancestors pass theirs in the reference's
// walk, whose order of leaves (left child first) is the order of the list. Walking on with a closest hit that
// is out of date, and entering a noted child without comparing its entry distance again as the reference
// does, only adds box tests. The counting variant does compare again and walks only while no leaf is waiting,
// which makes its box tests those of the reference's walk, too.
template <bool STATS, bool DIV = false>
__global__ __launch_bounds__(256) void k_traverse(TraceParams p, const float4* __restrict__ rays, const float4* __restrict__ nodes,
                                                  const float4* __restrict__ pairs, const float4* __restrict__ sorted_tris,
                                                  float4* __restrict__ hit_records, uint32_t* __restrict__ hit_ci,
                                                  uint32_t* __restrict__ queue, const uint2* __restrict__ order,
                                                  unsigned long long* __restrict__ stats, const uint32_t* __restrict__ live) {
    extern __shared__ uint32_t s_dyn[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    uint32_t num_rays = p.num_rays, queue_chunk = p.queue_chunk;      // (the kernel argument itself stays untouched: in registers)
    if (live) {
        // drop-missed-rays mode: the rays to walk are the first *live entries of order (k_ray_keys put the others last)
        num_rays = *live;
        if (num_rays == 0u) return;
        const uint32_t deal_block = 1u << p.deal_shift;
        queue_chunk = ((((num_rays + deal_block - 1u) >> p.deal_shift) + 7u) / 8u) * deal_block;
    }
    const uint32_t root_level = p.lv.count - 1u;
    const uint32_t first_leaf = 1u << root_level;              // slot of leaf 0
    const uint32_t first_leaf_parent = first_leaf >> 1;        // nodes from this slot on have leaves as children
    uint2* s_fifo = reinterpret_cast<uint2*>(s_dyn);                         // [kFifo][256] waiting leaves
    float* s_far = reinterpret_cast<float*>(s_dyn + kFifo * 512u);           // [depth][256] counting variant only
    const float4 root_lo = nodes[0], root_hi = nodes[1];

    uint32_t ray = kNoRay;            // index of the ray this lane is walking
    bool exhausted = false;           // the queue had no ray left for this lane
    bool walk_done = false;           // the walk is over; waiting leaves may remain
    bool popping = false;             // counting variant: the walk is on its way back to a noted child
    f3 org = mk(0.0f, 0.0f, 0.0f), dir = org, inv_d = org;
    float tmin_r = 0.0f, tmax_r = 0.0f;
    Hit best;
    best.t = 0.0f; best.u = 0.0f; best.v = 0.0f; best.w = 0.0f;
    uint32_t best_ci = 0u, h = 1u;
    uint32_t noted = 0u;              // bit d: the second child at depth d of the walk's ancestor at depth d - 1 is noted
    uint32_t wait_head = 0u, wait_n = 0u;
    uint32_t n_nodes = 0u, n_tris = 0u, n_hits = 0u, n_rays = 0u;
    uint32_t q = blockIdx.x & 7u, queues_tried = 0u;       // wave-uniform

    for (;;) {
        // 1. refill: every lane without a ray takes the next index; one atomic per wave.
        //    Eight queues, each over an eighth of the rays. Workgroups b and b + 8 are dealt to the same XCD, so
        //    the waves that share an L2 start on the same queue; a wave whose queue has run dry moves on to the
        //    next one (a speed choice only: any wave may take any ray).
        const bool need = (ray == kNoRay) && !exhausted;
        const unsigned long long need_mask = __builtin_amdgcn_ballot_w64(need);
        if (need_mask) {
            if (queues_tried >= 8u) {
                exhausted = exhausted || need;             // every queue has been seen empty
            } else {
                const uint32_t q_begin = q * queue_chunk;
                const uint32_t q_end = q_begin + queue_chunk;
                const uint32_t leader = (uint32_t)__builtin_ctzll(need_mask);
                uint32_t base = 0u;
                if (lane == leader) base = atomicAdd(queue + 8u * q, (uint32_t)__popcll(need_mask));
                base = q_begin + (uint32_t)__shfl((int)base, (int)leader, 64);
                if (base >= q_end) {                               // wave-uniform: this queue is empty
                    q = (q + 1u) & 7u;
                    queues_tried++;
                } else if (need) {
                    const uint32_t mine = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(need_mask >> 32),
                                                                           __builtin_amdgcn_mbcnt_lo((uint32_t)need_mask, 0u));
                    if (mine < q_end) {                    // lanes past the end stay hungry and try the next queue
                        // queue position -> position in the ray order (blocks of 64 positions are dealt to the eight
                        // queues in turn, so that every queue starts with long walks) -> ray
                        const uint32_t slot = mine - q_begin;
                        const uint32_t pos = (((slot >> p.deal_shift) * 8u + q) << p.deal_shift) | (slot & ((1u << p.deal_shift) - 1u));
                        if (pos < num_rays) {                  // the last block may be short: such a lane asks again
                        const uint32_t rid = order ? order[pos].y : pos;
                        const float4 r0 = rays[2 * (size_t)rid], r1 = rays[2 * (size_t)rid + 1];
                        org = mk(r0.x, r0.y, r0.z); dir = mk(r0.w, r1.x, r1.y);
                        tmin_r = r1.z; tmax_r = r1.w;
                        inv_d = DIV ? dir : mk(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
                        best.t = tmax_r; best.u = 0.0f; best.v = 0.0f; best.w = 0.0f;
                        best_ci = 0u;
                        if (STATS) { n_nodes++; n_rays++; }
                        // the trunk is entered on any box hit
                        if (slab<DIV>(org, inv_d, tmin_r, tmax_r, root_lo, root_hi) != 1e30f) {
                            ray = rid; h = 1u;
                            walk_done = false; popping = false; noted = 0u;
                        } else {
                            hit_records[rid] = make_float4(best.t, 0.0f, 0.0f, 0.0f);
                            hit_ci[rid] = 0u;
                        }
                        }
                    }
                }
            }
        }
        const unsigned long long live_mask = __builtin_amdgcn_ballot_w64(ray != kNoRay);
        if (!live_mask) {
            if (!__builtin_amdgcn_ballot_w64(!exhausted)) break;      // queue empty and nothing in flight: the wave is done
            continue;
        }

        // 2. the two children of the entered node h
        const bool can_walk = ray != kNoRay && !walk_done && (STATS ? wait_n == 0u : wait_n + 2u <= kFifo);
        bool back = STATS && popping;                              // go back to the nearest noted child
        if (can_walk && !back) {
            const float4* rec = pairs + 4 * (size_t)h;
            const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
            float bt0, bt1;
            slab_pair<DIV>(org, inv_d, tmin_r, tmax_r, q0, q1, q2, bt0, bt1);
            const bool valid1 = q0.y == q0.y;                      // a missing child is stored as NaNs
            if (STATS) n_nodes += valid1 ? 2u : 1u;
            const bool hit0 = (bt0 != 1e30f) && (bt0 < best.t);
            const bool hit1 = valid1 && (bt1 != 1e30f) && (bt1 < best.t);
            const bool leaves = h >= first_leaf_parent;
            const uint32_t child0 = 2u * h;
            if (leaves && hit0) {
                s_fifo[((wait_head + wait_n) & (kFifo - 1u)) * 256u + tid] = make_uint2(child0 - first_leaf, __float_as_uint(bt0));
                wait_n++;
            }
            if (leaves && hit1) {
                s_fifo[((wait_head + wait_n) & (kFifo - 1u)) * 256u + tid] = make_uint2(child0 + 1u - first_leaf, __float_as_uint(bt1));
                wait_n++;
            }
            const bool go = !leaves && (hit0 || hit1);
            if (!leaves && hit0 && hit1) {
                const uint32_t child_depth = 32u - (uint32_t)__builtin_clz(h);
                noted |= 1u << child_depth;
                if (STATS) s_far[child_depth * 256u + tid] = bt1;
            }
            h = go ? (child0 + (hit0 ? 0u : 1u)) : h;
            back = !go;
        }
        //    ... or, once its subtree is finished, the nearest noted second child (the counting variant first
        //    lets the leaves found in this step be tested, then compares the child's entry distance again)
        if (can_walk && back && (!STATS || wait_n == 0u)) {
            if (!noted) {
                walk_done = true;
                popping = false;
            } else {
                const uint32_t d_to = 31u - (uint32_t)__builtin_clz(noted);
                const uint32_t d_at = 31u - (uint32_t)__builtin_clz(h);
                noted ^= 1u << d_to;
                h = (h >> (d_at - d_to)) | 1u;
                popping = STATS ? !(s_far[d_to * 256u + tid] < best.t) : false;
            }
        } else if (STATS && can_walk) {
            popping = back;
        }

        // 3. triangle round once enough lanes have a leaf waiting (or nobody can walk any more)
        const unsigned long long waiting_mask = __builtin_amdgcn_ballot_w64(wait_n > 0u);
        if (waiting_mask) {
            const unsigned long long walking_mask =
                __builtin_amdgcn_ballot_w64(ray != kNoRay && !walk_done && (STATS ? wait_n == 0u : wait_n + 2u <= kFifo));
            if ((uint32_t)__popcll(waiting_mask) >= p.leaf_threshold || !walking_mask) {
                if (wait_n > 0u) {
                    const uint2 e = s_fifo[wait_head * 256u + tid];
                    wait_head = (wait_head + 1u) & (kFifo - 1u);
                    wait_n--;
                    if (__uint_as_float(e.y) < best.t) {
                        const float4* t = sorted_tris + RT_LEAF_STRIDE * (size_t)e.x;
                        const float4 l0 = t[0], l1 = t[1], l2 = t[2], l3 = t[3], l4 = t[4];
                        const uint32_t ci0 = __float_as_uint(l4.z), ci1 = __float_as_uint(l4.w);
                        const bool two = ci1 != 0xffffffffu;
                        Hit h0, h1;
                        intersect_leaf(org, dir, l0, l1, l2, l3, l4, h0, h1);
                        check_hit(h0, tmin_r, tmax_r, ci0, best, best_ci);
                        if (two) check_hit(h1, tmin_r, tmax_r, ci1, best, best_ci);
                        if (STATS) n_tris += two ? 2u : 1u;
                    }
                }
            }
        }

        // 4. a finished ray leaves its closest hit behind and frees the lane
        if (ray != kNoRay && walk_done && wait_n == 0u) {
            hit_records[ray] = make_float4(best.t, best.u, best.v, best.w);
            hit_ci[ray] = best_ci;
            if (STATS && best.t != tmax_r) n_hits++;
            ray = kNoRay;
        }
    }
    if (STATS) add_stats(stats, n_nodes, n_tris, n_hits, n_rays);
}

// v_min_f32 / v_max_f32 and their three-operand forms, issued as written: the compiler otherwise guards every
// fminf/fmaxf whose operand it cannot prove canonical with an extra instruction. Semantics are the ones the
// rest of the library states for min/max: the operand that is not NaN wins, -0 < +0.
__device__ __forceinline__ float vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmin3(float a, float b, float c) {
    float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;
}

// A lane's state in k_traverse_quad is one word w: the number of leaves it has waiting (0..kFifo), + kOver once its
// walk is over, or kIdle (no ray) or kDry (no ray and the queues are empty), so that every decision the wave takes
// together is one comparison per lane: walking w <= kFifo - 4, waiting (w & 15) != 0, finished w == kOver.
constexpr uint32_t kOver = 16u, kIdle = 32u, kDry = 64u;
static_assert(kFifo <= 15u, "the waiting count shares w with the state flags");

// ---- two levels per step: the walk on the four-grandchildren records ----
//
// Past the L2 the memory system delivers random reads at a fixed rate of 128-byte lines, whatever part of the line is
// used, and in front of it the L1 path takes about 1.4 lane-loads of 16 bytes per clock and CU; both bound the walk.
// One step of this kernel reads the 64-byte record that holds boxes of the entered node's four grandchildren (slots
// 4h .. 4h + 3; three loads per lane) and tests all four, instead of one 64-byte record for the two children and one
// more for every child that is hit.
//
// The boxes of those records are binary16 boxes that CONTAIN the true ones, and that is enough:
//  * every operation of the slab test is monotone, so a ray that passes a true box (entry distance below the closest
//    hit) passes every box that contains it - in particular the widened boxes of the node and of all its ancestors.
//    The walk therefore reaches every leaf the reference's walk reaches, in the same left-first order, plus some it
//    does not. (A direction component of exactly zero gives the reference's test 0 * inf = NaN on a box plane through
//    the origin; the widened test uses a huge finite reciprocal for such a component, which keeps it monotone.)
//  * a leaf that is reached goes into the lane's in-order list with its (widened) entry distance. When its turn comes,
//    the leaf's exact box is recomputed from its triangles - the build takes that box as the minimum and maximum of
//    the same six corners - and tested exactly as the reference tests it, against the closest hit of that moment.
//    If it passes, every ancestor's exact box passed in the reference's walk (it contains the leaf's box, and the
//    closest hit was no smaller then), so the reference visits this leaf, at this point of its order; if it fails,
//    the reference does not test its triangles either. The triangles tested, their order and the closest hit they
//    are compared with are the reference's; the box tests in between only decide how much extra work is done.
// Noted nodes: one nibble per step taken down; bit j of the lowest nibble = slot (h & ~3) | j waits, in order.
// A hierarchy with an odd number of levels below the trunk starts with the trunk's two children (pair record 1, exact).
// lim = tmin if tmin > 0, else the smallest positive float: "tfar <= 0 || tfar < tmin" of the reference's slab test
// (shader/CS_TraceRays.hlsl:81-97) is "tfar < lim" for every float, and its "tnear > tmax" never holds where
// tnear < closest <= tmax does. Byte offsets into the record and leaf arrays are 32-bit: the launcher uses this
// kernel only below 4 GiB each.
// (binary16 pair d) - o for both halves: v_fma_mix_f32 converts its binary16 operand on the way in, and b * 1.0 - o
// rounds once, exactly as the subtraction b - o does
__device__ __forceinline__ v2f half_pair_minus(uint32_t d, float o) {
    float lo, hi;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(d), "v"(o));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(hi) : "v"(d), "v"(o));
    return v2f{lo, hi};
}

// exact box of a leaf (minimum / maximum of its six corners, from +-1e30 as shader/CS_BVHBuildLeaves.hlsl:31-52 has
// it) and the reference's test of it; true if the leaf's triangles are to be tested
__device__ __forceinline__ bool leaf_box_passes(f3 o, f3 inv_d, float lim, float closest, float4 l0, float4 l1, float4 l2, float4 l3,
                                                float4 l4) {
    const float big = 1e30f, small = -1e30f;
    const float mnx = vmin3(vmin3(l0.x, l0.y, l1.z), vmin3(l1.w, l3.x, l3.y), big);
    const float mny = vmin3(vmin3(l0.z, l0.w, l2.x), vmin3(l2.y, l3.z, l3.w), big);
    const float mnz = vmin3(vmin3(l1.x, l1.y, l2.z), vmin3(l2.w, l4.x, l4.y), big);
    const float mxx = vmax3(vmax3(l0.x, l0.y, l1.z), vmax3(l1.w, l3.x, l3.y), small);
    const float mxy = vmax3(vmax3(l0.z, l0.w, l2.x), vmax3(l2.y, l3.z, l3.w), small);
    const float mxz = vmax3(vmax3(l1.x, l1.y, l2.z), vmax3(l2.w, l4.x, l4.y), small);
    const v2f tx = (v2f{mnx, mxx} - v2f{o.x, o.x}) * v2f{inv_d.x, inv_d.x};
    const v2f ty = (v2f{mny, mxy} - v2f{o.y, o.y}) * v2f{inv_d.y, inv_d.y};
    const v2f tz = (v2f{mnz, mxz} - v2f{o.z, o.z}) * v2f{inv_d.z, inv_d.z};
    const float tn = vmax3(vmin(tx.x, tx.y), vmin(ty.x, ty.y), vmin(tz.x, tz.y));
    const float tf = vmin3(vmax(tx.x, tx.y), vmax(ty.x, ty.y), vmax(tz.x, tz.y));
    return !(tf < vmax(tn, lim)) && (tn != 1e30f) && (tn < closest);
}

// the same test without the closest hit: the distance at which the ray enters the leaf's exact box, or +infinity
// if the reference's test lets it pass by
__device__ __forceinline__ float leaf_box_entry(f3 o, f3 inv_d, float lim, float4 l0, float4 l1, float4 l2, float4 l3, float4 l4) {
    const float big = 1e30f, small = -1e30f;
    const float mnx = vmin3(vmin3(l0.x, l0.y, l1.z), vmin3(l1.w, l3.x, l3.y), big);
    const float mny = vmin3(vmin3(l0.z, l0.w, l2.x), vmin3(l2.y, l3.z, l3.w), big);
    const float mnz = vmin3(vmin3(l1.x, l1.y, l2.z), vmin3(l2.w, l4.x, l4.y), big);
    const float mxx = vmax3(vmax3(l0.x, l0.y, l1.z), vmax3(l1.w, l3.x, l3.y), small);
    const float mxy = vmax3(vmax3(l0.z, l0.w, l2.x), vmax3(l2.y, l3.z, l3.w), small);
    const float mxz = vmax3(vmax3(l1.x, l1.y, l2.z), vmax3(l2.w, l4.x, l4.y), small);
    const v2f tx = (v2f{mnx, mxx} - v2f{o.x, o.x}) * v2f{inv_d.x, inv_d.x};
    const v2f ty = (v2f{mny, mxy} - v2f{o.y, o.y}) * v2f{inv_d.y, inv_d.y};
    const v2f tz = (v2f{mnz, mxz} - v2f{o.z, o.z}) * v2f{inv_d.z, inv_d.z};
    const float tn = vmax3(vmin(tx.x, tx.y), vmin(ty.x, ty.y), vmin(tz.x, tz.y));
    const float tf = vmin3(vmax(tx.x, tx.y), vmax(ty.x, ty.y), vmax(tz.x, tz.y));
    return (!(tf < vmax(tn, lim)) && (tn != 1e30f)) ? tn : __builtin_inff();
}

// ---- a short list of rays, walked in parts ----
//
// The rays k_traverse_spatial lists are few - a few dozen of two million on the reference's own room scene - and their
// walk in the reference's order is long: hundreds of dependent steps, a third of a millisecond for one lane, however
// few the rays. SPLIT deals such a list out in parts: lane j of a ray walks only the subtree of node j at depth
// `depth` (up to 64 of them), with the same step, the same list of waiting leaves and the same exact test of a leaf's
// box - but it cannot know the closest hit the walk would have brought along from the subtrees before it. So it keeps
// a bound instead that is never below that closest hit, and writes down every triangle that could become the closest
// hit under ANY incoming value - its candidates, in the order of the walk:
//  * bound starts at the ray's far end and falls to max(hit distance, entry distance of the leaf's box) with every
//    candidate. Whatever the closest hit was when the reference's walk came to that triangle, afterwards it is at most
//    that: if the leaf's box let the ray in before the closest hit, the triangle was tested and the closest hit is at
//    most its distance; if not, the closest hit was at most the box's entry distance already.
//  * a leaf is tested if its box lets the ray in before the bound, a triangle noted if its distance is below the bound:
//    both hold for everything the reference's walk tests and accepts, whose closest hit is at most the bound.
// k_listed_merge then goes through the parts in order (the order of the subtrees is the order of the walk) and applies
// the reference's two comparisons to every candidate with the closest hit it has by then - a leaf's triangles are
// tested if the box's entry distance is below the closest hit before the leaf's first triangle, a hit is kept if it is
// strictly closer -, which is the reference's walk with the triangles left out that it would have rejected anyway.
// A part that runs out of room for candidates puts its ray on a second list, which the plain form walks whole.
struct ListedSplit {
    uint32_t cap;                       // lists of up to this many rays are walked in parts (0: no list is)
    uint32_t depth;                     // the parts are the subtrees of the nodes at this depth below the trunk
    float4* cand;                       // [cap << depth][RT_SPLIT_CANDIDATES][2]
    uint32_t* cand_count;               // [cap << depth]
    const uint2* again;                 // plain form: a second list, walked after the first ...
    const uint32_t* again_count;        // ... of this length (nullptr: none)
    uint32_t* seen;                     // host-visible word that receives the length of the first list (nullptr: none)
};

template <bool SPLIT>
__global__ __launch_bounds__(256) void k_traverse_quad(TraceParams p, const float4* __restrict__ rays, const float4* __restrict__ nodes,
                                                       const float4* __restrict__ pairs, const float4* __restrict__ quads,
                                                       const float4* __restrict__ sorted_tris, float4* __restrict__ hit_records,
                                                       uint32_t* __restrict__ hit_ci, uint32_t* __restrict__ queue,
                                                       const uint2* __restrict__ order, const uint32_t* __restrict__ listed,
                                                       const ListedSplit split) {
    extern __shared__ uint32_t s_dyn[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    uint32_t num_rays = p.num_rays, queue_chunk = p.queue_chunk;      // (the kernel argument itself stays untouched: in registers)
    uint32_t first_list = 0u;                                         // plain form with two lists: entries of the first
    if (listed) {
        // the rays are the first *listed entries of order: a list another kernel of the same stream has filled
        num_rays = *listed;
        if (!SPLIT && split.seen && blockIdx.x == 0u && tid == 0u) *split.seen = num_rays;
        if (SPLIT) {
            if (num_rays == 0u || num_rays > split.cap) return;       // (a long list is the plain form's)
            num_rays <<= split.depth;
        } else {
            if (num_rays <= split.cap) num_rays = 0u;                 // (a short list has been walked in parts)
            first_list = num_rays;
            if (split.again_count) num_rays += *split.again_count;
        }
        if (num_rays == 0u) return;
        const uint32_t deal_block = 1u << p.deal_shift;
        queue_chunk = ((((num_rays + deal_block - 1u) >> p.deal_shift) + 7u) / 8u) * deal_block;
    }
    const uint32_t root_level = p.lv.count - 1u;               // >= 2
    const uint32_t first_leaf = 1u << root_level;              // slot of leaf 0
    const uint32_t first_leaf_grandparent = first_leaf >> 2;   // nodes from this slot on have leaves as grandchildren
    const bool odd_first = (root_level & 1u) != 0u;
    uint2* s_fifo = reinterpret_cast<uint2*>(s_dyn) + tid;     // [kFifo][256] waiting leaves, this lane's column
    const float4 root_lo = nodes[0], root_hi = nodes[1];
    const char* quad_bytes = reinterpret_cast<const char*>(quads);
    const char* leaf_bytes = reinterpret_cast<const char*>(sorted_tris);

    uint32_t ray = 0u;                // index of the ray this lane is walking (while w < kIdle)
    uint32_t w = kIdle;
    f3 org = mk(0.0f, 0.0f, 0.0f), dir = org, inv_d = org;
    f3 inv_w = org;                   // reciprocals used with the widened boxes: finite where the direction is zero
    bool neg_x = false, neg_y = false, neg_z = false;      // the ray travels downwards on that axis
    float tmin_r = 0.0f, tmax_r = 0.0f, lim = 0.0f;
    Hit best;
    best.t = 0.0f; best.u = 0.0f; best.v = 0.0f; best.w = 0.0f;
    uint32_t best_ci = 0u;
    uint32_t h = 1u;                  // slot of the entered node
    unsigned long long trail = 0ull;
    uint32_t wait_head = 0u;
    uint32_t q = blockIdx.x & 7u, queues_tried = 0u;       // wave-uniform
    uint32_t noted = 0u;              // SPLIT: candidates of this lane's part so far (ray = number of the part; best.t = the bound)

    for (;;) {
        // 1. refill: every lane without a ray takes the next index; one atomic per wave (see k_traverse)
        //    A refill stalls the whole wave for an atomic and two dependent loads, so it waits until several lanes are
        //    free, unless nobody could walk meanwhile.
        const unsigned long long need_mask = __builtin_amdgcn_ballot_w64(w == kIdle);
        if (need_mask && ((uint32_t)__popcll(need_mask) >= p.refill_threshold || !__builtin_amdgcn_ballot_w64(w <= kFifo - 4u))) {
            if (queues_tried >= 8u) {
                if (w == kIdle) w = kDry;
            } else {
                const uint32_t q_begin = q * queue_chunk;
                const uint32_t q_end = q_begin + queue_chunk;
                const uint32_t leader = (uint32_t)__builtin_ctzll(need_mask);
                uint32_t base = 0u;
                if (lane == leader) base = atomicAdd(queue + 8u * q, (uint32_t)__popcll(need_mask));
                base = q_begin + (uint32_t)__shfl((int)base, (int)leader, 64);
                if (base >= q_end) {                               // wave-uniform: this queue is empty
                    q = (q + 1u) & 7u;
                    queues_tried++;
                } else if (w == kIdle) {
                    const uint32_t mine = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(need_mask >> 32),
                                                                           __builtin_amdgcn_mbcnt_lo((uint32_t)need_mask, 0u));
                    if (mine < q_end) {
                        const uint32_t slot = mine - q_begin;
                        const uint32_t pos = (((slot >> p.deal_shift) * 8u + q) << p.deal_shift) | (slot & ((1u << p.deal_shift) - 1u));
                        if (pos < num_rays) {
                            uint32_t rid;
                            if (SPLIT) rid = order[pos >> split.depth].y;
                            else rid = !order ? pos : (pos < first_list || !split.again_count) ? order[pos].y : split.again[pos - first_list].y;
                            const float4 r0 = rays[2 * (size_t)rid], r1 = rays[2 * (size_t)rid + 1];
                            org = mk(r0.x, r0.y, r0.z); dir = mk(r0.w, r1.x, r1.y);
                            tmin_r = r1.z; tmax_r = r1.w;
                            lim = (tmin_r > 0.0f) ? tmin_r : __uint_as_float(1u);
                            inv_d = mk(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
                            inv_w = mk(dir.x == 0.0f ? copysignf(3.0e38f, inv_d.x) : inv_d.x, dir.y == 0.0f ? copysignf(3.0e38f, inv_d.y) : inv_d.y,
                                       dir.z == 0.0f ? copysignf(3.0e38f, inv_d.z) : inv_d.z);
                            neg_x = inv_w.x < 0.0f; neg_y = inv_w.y < 0.0f; neg_z = inv_w.z < 0.0f;
                            best.t = tmax_r; best.u = 0.0f; best.v = 0.0f; best.w = 0.0f;
                            best_ci = 0u;
                            // the trunk is entered on any box hit
                            bool walk = slab(org, inv_d, tmin_r, tmax_r, root_lo, root_hi) != 1e30f;
                            h = 1u; trail = 0ull;
                            if (SPLIT) {
                                // this lane's part: the subtree of node `part` at the parts' depth, if there is such a node
                                const uint32_t part = pos & ((1u << split.depth) - 1u);
                                walk = walk && part < p.lv.n[root_level - split.depth];
                                h = (1u << split.depth) + part;
                                noted = 0u;
                                if (walk) { ray = pos; w = 0u; }
                                else split.cand_count[pos] = 0u;
                            } else
                            if (walk && odd_first) {
                                // the trunk's two children, exact boxes, from the nodes in the reference's layout
                                const float4* c = nodes + 2 * (size_t)p.trunk_children;
                                const float bt0 = slab(org, inv_d, tmin_r, tmax_r, c[0], c[1]), bt1 = slab(org, inv_d, tmin_r, tmax_r, c[2], c[3]);
                                const bool hit0 = (bt0 != 1e30f) && (bt0 < best.t);
                                const bool hit1 = (bt1 != 1e30f) && (bt1 < best.t);
                                walk = hit0 || hit1;
                                h = hit0 ? 2u : 3u;
                                trail = (hit0 && hit1) ? 8ull : 0ull;
                            }
                            if (SPLIT) {
                            } else if (walk) {
                                ray = rid; w = 0u;
                            } else {
                                hit_records[rid] = make_float4(best.t, 0.0f, 0.0f, 0.0f);
                                hit_ci[rid] = 0u;
                            }
                        }
                    }
                }
            }
        }
        if (!__builtin_amdgcn_ballot_w64(w < kIdle)) {
            if (!__builtin_amdgcn_ballot_w64(w != kDry)) break;       // queues empty and nothing in flight: the wave is done
            continue;
        }

        // 2. walk until a triangle round is due, a lane has finished its ray, or nobody can walk
        unsigned long long waiting_mask, walking_mask;
        for (;;) {
            if (w <= kFifo - 4u) {
                const uint4* rec = reinterpret_cast<const uint4*>(quad_bytes + (h << 6));
                const uint4 u0 = rec[0], u1 = rec[1], u2 = rec[2];
                const v2f ix = {inv_w.x, inv_w.x}, iy = {inv_w.y, inv_w.y}, iz = {inv_w.z, inv_w.z};
                // first pair (slots 4h, 4h + 1): dwords 0 - 5 = min x y z, max x y z; second pair: dwords 6 - 11.
                // The plane a ray meets first on an axis is the minimum if it travels upwards, else the maximum.
                const v2f anx = half_pair_minus(neg_x ? u0.w : u0.x, org.x) * ix, afx = half_pair_minus(neg_x ? u0.x : u0.w, org.x) * ix;
                const v2f any = half_pair_minus(neg_y ? u1.x : u0.y, org.y) * iy, afy = half_pair_minus(neg_y ? u0.y : u1.x, org.y) * iy;
                const v2f anz = half_pair_minus(neg_z ? u1.y : u0.z, org.z) * iz, afz = half_pair_minus(neg_z ? u0.z : u1.y, org.z) * iz;
                const v2f bnx = half_pair_minus(neg_x ? u2.y : u1.z, org.x) * ix, bfx = half_pair_minus(neg_x ? u1.z : u2.y, org.x) * ix;
                const v2f bny = half_pair_minus(neg_y ? u2.z : u1.w, org.y) * iy, bfy = half_pair_minus(neg_y ? u1.w : u2.z, org.y) * iy;
                const v2f bnz = half_pair_minus(neg_z ? u2.w : u2.x, org.z) * iz, bfz = half_pair_minus(neg_z ? u2.x : u2.w, org.z) * iz;
                const float tn0 = vmax3(anx.x, any.x, anz.x), tx0 = vmin3(afx.x, afy.x, afz.x);
                const float tn1 = vmax3(anx.y, any.y, anz.y), tx1 = vmin3(afx.y, afy.y, afz.y);
                const float tn2 = vmax3(bnx.x, bny.x, bnz.x), tx2 = vmin3(bfx.x, bfy.x, bfz.x);
                const float tn3 = vmax3(bnx.y, bny.y, bnz.y), tx3 = vmin3(bfx.y, bfy.y, bfz.y);
                const bool hit0 = !(tx0 < vmax(tn0, lim)) && (tn0 < best.t);
                const bool hit1 = !(tx1 < vmax(tn1, lim)) && (tn1 < best.t);
                const bool hit2 = !(tx2 < vmax(tn2, lim)) && (tn2 < best.t);
                const bool hit3 = !(tx3 < vmax(tn3, lim)) && (tn3 < best.t);
                const uint32_t m = (hit0 ? 1u : 0u) | (hit1 ? 2u : 0u) | (hit2 ? 4u : 0u) | (hit3 ? 8u : 0u);
                if (h < first_leaf_grandparent && m != 0u) {
                    trail = (trail << 4) | (unsigned long long)(m & (m - 1u));
                    h = (h << 2) | (uint32_t)__builtin_ctz(m);
                } else {
                    if (h >= first_leaf_grandparent) {
                        const uint32_t leaf0 = 4u * h - first_leaf;
                        if (hit0) { s_fifo[((wait_head + w) & (kFifo - 1u)) * 256u] = make_uint2(leaf0, __float_as_uint(tn0)); w++; }
                        if (hit1) { s_fifo[((wait_head + w) & (kFifo - 1u)) * 256u] = make_uint2(leaf0 + 1u, __float_as_uint(tn1)); w++; }
                        if (hit2) { s_fifo[((wait_head + w) & (kFifo - 1u)) * 256u] = make_uint2(leaf0 + 2u, __float_as_uint(tn2)); w++; }
                        if (hit3) { s_fifo[((wait_head + w) & (kFifo - 1u)) * 256u] = make_uint2(leaf0 + 3u, __float_as_uint(tn3)); w++; }
                    }
                    // back to the first noted node of the nearest level that has one, or the walk is over
                    if (trail == 0ull) {
                        w |= kOver;
                    } else {
                        const uint32_t z = (uint32_t)__builtin_ctzll(trail);
                        const uint32_t up = z & ~3u;
                        unsigned long long t2 = trail >> up;
                        t2 &= t2 - 1ull;
                        h = ((h >> (up >> 1)) & ~3u) | (z & 3u);
                        trail = t2;
                    }
                }
            }
            waiting_mask = __builtin_amdgcn_ballot_w64((w & 15u) != 0u);
            walking_mask = __builtin_amdgcn_ballot_w64(w <= kFifo - 4u);
            if (!walking_mask || (uint32_t)__popcll(waiting_mask) >= p.leaf_threshold) break;
            if ((uint32_t)__popcll(__builtin_amdgcn_ballot_w64(w >= kOver && w != kDry)) >= p.refill_threshold) break;
        }

        // 3. triangle round once enough lanes have a leaf waiting (or nobody can walk any more)
        if (waiting_mask && ((uint32_t)__popcll(waiting_mask) >= p.leaf_threshold || !walking_mask)) {
            if ((w & 15u) != 0u) {
                const uint2 e = s_fifo[wait_head * 256u];
                wait_head = (wait_head + 1u) & (kFifo - 1u);
                w--;
                if (__uint_as_float(e.y) < best.t) {
                    const float4* t = reinterpret_cast<const float4*>(leaf_bytes + e.x * 128u);
                    const float4 l0 = t[0], l1 = t[1], l2 = t[2], l3 = t[3], l4 = t[4];
                    // Both triangles first, in one straight run of instructions. The leaf's exact box only matters if one
                    // of them would become the closest hit, which few leaves see: only then is the box worked out.
                    const uint32_t ci0 = __float_as_uint(l4.z), ci1 = __float_as_uint(l4.w);
                    Hit h0, h1;
                    intersect_leaf(org, dir, l0, l1, l2, l3, l4, h0, h1);
                    const bool may0 = (tmin_r <= h0.t) && (tmax_r > h0.t) && (h0.t < best.t);
                    const bool may1 = (ci1 != 0xffffffffu) && (tmin_r <= h1.t) && (tmax_r > h1.t) && (h1.t < best.t);
                    if (SPLIT) {
                        if (may0 || may1) {
                            const float entry = leaf_box_entry(org, inv_d, lim, l0, l1, l2, l3, l4);      // +infinity: the ray passes by
                            if (entry < best.t) {
                                float4* mine = split.cand + (size_t)ray * (2u * RT_SPLIT_CANDIDATES);
                                if (may0) {
                                    if (noted < RT_SPLIT_CANDIDATES) {
                                        mine[2u * noted] = make_float4(h0.t, h0.u, h0.v, h0.w);
                                        mine[2u * noted + 1u] = make_float4(__uint_as_float(ci0), entry, __uint_as_float(2u * e.x), 0.0f);
                                    }
                                    noted++;
                                    best.t = fminf(best.t, fmaxf(h0.t, entry));
                                }
                                if (may1 && h1.t < best.t) {
                                    if (noted < RT_SPLIT_CANDIDATES) {
                                        mine[2u * noted] = make_float4(h1.t, h1.u, h1.v, h1.w);
                                        mine[2u * noted + 1u] = make_float4(__uint_as_float(ci1), entry, __uint_as_float(2u * e.x + 1u), 0.0f);
                                    }
                                    noted++;
                                    best.t = fminf(best.t, fmaxf(h1.t, entry));
                                }
                            }
                        }
                    } else if (may0 || may1) {
                        const bool wanted = leaf_box_passes(org, inv_d, lim, best.t, l0, l1, l2, l3, l4);
                        check_hit_select(wanted, h0, tmin_r, tmax_r, ci0, best, best_ci);
                        check_hit_select(wanted && ci1 != 0xffffffffu, h1, tmin_r, tmax_r, ci1, best, best_ci);
                    }
                }
            }
        }

        // 4. a finished ray leaves its closest hit behind and frees the lane
        if (w == kOver) {
            if (SPLIT) {
                split.cand_count[ray] = noted;
            } else {
                hit_records[ray] = make_float4(best.t, best.u, best.v, best.w);
                hit_ci[ray] = best_ci;
            }
            w = kIdle;
        }
    }
}

// The parts of every ray of a short list, put together (see ListedSplit): one lane per ray. A ray one of whose parts
// ran out of room goes on the second list (again / again_count), for the plain form of the walk.
__global__ __launch_bounds__(256) void k_listed_merge(const float4* __restrict__ rays, const uint2* __restrict__ order,
                                                      const uint32_t* __restrict__ listed, const ListedSplit split,
                                                      float4* __restrict__ hit_records, uint32_t* __restrict__ hit_ci,
                                                      uint2* __restrict__ again, uint32_t* __restrict__ again_count) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t count = *listed;
    if (split.seen && i == 0u) *split.seen = count;
    if (count > split.cap || i >= count) return;
    const uint32_t rid = order[i].y;
    const uint32_t parts = 1u << split.depth, first = i << split.depth;
    Hit best;
    best.t = rays[2 * (size_t)rid + 1].w; best.u = 0.0f; best.v = 0.0f; best.w = 0.0f;      // the ray's far end: no hit yet
    uint32_t best_ci = 0u;
    bool complete = true;
    uint32_t leaf = 0xffffffffu;          // the leaf the last candidate belonged to, and whether the walk tests its triangles
    bool open = false;
    for (uint32_t part = 0u; part < parts; part++) {
        const uint32_t n = split.cand_count[first + part];
        if (n > RT_SPLIT_CANDIDATES) { complete = false; break; }
        const float4* c = split.cand + (size_t)(first + part) * (2u * RT_SPLIT_CANDIDATES);
        for (uint32_t k = 0u; k < n; k++) {
            const float4 hit = c[2u * k], more = c[2u * k + 1u];
            const uint32_t place = __float_as_uint(more.z);
            if ((place >> 1) != leaf) {                               // shader/CS_TraceRays.hlsl:196: the box, against the closest hit so far
                leaf = place >> 1;
                open = more.y < best.t;
            }
            if (open && hit.x < best.t) {                             // shader/CS_TraceRays.hlsl:110: strictly closer
                best.t = hit.x; best.u = hit.y; best.v = hit.z; best.w = hit.w;
                best_ci = __float_as_uint(more.x);
            }
        }
    }
    if (complete) {
        hit_records[rid] = make_float4(best.t, best.u, best.v, best.w);
        hit_ci[rid] = best_ci;
    } else {
        again[atomicAdd(again_count, 1u)] = make_uint2(0u, rid);
    }
}

// ---- the walk on a hierarchy of the library's own ----
//
// The reference's hierarchy is what its Morton codes make it: codes whose bits overlap (CS_GenerateMortonCodes.hlsl:30-38)
// put triangles from all over the scene next to each other, its boxes are large, and its walk tests thousands of
// them per ray. What the reference's walk RETURNS, though, can be said without its hierarchy: it goes through the
// leaves in their order (pairs of neighbours of the sorted triangles), tests a leaf's triangles if the leaf's box lets
// the ray in before the closest hit so far, and keeps a hit that is strictly closer. Leaving aside box entries that lie
// behind their own triangles' hits, that is: of all triangles whose leaf box (in the reference's pairing) lets the
// ray in, the one with the smallest distance, the earlier in the reference's order among equal distances.
// k_traverse_spatial finds that triangle on a second hierarchy, built by the same kernels from non-overlapping codes
// of the same centroids (leaf slots laid out by Morton cell - see k_cell_choose in rt_build.hip -, the eight triangles
// under a record paired into leaves by surface area), with the step of k_traverse_quad:
//  * triangles whose box is a large part of the scene's (outliers: a wall, a floor) are not in that hierarchy, where they
//    would make every node above them as large as themselves - a dozen such triangles among a million small ones cost
//    every ray some fifty steps. They are tested from a list, by the whole wave, when it fills its stock of rays; what they
//    leave (a closest hit, a reach) is where the walk starts from, and a ray that the rest of the scene cannot reach
//    within that is not walked at all;
//  * a box is entered while its entry distance does not exceed "reach" (not: is below the closest hit), so that
//    hits at the same distance are all seen and the order of the reference decides between them;
//  * for a triangle that is hit, the reference's leaf of it (its place in the reference's order / 2) is fetched and
//    that leaf's box tested as the reference tests it; a triangle whose box there lets the ray in late - entry
//    distance above the hit distance, by rounding or by a hit distance far off for a ray nearly in the triangle's
//    plane - is kept with reach = that entry distance, and every other hit up to reach is remembered;
//  * a ray for which the order of the reference's walk could matter (another hit between the closest hit and the
//    entry of its box, or a closer hit whose box entry is beyond reach) goes on a list, and k_traverse_quad walks the
//    listed rays again on the reference's hierarchy.
// What stays out of reach of this argument are hits that are numerically false: a triangle whose computed hit distance
// is closer than the closest hit kept here although its boxes are entered beyond it by more than the margin boxes are
// entered with (2^-8 of the distance), or whose hit test accepts a ray that passes beside its box - near-degenerate
// triangles seen edge-on, with a determinant just above the reference's cut-off (the build gives the boxes of slivers
// some room for this reason). The reference may keep such a hit, this kernel never sees it. Differences of a rounding -
// triangles in the faces of flat boxes, overlapping coplanar triangles - are inside the margin and end on the list
// (scripts/fuzz_walks.py compares the two walks over random scenes of these kinds).

// origin, direction, reciprocals, tmin, tmax, ray, first node, noted sibling (+ bit 31: walk again in the reference's
// order), and what the outliers have left: closest hit (t, u, v, facing), its index slot and place, reach, runner-up
constexpr uint32_t kStockFields = 22u;

// What a lane of k_traverse_spatial knows about its ray's hits. Beyond the closest hit: its place in the reference's
// order, the distance up to which boxes are still entered (the larger of the hit's distance and the entry distance of
// its leaf's box in the reference's hierarchy), the smallest distance of any other hit seen, and whether the ray has
// to be walked again in the reference's order.
struct WalkHits {
    Hit best;
    uint32_t best_ci, best_rank;      // best_rank 0xffffffff: no hit yet
    float reach, runner_up;
    // Boxes are entered up to a little beyond reach: a triangle's hit distance and the entry distance of a box around
    // it are two roundings of (nearly) the same number - the same one for a triangle in the face of a flat box - and
    // a hit that is closer than the closest one by a rounding must not be lost behind a box that is farther by one.
    float reach_in;
    bool again;
};
__device__ __forceinline__ void walk_hits_start(WalkHits& s, float tmax_r, float beyond) {
    s.best.t = tmax_r; s.best.u = 0.0f; s.best.v = 0.0f; s.best.w = 0.0f;
    s.best_ci = 0u; s.best_rank = 0xffffffffu;
    s.reach = tmax_r; s.runner_up = __builtin_inff();
    s.reach_in = tmax_r * beyond;
    s.again = false;
}
// a triangle that the ray hits at hk (index slot ck, place `rank` in the reference's order), found in whatever order
__device__ __forceinline__ void walk_hits_consider(WalkHits& s, const Hit& hk, uint32_t ck, uint32_t rank, f3 org, f3 inv_d, float lim,
                                                   float tmax_r, float beyond, const char* __restrict__ ref_leaf_bytes) {
    if (!(hk.t <= s.reach)) return;
    // The reference tests this triangle only if the box of ITS leaf of the triangle lets the ray in: that leaf is the
    // pair of neighbours in the reference's order.
    const float4* rt = reinterpret_cast<const float4*>(ref_leaf_bytes + (size_t)(rank >> 1) * 128u);
    const float entry = leaf_box_entry(org, inv_d, lim, rt[0], rt[1], rt[2], rt[3], rt[4]);
    if (!(entry < __builtin_inff())) return;
    const bool have_hit = s.best_rank != 0xffffffffu;
    const float far_end = fmaxf(entry, hk.t);
    const bool closer = !have_hit || hk.t < s.best.t || (hk.t == s.best.t && rank < s.best_rank);
    if (!closer) {
        s.runner_up = fminf(s.runner_up, hk.t);
    } else if (have_hit ? far_end <= s.reach : far_end < tmax_r) {
        if (have_hit) s.runner_up = fminf(s.runner_up, s.best.t);
        s.best = hk; s.best_ci = ck; s.best_rank = rank; s.reach = far_end;
        s.reach_in = s.reach * beyond;
    } else if (have_hit) {
        s.again = true;          // a hit that is closer, behind a box entry that is not
    }
}
// the two triangles of a leaf record (l0 .. l4; places: theirs in the reference's order)
__device__ __forceinline__ void walk_hits_leaf(WalkHits& s, f3 org, f3 dir, f3 inv_d, float tmin_r, float tmax_r, float lim, float beyond,
                                               float4 l0, float4 l1, float4 l2, float4 l3, float4 l4, float2 places,
                                               const char* __restrict__ ref_leaf_bytes) {
    const uint32_t ci0 = __float_as_uint(l4.z), ci1 = __float_as_uint(l4.w);
    Hit h0, h1;
    intersect_leaf(org, dir, l0, l1, l2, l3, l4, h0, h1);
    const bool may0 = (tmin_r <= h0.t) && (tmax_r > h0.t) && (h0.t <= s.reach);
    const bool may1 = (ci1 != 0xffffffffu) && (tmin_r <= h1.t) && (tmax_r > h1.t) && (h1.t <= s.reach);
    // few leaves have a triangle that is hit, fewer two: one turn of this loop per such triangle
#pragma unroll 1
    for (uint32_t todo = (may0 ? 1u : 0u) | (may1 ? 2u : 0u); todo != 0u; todo &= todo - 1u) {
        const bool k = (todo & 1u) == 0u;         // the second triangle's turn
        Hit hk;
        hk.t = k ? h1.t : h0.t; hk.u = k ? h1.u : h0.u; hk.v = k ? h1.v : h0.v; hk.w = k ? h1.w : h0.w;
        walk_hits_consider(s, hk, k ? ci1 : ci0, __float_as_uint(k ? places.y : places.x), org, inv_d, lim, tmax_r, beyond, ref_leaf_bytes);
    }
}
// a finished ray leaves its closest hit behind; a ray for which the reference's order decides goes on the list
__device__ __forceinline__ void walk_hits_finish(WalkHits& s, uint32_t ray, float4* __restrict__ hit_records, uint32_t* __restrict__ hit_ci,
                                                 uint2* __restrict__ flagged, uint32_t* __restrict__ flag_count) {
    hit_records[ray] = make_float4(s.best.t, s.best.u, s.best.v, s.best.w);
    hit_ci[ray] = s.best_ci;
    // the closest hit lies before the entry of its leaf's box and another hit lies in between: which of
    // them the reference keeps depends on the order it meets them in
    const bool during_walk = s.again;       // (a closer hit behind a box entry beyond reach; counted apart in flag_count[2])
    if (s.best_rank != 0xffffffffu && s.reach > s.best.t && s.runner_up <= s.reach) s.again = true;
    if (s.again) {
        flagged[atomicAdd(flag_count, 1u)] = make_uint2(0u, ray);
        if (during_walk) atomicAdd(flag_count + 2, 1u);
    }
}

template <bool COUNT>
__global__ __launch_bounds__(256) void k_traverse_spatial(TraceParams p, const float4* __restrict__ rays, const float4* __restrict__ nodes,
                                                       const float4* __restrict__ pairs, const float4* __restrict__ quads,
                                                       const float4* __restrict__ sorted_tris, float4* __restrict__ hit_records,
                                                       uint32_t* __restrict__ hit_ci, uint32_t* __restrict__ queue,
                                                       const uint2* __restrict__ order, const float4* __restrict__ ref_tris,
                                                       uint2* __restrict__ flagged,
                                                       uint32_t* __restrict__ flag_count, unsigned long long* __restrict__ walk_stats,
                                                       const uint32_t* __restrict__ live, const float4* __restrict__ outlier_leaves,
                                                       const uint32_t* __restrict__ outlier_count, const uint8_t* __restrict__ skip) {
    extern __shared__ uint32_t s_dyn[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    uint32_t num_rays = p.num_rays, queue_chunk = p.queue_chunk;      // (the kernel argument itself stays untouched: in registers)
    if (live) {
        // drop-missed-rays mode: the rays to walk are the first *live entries of order (k_ray_keys put the others last)
        num_rays = *live;
        if (num_rays == 0u) return;
        const uint32_t deal_block = 1u << p.deal_shift;
        queue_chunk = ((((num_rays + deal_block - 1u) >> p.deal_shift) + 7u) / 8u) * deal_block;
    }
    const uint32_t root_level = p.lv.count - 1u;               // >= 2
    const uint32_t first_leaf = 1u << root_level;              // slot of leaf 0
    const uint32_t first_leaf_grandparent = first_leaf >> 2;   // nodes from this slot on have leaves as grandchildren
    const bool odd_first = (root_level & 1u) != 0u;
    uint2* s_fifo = reinterpret_cast<uint2*>(s_dyn) + tid;     // [kFifo][256] waiting leaves, this lane's column
    float* s_stock = reinterpret_cast<float*>(s_dyn + kFifo * 512u) + (tid >> 6) * (kStockFields * 64u);   // [14][64] this wave's rays to come
    uint32_t stock_next = 0u, stock_count = 0u;                // wave-uniform: entries taken / held
    const float4 root_lo = nodes[0], root_hi = nodes[1];
    const char* quad_bytes = reinterpret_cast<const char*>(quads);
    const char* leaf_bytes = reinterpret_cast<const char*>(sorted_tris);
    const char* ref_leaf_bytes = reinterpret_cast<const char*>(ref_tris);

    uint32_t ray = 0u;                // index of the ray this lane is walking (while w < kIdle)
    uint32_t w = kIdle;
    f3 org = mk(0.0f, 0.0f, 0.0f), dir = org, inv_d = org;
    f3 inv_w = org;                   // reciprocals used with the widened boxes: finite where the direction is zero
    bool neg_x = false, neg_y = false, neg_z = false;      // the ray travels downwards on that axis
    float tmin_r = 0.0f, tmax_r = 0.0f, lim = 0.0f;
    WalkHits hits;
    walk_hits_start(hits, 0.0f, 1.0f);
    const float kBeyond = p.beyond;             // 1 + 2^-8 unless a test has asked for less (rt_device_set_walk_margins)
    // leaf records of the build's outliers (triangles the hierarchy leaves out), tested before a ray's walk starts
    const uint32_t outlier_records = (min(*outlier_count, RT_OUTLIERS_MAX) + 1u) / 2u;
    uint32_t h = 1u;                  // slot of the entered node
    unsigned long long trail = 0ull;
    uint32_t wait_head = 0u;
    uint32_t q = blockIdx.x & 7u, queues_tried = 0u;       // wave-uniform
    uint32_t n_steps = 0u, n_empty = 0u, n_leaves = 0u;    // COUNT: this wave's steps, steps that hit nothing, leaves tested
    uint32_t n_turns = 0u, n_rounds = 0u;                  // COUNT: turns of the walking loop and triangle rounds of this wave
    // The end of a launch. When the queues run dry every wave still holds a hundred rays, and works them off with fewer and
    // fewer lanes - while the three other waves of its SIMD do the same, each turn of each of them taking its full share of
    // the SIMD's issue slots: a quarter of the rays of a two-million-ray launch, and half a millisecond on the benchmark's
    // scene, whatever the size of the launch. So the workgroups do not all stay to the end: those of class c (a quarter of
    // the grid each, in launch order - each CU gets one workgroup of every class, i.e. every SIMD one wave) take no more
    // rays once the queues will be empty within c * retire_ticks at the pace so far (what is left in the wave's queue, over
    // what has been taken from it, times the time since the wave started), run out while the others still work at full
    // width, and leave; the last rays are walked by one wave per SIMD, whose turns then take a fraction of the time.
    // Class 0 never stops: every ray is taken.
    const uint32_t retire_class = p.retire_ticks ? blockIdx.x / max(gridDim.x >> 2, 1u) : 0u;
    const unsigned long long retire_t0 = retire_class ? wall_clock64() : 0ull;
    // COUNT: this wave's steps by depth of the entered node (two levels per entry: slot 1 is depth 0)
    uint32_t* s_hist = s_dyn + kFifo * 512u + 4u * kStockFields * 64u + (tid >> 6) * 16u;
    if (COUNT) {
        if (lane < 16u) s_hist[lane] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }

    for (;;) {
        // 1. Lanes without a ray take one from the wave's stock in LDS. The stock is filled 64 rays at a time by the
        //    whole wave - one atomic on the queue, the ray loads, the reciprocals and the test of the trunk's box done
        //    by all lanes at once - so that taking a ray is a dozen LDS reads, worth doing for a few free lanes
        //    already (a fill through the free lanes only, as in k_traverse_quad, has to wait for many of them).
        const unsigned long long need_mask = __builtin_amdgcn_ballot_w64(w == kIdle);
        if (need_mask && ((uint32_t)__popcll(need_mask) >= p.refill_threshold || !__builtin_amdgcn_ballot_w64(w <= kFifo - 1u))) {
            while (stock_next == stock_count && queues_tried < 8u) {             // wave-uniform
                const uint32_t q_begin = q * queue_chunk;
                const uint32_t q_end = q_begin + queue_chunk;
                uint32_t base = 0u;
                if (lane == 0u) base = atomicAdd(queue + 8u * q, 64u);
                base = q_begin + (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (base >= q_end) {                               // this queue is empty
                    q = (q + 1u) & 7u;
                    queues_tried++;
                    continue;
                }
                // (the eight queues empty at the same pace: what is left in this one is an eighth of what is left)
                if (retire_class != 0u) {
                    // (the eight queues empty at the same pace)
                    const uint32_t taken = base + 64u - q_begin, left = q_end - min(q_end, base + 64u);
                    const float to_go = (float)(wall_clock64() - retire_t0) * ((float)left / (float)taken);
                    if (to_go < (float)(retire_class * p.retire_ticks)) queues_tried = 8u;      // these 64 are its last
                }
                const uint32_t mine = base + lane;
                bool keep = false;
                f3 c_org = mk(0.0f, 0.0f, 0.0f), c_dir = c_org, c_inv = c_org;
                float c_tmin = 0.0f, c_tmax = 0.0f;
                uint32_t c_rid = 0u, c_h = 1u, c_trail = 0u;
                WalkHits c_hits;
                walk_hits_start(c_hits, 0.0f, 1.0f);
                bool valid = false;
                if (mine < q_end) {
                    const uint32_t slot = mine - q_begin;
                    const uint32_t pos = (((slot >> p.deal_shift) * 8u + q) << p.deal_shift) | (slot & ((1u << p.deal_shift) - 1u));
                    // (skip: rays as they come in drop-missed-rays mode - a ray that has missed is passed over here; where the
                    // rays are ordered, such rays lie behind the first *live entries and are never handed out)
                    if (pos < num_rays && !(skip && skip[order ? order[pos].y : pos])) {
                        valid = true;
                        c_rid = order ? order[pos].y : pos;
                        const float4 r0 = rays[2 * (size_t)c_rid], r1 = rays[2 * (size_t)c_rid + 1];
                        c_org = mk(r0.x, r0.y, r0.z); c_dir = mk(r0.w, r1.x, r1.y);
                        c_tmin = r1.z; c_tmax = r1.w;
                        c_inv = mk(1.0f / c_dir.x, 1.0f / c_dir.y, 1.0f / c_dir.z);
                        walk_hits_start(c_hits, c_tmax, kBeyond);
                    }
                }
                if (outlier_records) {
                    // the outliers first, by every lane that has a ray (the records' addresses are the same for all lanes):
                    // a scene's walls and floor bound the reach of most rays before the walk begins.
                    // A ray's hit is confirmed against its leaf in the reference's order (walk_hits_consider) once, behind the
                    // last record, by all lanes together - most rays hit one outlier at most (the wall they end on); a lane that
                    // finds a second hit confirms the other one there and then. The order of confirming does not enter the result.
                    const float c_lim = (c_tmin > 0.0f) ? c_tmin : __uint_as_float(1u);
                    Hit held;
                    held.t = held.u = held.v = held.w = 0.0f;
                    uint32_t held_ci = 0u, held_rank = 0xffffffffu;
                    for (uint32_t j = 0u; j <= outlier_records; j++) {
                        const bool last = j == outlier_records;
                        uint32_t todo = held_rank != 0xffffffffu ? 4u : 0u;
                        Hit h0, h1;
                        h0.t = h0.u = h0.v = h0.w = 0.0f; h1 = h0;
                        uint32_t ci0 = 0u, ci1 = 0u, place0 = 0u, place1 = 0u;
                        if (!last) {
                            const float4* t = outlier_leaves + RT_LEAF_STRIDE * j;
                            const float4 l0 = t[0], l1 = t[1], l2 = t[2], l3 = t[3], l4 = t[4];
                            const float2 places = *reinterpret_cast<const float2*>(t + 5);
                            if (valid) {
                                ci0 = __float_as_uint(l4.z); ci1 = __float_as_uint(l4.w);
                                place0 = __float_as_uint(places.x); place1 = __float_as_uint(places.y);
                                intersect_leaf(c_org, c_dir, l0, l1, l2, l3, l4, h0, h1);
                                if ((c_tmin <= h0.t) && (c_tmax > h0.t) && (h0.t <= c_hits.reach)) todo |= 1u;
                                if ((ci1 != 0xffffffffu) && (c_tmin <= h1.t) && (c_tmax > h1.t) && (h1.t <= c_hits.reach)) todo |= 2u;
                            }
                        }
#pragma unroll 1
                        while ((todo & (todo - 1u)) != 0u || (last && todo != 0u)) {
                            const bool k1 = (todo & 3u) == 2u, kh = (todo & 3u) == 0u;      // the record's second triangle, the held hit
                            Hit hk;
                            hk.t = kh ? held.t : (k1 ? h1.t : h0.t); hk.u = kh ? held.u : (k1 ? h1.u : h0.u);
                            hk.v = kh ? held.v : (k1 ? h1.v : h0.v); hk.w = kh ? held.w : (k1 ? h1.w : h0.w);
                            walk_hits_consider(c_hits, hk, kh ? held_ci : (k1 ? ci1 : ci0), kh ? held_rank : (k1 ? place1 : place0), c_org, c_inv,
                                               c_lim, c_tmax, kBeyond, ref_leaf_bytes);
                            todo &= todo - 1u;
                        }
                        if (todo & 3u) {                      // this record's one hit that is left waits
                            const bool k1 = (todo & 1u) == 0u;
                            held.t = k1 ? h1.t : h0.t; held.u = k1 ? h1.u : h0.u; held.v = k1 ? h1.v : h0.v; held.w = k1 ? h1.w : h0.w;
                            held_ci = k1 ? ci1 : ci0;
                            held_rank = k1 ? place1 : place0;
                        }
                    }
                }
                if (valid) {
                    // the trunk is entered on any box hit within reach
                    keep = slab(c_org, c_inv, c_tmin, c_hits.reach_in, root_lo, root_hi) != 1e30f;
                    if (keep && odd_first) {
                        // the trunk's two children, exact boxes, from the nodes in the reference's layout
                        const float4* c = nodes + 2 * (size_t)p.trunk_children;
                        const float bt0 = slab(c_org, c_inv, c_tmin, c_tmax, c[0], c[1]), bt1 = slab(c_org, c_inv, c_tmin, c_tmax, c[2], c[3]);
                        const bool hit0 = (bt0 != 1e30f) && (bt0 <= c_hits.reach_in);
                        const bool hit1 = (bt1 != 1e30f) && (bt1 <= c_hits.reach_in);
                        keep = hit0 || hit1;
                        c_h = hit0 ? 2u : 3u;
                        c_trail = (hit0 && hit1) ? 8u : 0u;
                    }
                    // nothing to walk: the ray's record is what the outliers have left, or that of a miss
                    if (!keep) walk_hits_finish(c_hits, c_rid, hit_records, hit_ci, flagged, flag_count);
                }
                const unsigned long long kept = __builtin_amdgcn_ballot_w64(keep);
                if (keep) {
                    float* e = s_stock + __builtin_amdgcn_mbcnt_hi((uint32_t)(kept >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)kept, 0u));
                    e[0 * 64] = c_org.x; e[1 * 64] = c_org.y; e[2 * 64] = c_org.z;
                    e[3 * 64] = c_dir.x; e[4 * 64] = c_dir.y; e[5 * 64] = c_dir.z;
                    e[6 * 64] = c_inv.x; e[7 * 64] = c_inv.y; e[8 * 64] = c_inv.z;
                    e[9 * 64] = c_tmin; e[10 * 64] = c_tmax;
                    e[11 * 64] = __uint_as_float(c_rid); e[12 * 64] = __uint_as_float(c_h);
                    e[13 * 64] = __uint_as_float(c_trail | (c_hits.again ? 0x80000000u : 0u));
                    e[14 * 64] = c_hits.best.t; e[15 * 64] = c_hits.best.u; e[16 * 64] = c_hits.best.v; e[17 * 64] = c_hits.best.w;
                    e[18 * 64] = __uint_as_float(c_hits.best_ci); e[19 * 64] = __uint_as_float(c_hits.best_rank);
                    e[20 * 64] = c_hits.reach; e[21 * 64] = c_hits.runner_up;
                }
                stock_next = 0u;
                stock_count = (uint32_t)__popcll(kept);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
            if (w == kIdle) {
                const uint32_t at = stock_next + __builtin_amdgcn_mbcnt_hi((uint32_t)(need_mask >> 32),
                                                                           __builtin_amdgcn_mbcnt_lo((uint32_t)need_mask, 0u));
                if (at < stock_count) {
                    const float* e = s_stock + at;
                    org = mk(e[0 * 64], e[1 * 64], e[2 * 64]); dir = mk(e[3 * 64], e[4 * 64], e[5 * 64]);
                    inv_d = mk(e[6 * 64], e[7 * 64], e[8 * 64]);
                    tmin_r = e[9 * 64]; tmax_r = e[10 * 64];
                    ray = __float_as_uint(e[11 * 64]); h = __float_as_uint(e[12 * 64]);
                    const uint32_t noted = __float_as_uint(e[13 * 64]);
                    trail = (unsigned long long)(noted & 0x7fffffffu);
                    lim = (tmin_r > 0.0f) ? tmin_r : __uint_as_float(1u);
                    inv_w = mk(dir.x == 0.0f ? copysignf(3.0e38f, inv_d.x) : inv_d.x, dir.y == 0.0f ? copysignf(3.0e38f, inv_d.y) : inv_d.y,
                               dir.z == 0.0f ? copysignf(3.0e38f, inv_d.z) : inv_d.z);
                    neg_x = inv_w.x < 0.0f; neg_y = inv_w.y < 0.0f; neg_z = inv_w.z < 0.0f;
                    hits.best.t = e[14 * 64]; hits.best.u = e[15 * 64]; hits.best.v = e[16 * 64]; hits.best.w = e[17 * 64];
                    hits.best_ci = __float_as_uint(e[18 * 64]); hits.best_rank = __float_as_uint(e[19 * 64]);
                    hits.reach = e[20 * 64]; hits.runner_up = e[21 * 64];
                    hits.reach_in = hits.reach * kBeyond;
                    hits.again = (noted & 0x80000000u) != 0u;
                    w = 0u;
                } else if (queues_tried >= 8u) {
                    w = kDry;                                      // no stock left and none to come
                }
            }
            stock_next = min(stock_count, stock_next + (uint32_t)__popcll(need_mask));
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        if (!__builtin_amdgcn_ballot_w64(w < kIdle)) {
            if (!__builtin_amdgcn_ballot_w64(w != kDry)) {            // queues empty and nothing in flight: the wave is done
                if (COUNT && lane == 0u) {
                    atomicAdd(walk_stats, (unsigned long long)n_steps);
                    atomicAdd(walk_stats + 1, (unsigned long long)n_empty);
                    atomicAdd(walk_stats + 2, (unsigned long long)n_leaves);
                    atomicAdd(walk_stats + 3, (unsigned long long)n_turns | ((unsigned long long)n_rounds << 32));
                }
                if (COUNT && p.walk_hist && lane < 16u) atomicAdd(walk_stats + 4u + lane, (unsigned long long)s_hist[lane]);
                break;
            }
            continue;
        }

        // 2. walk until a triangle round is due, a lane has finished its ray, or nobody can walk
        unsigned long long waiting_mask, walking_mask;
        uint32_t stuck = 0u;
        // (leaf_threshold is meant of 64 lanes with a ray each: a wave with fewer rays - the end of a launch, a small scene,
        // lanes waiting for the next refill - starts its rounds at the same share of them, so that a hit still shortens its
        // ray's walk in time)
        const uint32_t with_ray = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(w < kIdle));
        const uint32_t round_at = max(1u, (with_ray * p.leaf_threshold + 63u) >> 6);
        for (;;) {
            bool empty_step = false;                       // COUNT: this lane's step found none of the four boxes hit
            if (COUNT) { n_turns += 1u; n_steps += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(w <= kFifo - 1u)); }
            if (w <= kFifo - 1u) {
                if (COUNT) atomicAdd(&s_hist[min(15u, (31u - (uint32_t)__builtin_clz(h)) >> 1)], 1u);
                const uint4* rec = reinterpret_cast<const uint4*>(quad_bytes + (h << 6));
                const uint4 u0 = rec[0], u1 = rec[1], u2 = rec[2];
                const v2f ix = {inv_w.x, inv_w.x}, iy = {inv_w.y, inv_w.y}, iz = {inv_w.z, inv_w.z};
                // first pair (slots 4h, 4h + 1): dwords 0 - 5 = min x y z, max x y z; second pair: dwords 6 - 11.
                // The plane a ray meets first on an axis is the minimum if it travels upwards, else the maximum.
                const v2f anx = half_pair_minus(neg_x ? u0.w : u0.x, org.x) * ix, afx = half_pair_minus(neg_x ? u0.x : u0.w, org.x) * ix;
                const v2f any = half_pair_minus(neg_y ? u1.x : u0.y, org.y) * iy, afy = half_pair_minus(neg_y ? u0.y : u1.x, org.y) * iy;
                const v2f anz = half_pair_minus(neg_z ? u1.y : u0.z, org.z) * iz, afz = half_pair_minus(neg_z ? u0.z : u1.y, org.z) * iz;
                const v2f bnx = half_pair_minus(neg_x ? u2.y : u1.z, org.x) * ix, bfx = half_pair_minus(neg_x ? u1.z : u2.y, org.x) * ix;
                const v2f bny = half_pair_minus(neg_y ? u2.z : u1.w, org.y) * iy, bfy = half_pair_minus(neg_y ? u1.w : u2.z, org.y) * iy;
                const v2f bnz = half_pair_minus(neg_z ? u2.w : u2.x, org.z) * iz, bfz = half_pair_minus(neg_z ? u2.x : u2.w, org.z) * iz;
                const float tn0 = vmax3(anx.x, any.x, anz.x), tx0 = vmin3(afx.x, afy.x, afz.x);
                const float tn1 = vmax3(anx.y, any.y, anz.y), tx1 = vmin3(afx.y, afy.y, afz.y);
                const float tn2 = vmax3(bnx.x, bny.x, bnz.x), tx2 = vmin3(bfx.x, bfy.x, bfz.x);
                const float tn3 = vmax3(bnx.y, bny.y, bnz.y), tx3 = vmin3(bfx.y, bfy.y, bfz.y);
                const bool hit0 = !(tx0 < vmax(tn0, lim)) && (tn0 <= hits.reach_in);
                const bool hit1 = !(tx1 < vmax(tn1, lim)) && (tn1 <= hits.reach_in);
                const bool hit2 = !(tx2 < vmax(tn2, lim)) && (tn2 <= hits.reach_in);
                const bool hit3 = !(tx3 < vmax(tn3, lim)) && (tn3 <= hits.reach_in);
                const uint32_t m = (hit0 ? 1u : 0u) | (hit1 ? 2u : 0u) | (hit2 ? 4u : 0u) | (hit3 ? 8u : 0u);
                if (COUNT) empty_step = m == 0u;
                if (h < first_leaf_grandparent && m != 0u) {
                    // into the nearest of the grandchildren that are hit; the others wait in slot order
                    const float d0 = hit0 ? tn0 : __builtin_inff(), d1 = hit1 ? tn1 : __builtin_inff();
                    const float d2 = hit2 ? tn2 : __builtin_inff(), d3 = hit3 ? tn3 : __builtin_inff();
                    const uint32_t j01 = d1 < d0 ? 1u : 0u, j23 = d3 < d2 ? 3u : 2u;
                    const uint32_t j = vmin(d2, d3) < vmin(d0, d1) ? j23 : j01;
                    trail = (trail << 4) | (unsigned long long)(m & ~(1u << j));
                    h = (h << 2) | j;
                } else {
                    if (h >= first_leaf_grandparent && m != 0u) {
                        // one entry for the up to four leaves that are hit: first leaf, which of the four (bits 28 - 31:
                        // leaf numbers stay below 2^25), and the smallest of their entry distances
                        const float d0 = hit0 ? tn0 : __builtin_inff(), d1 = hit1 ? tn1 : __builtin_inff();
                        const float d2 = hit2 ? tn2 : __builtin_inff(), d3 = hit3 ? tn3 : __builtin_inff();
                        s_fifo[((wait_head + w) & (kFifo - 1u)) * 256u] =
                            make_uint2((4u * h - first_leaf) | (m << 28), __float_as_uint(vmin(vmin(d0, d1), vmin(d2, d3))));
                        w++;
                    }
                    // back to the first noted node of the nearest level that has one, or the walk is over
                    if (trail == 0ull) {
                        w |= kOver;
                    } else {
                        const uint32_t z = (uint32_t)__builtin_ctzll(trail);
                        const uint32_t up = z & ~3u;
                        unsigned long long t2 = trail >> up;
                        t2 &= t2 - 1ull;
                        h = ((h >> (up >> 1)) & ~3u) | (z & 3u);
                        trail = t2;
                    }
                }
            }
            if (COUNT) n_empty += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(empty_step));
            waiting_mask = __builtin_amdgcn_ballot_w64((w & 15u) != 0u);
            walking_mask = __builtin_amdgcn_ballot_w64(w <= kFifo - 1u);
            // (lanes whose walk is over wait for a triangle round with nothing else to do: a few of them start one
            // before the waiting leaves alone would)
            stuck = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(w > kOver && w < kIdle));
            if (!walking_mask || (uint32_t)__popcll(waiting_mask) >= round_at || stuck >= p.stuck_threshold) break;
            if ((uint32_t)__popcll(__builtin_amdgcn_ballot_w64(w >= kOver && w != kDry)) >= p.refill_threshold) break;
        }

        // 3. triangle round once enough lanes have a leaf waiting (or nobody can walk any more)
        if (waiting_mask && ((uint32_t)__popcll(waiting_mask) >= round_at || !walking_mask || stuck >= p.stuck_threshold)) {
            if (COUNT) n_rounds += 1u;
            bool tested = false;                           // COUNT: this lane tests a leaf in this round
            if ((w & 15u) != 0u) {
                // the oldest entry's next leaf; the entry goes when its last leaf is taken or its leaves lie beyond reach
                const uint2 e = s_fifo[wait_head * 256u];
                const bool in_reach = __uint_as_float(e.y) <= hits.reach_in;
                const uint32_t left = e.x >> 28, leaf = (e.x & 0x0fffffffu) + (uint32_t)__builtin_ctz(left);
                const uint32_t rest = left & (left - 1u);
                if (rest != 0u && in_reach) {
                    s_fifo[wait_head * 256u].x = (e.x & 0x0fffffffu) | (rest << 28);
                } else {
                    wait_head = (wait_head + 1u) & (kFifo - 1u);
                    w--;
                }
                if (in_reach) {
                    if (COUNT) tested = true;
                    const float4* t = reinterpret_cast<const float4*>(leaf_bytes + (size_t)leaf * 128u);      // few per ray: 64-bit offsets cost nothing here
                    const float4 l0 = t[0], l1 = t[1], l2 = t[2], l3 = t[3], l4 = t[4];
                    const float2 places = *reinterpret_cast<const float2*>(t + 5);      // of the two triangles in the reference's order
                    walk_hits_leaf(hits, org, dir, inv_d, tmin_r, tmax_r, lim, kBeyond, l0, l1, l2, l3, l4, places, ref_leaf_bytes);
                }
            }
            if (COUNT) n_leaves += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(tested));
        }

        // 4. a finished ray leaves its closest hit behind and frees the lane
        if (w == kOver) {
            walk_hits_finish(hits, ray, hit_records, hit_ci, flagged, flag_count);
            w = kIdle;
        }
    }
}

// Key of every ray for the order the traversal queues hand rays out in (the reference has no such step; the
// order rays are traced in does not enter any result).
//  * Major part, descending: an estimate of the walk's length - how many of a sample of the nodes at one depth
//    the ray's segment touches. Walk lengths are spread over three orders of magnitude, and a launch ends
//    with its longest walks; handing those out first keeps all lanes busy to the end.
//  * Minor part: locality - rays that start in the same cell of a 16^3 grid over the trunk's box and point
//    into the same of 4^3 direction cells get neighbouring queue positions.
// first_record / num_records: the records (pairs of boxes) of the nodes at the probing depth, which every
// lane reads in the same order (uniform addresses).
__device__ __forceinline__ uint32_t spread3(uint32_t v) {      // abcd -> a00b00c00d
    v = (v | (v << 8)) & 0x0000f00fu;
    v = (v | (v << 4)) & 0x000c30c3u;
    v = (v | (v << 2)) & 0x00249249u;
    return v;
}
// the key of a ray without a probe: (descending) length of its way through the trunk's box lo / hi in units of the box's
// diagonal, then the cell of its origin in a 16^3 grid over that box and the cell of its direction; 24 bits, of which
// the leading key_bits are kept
__device__ __forceinline__ uint32_t chord_key(const float4 r0, const float4 r1, const float4 lo, const float4 hi, uint32_t cost_bits,
                                              uint32_t key_bits) {
    const f3 org = mk(r0.x, r0.y, r0.z), dir = mk(r0.w, r1.x, r1.y);
    const f3 inv_d = mk(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
    const uint32_t top_class = (1u << cost_bits) - 1u;
    const f3 t1 = mul(sub(mk(lo.x, lo.y, lo.z), org), inv_d), t2 = mul(sub(mk(lo.w, hi.x, hi.y), org), inv_d);
    const float tn = fmaxf(fmaxf(fminf(t1.x, t2.x), fminf(t1.y, t2.y)), fmaxf(fminf(t1.z, t2.z), r1.z));
    const float tf = fminf(fminf(fmaxf(t1.x, t2.x), fmaxf(t1.y, t2.y)), fminf(fmaxf(t1.z, t2.z), r1.w));
    const f3 ext = sub(mk(lo.w, hi.x, hi.y), mk(lo.x, lo.y, lo.z));
    const float diag = sqrtf(dot3(ext, ext)) * sqrtf(dot3(dir, dir));
    const float part = fmaxf(tf - tn, 0.0f) / diag;           // NaN -> 0 below
    const uint32_t cost_class = (uint32_t)fminf(fmaxf(part * (float)(top_class + 1u), 0.0f), (float)top_class);
    const float cx = (r0.x - lo.x) / (lo.w - lo.x), cy = (r0.y - lo.y) / (hi.x - lo.y), cz = (r0.z - lo.z) / (hi.y - lo.z);
    const uint32_t ox = (uint32_t)fminf(fmaxf(cx * 16.0f, 0.0f), 15.0f);
    const uint32_t oy = (uint32_t)fminf(fmaxf(cy * 16.0f, 0.0f), 15.0f);
    const uint32_t oz = (uint32_t)fminf(fmaxf(cz * 16.0f, 0.0f), 15.0f);
    const uint32_t dx = (uint32_t)fminf(fmaxf(r0.w * 2.0f + 2.0f, 0.0f), 3.0f);
    const uint32_t dy = (uint32_t)fminf(fmaxf(r1.x * 2.0f + 2.0f, 0.0f), 3.0f);
    const uint32_t dz = (uint32_t)fminf(fmaxf(r1.y * 2.0f + 2.0f, 0.0f), 3.0f);
    const uint32_t okey = spread3(ox) | (spread3(oy) << 1) | (spread3(oz) << 2);      // 12 bits
    const uint32_t dkey = spread3(dx) | (spread3(dy) << 1) | (spread3(dz) << 2);      // 6 bits
    return (((top_class - cost_class) << (24u - cost_bits)) | (((okey << 6) | dkey) >> (cost_bits - 6u))) >> (24u - key_bits);
}

__global__ __launch_bounds__(256) void k_ray_keys(uint32_t num_rays, const float4* __restrict__ rays,
                                                  const float4* __restrict__ nodes, const float4* __restrict__ pairs,
                                                  uint32_t first_record, uint32_t num_records, uint32_t record_stride,
                                                  uint32_t cost_shift, uint32_t cost_bits, uint32_t key_bits, uint2* __restrict__ keys,
                                                  const uint8_t* __restrict__ missed, uint32_t* __restrict__ live) {
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    if (missed) {
        // drop-missed-rays mode: rays that have missed get the largest key (they sort behind every live ray, whose keys stay
        // below it) and are not counted; the traversal walks the first *live entries of the sorted order only
        const bool alive = tid < num_rays && !missed[tid];
        const unsigned long long m = __builtin_amdgcn_ballot_w64(alive);
        if ((threadIdx.x & 63u) == 0u && m) atomicAdd(live, (uint32_t)__popcll(m));
        if (tid < num_rays && !alive) { keys[tid] = make_uint2(0xffffffffu >> (32u - key_bits), tid); return; }
    }
    if (tid >= num_rays) return;
    const float4 lo = nodes[0], hi = nodes[1];
    const float4 r0 = rays[2 * (size_t)tid], r1 = rays[2 * (size_t)tid + 1];
    const f3 org = mk(r0.x, r0.y, r0.z), dir = mk(r0.w, r1.x, r1.y);
    const f3 inv_d = mk(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
    uint32_t touched = 0u;
    for (uint32_t j = 0u; j < num_records; j += record_stride) {
        const float4* rec = pairs + 4 * (size_t)(first_record + j);
        const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
        float bt0, bt1;
        slab_pair(org, inv_d, r1.z, r1.w, q0, q1, q2, bt0, bt1);
        touched += (bt0 != 1e30f ? 1u : 0u) + ((q0.y == q0.y) && (bt1 != 1e30f) ? 1u : 0u);
    }
    const uint32_t top_class = (1u << cost_bits) - 1u;
    uint32_t cost_class = min(touched >> cost_shift, top_class);
    if (num_records == 0u) {
        // no probe: the length of the ray's segment inside the trunk's box, in units of the box's diagonal
        const f3 t1 = mul(sub(mk(lo.x, lo.y, lo.z), org), inv_d), t2 = mul(sub(mk(lo.w, hi.x, hi.y), org), inv_d);
        const float tn = fmaxf(fmaxf(fminf(t1.x, t2.x), fminf(t1.y, t2.y)), fmaxf(fminf(t1.z, t2.z), r1.z));
        const float tf = fminf(fminf(fmaxf(t1.x, t2.x), fmaxf(t1.y, t2.y)), fminf(fmaxf(t1.z, t2.z), r1.w));
        const f3 ext = sub(mk(lo.w, hi.x, hi.y), mk(lo.x, lo.y, lo.z));
        const float diag = sqrtf(dot3(ext, ext)) * sqrtf(dot3(dir, dir));
        const float part = fmaxf(tf - tn, 0.0f) / diag;           // NaN -> 0 below
        cost_class = (uint32_t)fminf(fmaxf(part * (float)(top_class + 1u), 0.0f), (float)top_class);
    }
    const float cx = (r0.x - lo.x) / (lo.w - lo.x), cy = (r0.y - lo.y) / (hi.x - lo.y), cz = (r0.z - lo.z) / (hi.y - lo.z);
    const uint32_t ox = (uint32_t)fminf(fmaxf(cx * 16.0f, 0.0f), 15.0f);
    const uint32_t oy = (uint32_t)fminf(fmaxf(cy * 16.0f, 0.0f), 15.0f);
    const uint32_t oz = (uint32_t)fminf(fmaxf(cz * 16.0f, 0.0f), 15.0f);
    const uint32_t dx = (uint32_t)fminf(fmaxf(r0.w * 2.0f + 2.0f, 0.0f), 3.0f);
    const uint32_t dy = (uint32_t)fminf(fmaxf(r1.x * 2.0f + 2.0f, 0.0f), 3.0f);
    const uint32_t dz = (uint32_t)fminf(fmaxf(r1.y * 2.0f + 2.0f, 0.0f), 3.0f);
    const uint32_t okey = spread3(ox) | (spread3(oy) << 1) | (spread3(oz) << 2);      // 12 bits
    const uint32_t dkey = spread3(dx) | (spread3(dy) << 1) | (spread3(dz) << 2);      // 6 bits
    // 24 bits: the (descending) cost class leads, the locality bits that still fit follow
    // (key_bits = 8: the class alone, one sorting pass - rays of a class keep the order they come in)
    uint32_t key = (((top_class - cost_class) << (24u - cost_bits)) | (((okey << 6) | dkey) >> (cost_bits - 6u))) >> (24u - key_bits);
    if (missed) key = min(key, (0xffffffffu >> (32u - key_bits)) - 1u);
    keys[tid] = make_uint2(key, tid);
}

// next_keys (optional; never together with missed): the key k_ray_keys would give the ray this call leaves behind, for the
// next trace call on the same hierarchy - the ray is in registers here, so that call need not read all rays once more.
// key_nodes: that hierarchy's nodes (the trunk's box).
__global__ __launch_bounds__(256) void k_shade(TraceParams p, ShadeBuffers b, const float4* __restrict__ hit_records,
                                               const uint32_t* __restrict__ hit_ci, uint8_t* __restrict__ missed,
                                               uint2* __restrict__ next_keys, const float4* __restrict__ key_nodes, uint32_t cost_bits,
                                               uint32_t key_bits) {
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    if (tid >= p.num_rays) return;
    if (missed && missed[tid]) return;           // not walked by this call: it would have missed again, which changes nothing
    const float4 r0 = b.rays[2 * (size_t)tid], r1 = b.rays[2 * (size_t)tid + 1];
    const float4 h = hit_records[tid];
    if (missed && h.x == r1.w) missed[tid] = 1u;
    Hit best;
    best.t = h.x; best.u = h.y; best.v = h.z; best.w = h.w;
    float4 n0, n1;
    shade_ray(p, b, tid, r0, r1, best, hit_ci[tid], n0, n1);
    if (next_keys) next_keys[tid] = make_uint2(chord_key(n0, n1, key_nodes[0], key_nodes[1], cost_bits, key_bits), tid);
}

int trace_workspace(rt_device* dev, uint32_t num_rays) {
    // 8 ray queues of the first traversal, 8 of the second (the listed rays), and the length of the list
    if (!dev->queue_counter) RT_HIP(hipMalloc((void**)&dev->queue_counter, 1024));
    if (num_rays <= dev->hit_capacity) return RT_OK;
    RT_HIP(hipStreamSynchronize(dev->stream));
    if (dev->hit_records) { (void)hipFree(dev->hit_records); dev->hit_records = nullptr; }
    if (dev->hit_ci) { (void)hipFree(dev->hit_ci); dev->hit_ci = nullptr; }
    if (dev->ray_keys_a) { (void)hipFree(dev->ray_keys_a); dev->ray_keys_a = nullptr; }
    if (dev->ray_keys_b) { (void)hipFree(dev->ray_keys_b); dev->ray_keys_b = nullptr; }
    if (dev->listed_rays) { (void)hipFree(dev->listed_rays); dev->listed_rays = nullptr; }
    if (dev->missed) { (void)hipFree(dev->missed); dev->missed = nullptr; }
    dev->missed_valid = false;
    dev->keys_ready = false;
    dev->hit_capacity = 0;
    hipError_t e = hipMalloc((void**)&dev->hit_records, (size_t)num_rays * 16u);
    if (e == hipSuccess) e = hipMalloc((void**)&dev->hit_ci, (size_t)num_rays * 4u);
    if (e == hipSuccess) e = hipMalloc((void**)&dev->ray_keys_a, (size_t)num_rays * 8u);
    if (e == hipSuccess) e = hipMalloc((void**)&dev->ray_keys_b, (size_t)num_rays * 8u);
    if (e == hipSuccess) e = hipMalloc((void**)&dev->listed_rays, (size_t)num_rays * 8u);
    if (e == hipSuccess) e = hipMalloc((void**)&dev->missed, (size_t)num_rays);
    if (e != hipSuccess) return e == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_HIP_BASE + (int)e;
    dev->hit_capacity = num_rays;
    return RT_OK;
}

// what a short list of rays is walked in parts with (ListedSplit), and the host-visible word the list lengths are reported in
int split_workspace(rt_device* dev) {
    if (dev->split_cand) return RT_OK;
    hipError_t e = hipSuccess;
    if (!dev->listed_seen_host) {
        e = hipHostMalloc((void**)&dev->listed_seen_host, sizeof(uint32_t), hipHostMallocMapped);
        if (e == hipSuccess) {
            *dev->listed_seen_host = 0u;
            e = hipHostGetDevicePointer((void**)&dev->listed_seen_dev, dev->listed_seen_host, 0);
        }
    }
    const size_t parts = (size_t)RT_SPLIT_RAYS << 6;
    if (e == hipSuccess && !dev->split_count) e = hipMalloc((void**)&dev->split_count, parts * sizeof(uint32_t));
    if (e == hipSuccess && !dev->listed_again) e = hipMalloc((void**)&dev->listed_again, (size_t)RT_SPLIT_RAYS * sizeof(uint2));
    if (e == hipSuccess) e = hipMalloc((void**)&dev->split_cand, parts * 2u * RT_SPLIT_CANDIDATES * sizeof(float4));
    if (e != hipSuccess) return e == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_HIP_BASE + (int)e;
    return RT_OK;
}

// buffers of the private hierarchy copy made for trace calls without an accel
int convert_workspace(rt_device* dev, uint32_t num_prims, const rt_levels& lv) {
    if (num_prims <= dev->conv_capacity) return RT_OK;
    RT_HIP(hipStreamSynchronize(dev->stream));
    if (dev->conv_pairs) { (void)hipFree(dev->conv_pairs); dev->conv_pairs = nullptr; }
    if (dev->conv_tris) { (void)hipFree(dev->conv_tris); dev->conv_tris = nullptr; }
    if (dev->conv_quads) { (void)hipFree(dev->conv_quads); dev->conv_quads = nullptr; }
    dev->conv_capacity = 0;
    // both sizes grow with the primitive count
    hipError_t e = hipMalloc((void**)&dev->conv_pairs, rt_aligned_node_slots(&lv) * sizeof(rt_aabb));
    if (e == hipSuccess) e = hipMalloc((void**)&dev->conv_tris, (size_t)lv.n[0] * RT_LEAF_STRIDE * sizeof(float4));
    if (e == hipSuccess) e = hipMalloc((void**)&dev->conv_quads, rt_quad_records(&lv) * RT_QUAD_STRIDE * sizeof(float4));
    if (e != hipSuccess) return e == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_HIP_BASE + (int)e;
    dev->conv_capacity = num_prims;
    return RT_OK;
}

int trace_common(rt_device* dev, const rt_trace_rays_info* info, int use_bvh, const rt_accel* accel, const rt_buffer* indices,
                 const rt_buffer* vertices, const rt_buffer* materials, const rt_buffer* texture_ids,
                 const rt_buffer* texture_atlas, uint32_t atlas_elements, rt_buffer* rays, rt_buffer* old_rays,
                 const rt_buffer* ray_pixels, rt_buffer* scattered_light, rt_buffer* emitted_light, const rt_buffer* bvh,
                 const rt_image_rows* rows) {
    if (!dev || !info || !indices || !vertices || !materials || !texture_ids || !texture_atlas || !rays || !old_rays ||
        !ray_pixels || !scattered_light || !emitted_light)
        return RT_ERR_INVALID_ARGUMENT;
    if (use_bvh && !bvh) return RT_ERR_INVALID_ARGUMENT;
    unsigned long long* stats = dev->trace_stats;
    const uint32_t nr = info->num_rays;
    if (info->screen_dimensions[0] <= 0 || !info->max_rays_per_pixel) return RT_ERR_INVALID_ARGUMENT;
    const uint64_t row_len = (uint64_t)info->screen_dimensions[0] * info->max_rays_per_pixel;
    const uint32_t row_stride = rows ? rows->row_stride : 1u;
    const uint64_t first_ray = rows ? (uint64_t)rows->first_row * row_len : 0u;
    if (!row_stride || row_len > 0xffffffffull || first_ray > 0xffffffffull) return RT_ERR_INVALID_ARGUMENT;
    if (info->screen_dimensions[1] <= 0) return RT_ERR_INVALID_ARGUMENT;
    if (rows) {
        if (!rt_rows_fit(rows->first_row, rows->num_rows, rows->row_stride, (uint32_t)info->screen_dimensions[1])) return RT_ERR_INVALID_ARGUMENT;
        if ((uint64_t)rows->num_rows * row_len != nr) return RT_ERR_INVALID_ARGUMENT;
    } else if ((uint64_t)info->screen_dimensions[1] * row_len != nr) {
        return RT_ERR_INVALID_ARGUMENT;
    }
    if (rays->bytes < (size_t)nr * 32u || old_rays->bytes < (size_t)nr * 32u || ray_pixels->bytes < (size_t)nr * 4u ||
        scattered_light->bytes < (size_t)nr * 16u || emitted_light->bytes < (size_t)nr * 16u)
        return RT_ERR_BUFFER_TOO_SMALL;
    if (indices->bytes < (size_t)info->num_indices * 4u || texture_atlas->bytes < (size_t)atlas_elements * 16u)
        return RT_ERR_BUFFER_TOO_SMALL;
    if (!nr) return RT_OK;
    int rc = rt_set_device(dev);
    if (rc) return rc;

    // keys the last call's shading left for these rays hold for this call only; whatever this call does to the rays, the
    // keys are its own or none
    const bool keys_were_ready = dev->keys_ready;
    dev->keys_ready = false;
    // the traversal kernel and its tuning are device settings (rt_device_set_trace_kernel, rt_device_set_trace_tuning)
    const rt_trace_tuning& tune = dev->tuning;
    const bool divide = dev->slab_division;
    const bool simple_kernel = dev->trace_kernel == RT_TRACE_KERNEL_SIMPLE;
    const bool spatial_kernel = !divide && dev->trace_kernel == RT_TRACE_KERNEL_SPATIAL && accel && accel->sp_built;
    const bool quad_kernel = !divide && (dev->trace_kernel == RT_TRACE_KERNEL_QUAD || dev->trace_kernel == RT_TRACE_KERNEL_SPATIAL);
    // (lanes with a leaf waiting that start a triangle round: the walk on the library's own hierarchy meets few
    // leaves and does better with fuller rounds)
    const uint32_t leaf_threshold = tune.leaf_threshold ? tune.leaf_threshold : (spatial_kernel ? 32u : 16u);
    const uint32_t blocks_per_cu = tune.blocks_per_cu;
    const uint32_t sort_rays = tune.sort_rays == 2u ? ((spatial_kernel && !stats) ? 0u : 1u) : tune.sort_rays;
    dev->listed_valid = false;

    TraceParams p;
    p.width = info->screen_dimensions[0];
    p.num_indices = info->num_indices;
    p.num_rays = nr;
    p.rpp = info->max_rays_per_pixel;
    for (int k = 0; k < 3; k++) p.seed[k] = info->rng_seed[k];
    p.first_ray = (uint32_t)first_ray;
    p.row_len = (uint32_t)row_len;
    p.row_stride = row_stride;
    p.atlas_elems = atlas_elements;
    p.num_prims = info->num_indices / 3u;
    p.num_materials = (uint32_t)(materials->bytes / sizeof(rt_pbr_material));
    p.num_textures = (uint32_t)(texture_ids->bytes / sizeof(rt_texture_id));
    p.leaf_threshold = leaf_threshold < 1u ? 1u : (leaf_threshold > 64u ? 64u : leaf_threshold);
    const uint32_t cost_bits = tune.cost_bits;
    // (free lanes that start a refill: taking rays from the spatial kernel's stock is cheap, a refill of the others is not)
    p.refill_threshold = tune.refill_threshold ? tune.refill_threshold : (spatial_kernel ? 8u : 16u);
    p.deal_shift = tune.deal_shift;
    p.stuck_threshold = tune.stuck_threshold ? tune.stuck_threshold : 4u;
    p.retire_ticks = 100u * tune.retire_us;           // of the 100 MHz clock the kernel reads
    p.beyond = dev->walk_entry_margin;
    const uint32_t deal_block = 1u << p.deal_shift;
    p.queue_chunk = ((((nr + deal_block - 1u) / deal_block) + 7u) / 8u) * deal_block;      // whole blocks, dealt to the eight queues in turn
    p.lv.count = 0;
    if (use_bvh) {
        if (!rt_make_levels(p.num_prims, &p.lv)) return RT_ERR_TOO_FEW_PRIMITIVES;
        if (bvh->bytes < (size_t)rt_bvh_node_count(p.num_prims) * 32u) return RT_ERR_BUFFER_TOO_SMALL;
        if (accel && accel->num_prims != p.num_prims) return RT_ERR_INVALID_ARGUMENT;
    }
    p.trunk_children = p.lv.count >= 2u ? p.lv.off[p.lv.count - 2u] : 0u;
    p.walk_hist = 0u;
    ShadeBuffers b;
    b.indices = (const uint32_t*)indices->ptr;
    b.verts = (const rt_vertex*)vertices->ptr;
    b.mats = (const rt_pbr_material*)materials->ptr;
    b.tex = (const rt_texture_id*)texture_ids->ptr;
    b.atlas = (const uint4*)texture_atlas->ptr;
    b.rays = (float4*)rays->ptr;
    b.old_rays = (float4*)old_rays->ptr;
    b.ray_pixels = (const uint32_t*)ray_pixels->ptr;
    b.scattered = (float4*)scattered_light->ptr;
    b.emitted = (float4*)emitted_light->ptr;
    const dim3 grid((nr + 255u) / 256u), block(256);
    const float4* nodes = bvh ? (const float4*)bvh->ptr : nullptr;
    const float4* st = accel ? (const float4*)accel->sorted_tris : nullptr;
    hipStream_t s = dev->stream;

    if (dev->drop_missed || (use_bvh && !simple_kernel)) {      // (the one-thread-per-ray paths need no workspace of their own)
        rc = trace_workspace(dev, nr);
        if (rc) return rc;
    }
    // drop-missed-rays mode (rt_device_set_drop_missed_rays): one byte per ray, set by the shading step of the call in which
    // the ray missed. The marks belong to one set of rays: they start from zero whenever the rays may be new (new camera
    // rays, a new hierarchy, another ray buffer or count, the mode switched on).
    uint8_t* missed = nullptr;
    if (dev->drop_missed) {
        if (!dev->missed_valid || dev->missed_rays != rays->ptr || dev->missed_count != nr) {
            RT_HIP(hipMemsetAsync(dev->missed, 0, nr, s));
            dev->missed_valid = true;
            dev->missed_rays = rays->ptr;
            dev->missed_count = nr;
        }
        missed = dev->missed;
    }
    if (!use_bvh) {
        if (stats) hipLaunchKernelGGL((k_trace<false, true>), grid, block, 0, s, p, b, nodes, stats, missed);
        else hipLaunchKernelGGL((k_trace<false, false>), grid, block, 0, s, p, b, nodes, stats, missed);
        return rt_launch_status();
    }
    if (simple_kernel) {
        if (divide) {
            if (stats) hipLaunchKernelGGL((k_trace<true, true, true>), grid, block, 0, s, p, b, nodes, stats, missed);
            else hipLaunchKernelGGL((k_trace<true, false, true>), grid, block, 0, s, p, b, nodes, stats, missed);
        } else {
            if (stats) hipLaunchKernelGGL((k_trace<true, true>), grid, block, 0, s, p, b, nodes, stats, missed);
            else hipLaunchKernelGGL((k_trace<true, false>), grid, block, 0, s, p, b, nodes, stats, missed);
        }
        return rt_launch_status();
    }

    RT_HIP(hipMemsetAsync(dev->queue_counter, 0, 1024, s));
    // Where the rays are walked in a sorted order, the dead ones lie behind the first *live entries and are never handed
    // out; the default kernel, taking rays as they come, passes over marked rays when it fills its stock; the other kernels
    // without an order walk them all again (they miss again) and only the marks are kept
    uint32_t* live = (missed && sort_rays) ? dev->queue_counter + 129 : nullptr;
    // persistent grid: enough workgroups to fill every CU, never more than there are rays to hand out
    uint32_t groups = (uint32_t)dev->compute_units * blocks_per_cu;
    if (groups > grid.x) groups = grid.x;
    const float4* ray_in = (const float4*)rays->ptr;
    // the walk runs on the private copy of the hierarchy; a call without an accel gets one made from its buffers
    const float4* pairs = nullptr;
    const float4* quads = nullptr;
    const float4* key_nodes = nodes;      // the hierarchy the walk-length estimate is taken on
    const float4* key_pairs = nullptr;
    // (sibling-pair records: read by the one-level walk - also the counting variant and the fallback for what the
    // two-level kernels cannot address - and by the ray-order probe; an accel has them made on first use after a build)
    const bool offsets_fit = rt_aligned_node_slots(&p.lv) * sizeof(rt_aabb) <= 0xffffffffull && (size_t)p.lv.n[0] * 128u <= 0xffffffffull;
    const bool one_level_walk = stats || divide || !((spatial_kernel || quad_kernel) && offsets_fit && p.lv.count >= 3u);
    if (accel) {
        quads = (const float4*)accel->quads;
        if (one_level_walk) {
            rc = rt_accel_pairs(const_cast<rt_accel*>(accel), false, (const rt_aabb*)bvh->ptr, &pairs);
            if (rc) return rc;
        }
    } else {
        rc = convert_workspace(dev, p.num_prims, p.lv);
        if (rc) return rc;
        rc = rt_private_copy_from_reference(dev, p.num_prims, (const rt_aabb*)bvh->ptr, b.indices, b.verts, dev->conv_pairs,
                                            dev->conv_tris, dev->conv_quads);
        if (rc) return rc;
        pairs = (const float4*)dev->conv_pairs;
        st = (const float4*)dev->conv_tris;
        quads = (const float4*)dev->conv_quads;
    }
    uint2* order = nullptr;
    // keys for the next call, made by this call's shading kernel (the default kernel's probe-free keys only; not with
    // dropped rays, whose keys also carry the live count)
    bool leave_keys = false;
    uint32_t leave_cost_bits = 0u, leave_key_bits = 0u;
    if (sort_rays) {
        // walk-length estimate from every fourth pair of nodes at depth 10 (or the deepest level above the leaves of a
        // small tree): deeper, smaller boxes tell long walks from short ones better than all the boxes of depth 8 do
        // (on the library's own hierarchy walks are short and alike: all 16 pairs of depth 5 are enough to order them)
        const bool on_spatial = spatial_kernel && !stats;
        const rt_levels& kl = on_spatial ? accel->sp_lv : p.lv;
        const uint32_t root_level = kl.count - 1u;
        const uint32_t probe_depth_env = tune.cost_depth;
        const uint32_t probe_depth = probe_depth_env >= 2u ? probe_depth_env : (on_spatial ? 5u : 10u);
        const uint32_t depth = root_level > probe_depth ? probe_depth : (root_level > 1u ? root_level - 1u : 1u);
        // (... and better still, and nearly free: the length of the ray's way through the trunk's box; cost_chord = 0
        // or a cost_depth bring the probe back)
        const bool chord_only = tune.cost_chord == 2u ? (on_spatial && probe_depth_env < 2u) : tune.cost_chord != 0u;
        const uint32_t nodes_there = chord_only ? 0u : kl.n[root_level - depth];
        const uint32_t probe_stride = tune.cost_stride ? tune.cost_stride : (on_spatial ? 1u : 4u);
        // sorting passes = bytes of the key that count (the class, then as much of the locality part as fits)
        const uint32_t key_passes = tune.key_passes < 1u ? (on_spatial ? 2u : 3u) : tune.key_passes;
        uint32_t cost_shift = 0u;
        while (((nodes_there / probe_stride) >> cost_shift) > (1u << cost_bits) - 1u) cost_shift++;
        key_nodes = on_spatial ? (const float4*)accel->sp_nodes : nodes;
        key_pairs = pairs;
        if (nodes_there && accel) {          // the probe reads pair records
            rc = rt_accel_pairs(const_cast<rt_accel*>(accel), on_spatial, on_spatial ? accel->sp_nodes : (const rt_aabb*)bvh->ptr, &key_pairs);
            if (rc) return rc;
            if (!on_spatial) pairs = key_pairs;
        }
        const uint32_t key_cost_bits = key_passes == 1u ? 8u : cost_bits, key_bits = 8u * key_passes;
        leave_keys = on_spatial && chord_only && !missed;
        leave_cost_bits = key_cost_bits; leave_key_bits = key_bits;
        const bool have_keys = leave_keys && keys_were_ready && dev->keys_rays == rays->ptr && dev->keys_accel == (const void*)accel &&
                               dev->keys_count == nr && dev->keys_cost_bits == key_cost_bits && dev->keys_bits == key_bits;
        if (!have_keys)
            hipLaunchKernelGGL(k_ray_keys, grid, block, 0, s, nr, ray_in, key_nodes, key_pairs, 1u << (depth - 1u), (nodes_there + 1u) / 2u,
                               probe_stride, cost_shift, key_cost_bits, key_bits, dev->ray_keys_a, (const uint8_t*)(live ? missed : nullptr),
                               live);
        rc = rt_sort_pairs(dev, dev->ray_keys_a, dev->ray_keys_b, nr, key_passes, &order);      // 8, 16 or 24-bit keys
        if (rc) return rc;
    }
    // waiting leaves; the counting variant also keeps the entry distances of noted children, one row per depth
    const size_t lds_bytes = ((size_t)kFifo * 512u + (stats ? (size_t)p.lv.count * 256u : 0u)) * sizeof(uint32_t);
#define RT_LAUNCH_TRAVERSE(STATS, DIV)                                                                                         \
    hipLaunchKernelGGL((k_traverse<STATS, DIV>), dim3(groups), block, lds_bytes, s, p, ray_in, nodes, pairs, st, dev->hit_records, \
                       dev->hit_ci, dev->queue_counter, order, stats, (const uint32_t*)live)
    hipEvent_t* timing = nullptr;
    if (dev->time_traverse) {
        if (dev->timing_count == dev->timing_capacity) {      // grow the pool of event pairs
            const uint32_t cap = dev->timing_capacity ? 2u * dev->timing_capacity : 64u;
            hipEvent_t* ev = new (std::nothrow) hipEvent_t[2u * cap];
            if (!ev) return RT_ERR_OUT_OF_MEMORY;
            for (uint32_t e = 0; e < 2u * dev->timing_capacity; e++) ev[e] = dev->timing_events[e];
            for (uint32_t e = 2u * dev->timing_capacity; e < 2u * cap; e++) RT_HIP(hipEventCreate(&ev[e]));
            delete[] dev->timing_events;
            dev->timing_events = ev;
            dev->timing_capacity = cap;
        }
        timing = dev->timing_events + 2u * dev->timing_count;
        RT_HIP(hipEventRecord(timing[0], s));
    }
    // (k_traverse_quad addresses its records and the leaves with 32-bit byte offsets: offsets_fit)
    if (stats && divide) RT_LAUNCH_TRAVERSE(true, true);
    else if (stats) RT_LAUNCH_TRAVERSE(true, false);
    else if (divide) RT_LAUNCH_TRAVERSE(false, true);
    else if (spatial_kernel && offsets_fit && p.lv.count >= 3u && accel->sp_lv.n[0] <= RT_SPATIAL_SLOTS_MAX) {
        // first the walk on the library's own hierarchy, then the reference's walk for the rays it has listed
        uint32_t* listed = dev->queue_counter + 128;
        dev->listed_valid = true;
        TraceParams ps = p;
        ps.lv = accel->sp_lv;
        ps.trunk_children = ps.lv.count >= 2u ? ps.lv.off[ps.lv.count - 2u] : 0u;
        ps.walk_hist = dev->walk_stats_hist ? 1u : 0u;
#define RT_LAUNCH_SPATIAL(COUNT)                                                                                              \
    hipLaunchKernelGGL((k_traverse_spatial<COUNT>), dim3(groups), block,                                                      \
                       (size_t)(kFifo * 512u + 4u * kStockFields * 64u + ((COUNT) ? 64u : 0u)) * sizeof(uint32_t), s, ps, ray_in, \
                       (const float4*)accel->sp_nodes, (const float4*)accel->sp_pairs, (const float4*)accel->sp_quads,          \
                       (const float4*)accel->sp_tris, dev->hit_records, dev->hit_ci, dev->queue_counter, order, st,            \
                       dev->listed_rays, listed, dev->walk_stats, (const uint32_t*)live, (const float4*)accel->outlier_leaves,   \
                       (const uint32_t*)(accel->cells + RT_CELLS_OUTLIERS), (const uint8_t*)((missed && !live) ? missed : nullptr))
        if (dev->walk_stats) RT_LAUNCH_SPATIAL(true);
        else RT_LAUNCH_SPATIAL(false);
#undef RT_LAUNCH_SPATIAL
        // The listed rays. A short list is walked in parts (k_traverse_quad<true>, k_listed_merge) - but only where a list
        // is expected at all: the two extra launches cost a few microseconds each even when they find nothing to do, and
        // most scenes never list a ray. What to expect is what a recent call on this device saw (the kernels leave the
        // length of their list in a word of host memory; it may be a call or two old, which only ever costs time: the
        // plain form alone is complete for any list).
        rc = split_workspace(dev);
        if (rc) return rc;
        ListedSplit split;
        split.cap = 0u; split.depth = 0u; split.cand = nullptr; split.cand_count = nullptr;
        split.again = nullptr; split.again_count = nullptr;
        split.seen = dev->listed_seen_dev;
        const uint32_t ref_root_level = p.lv.count - 1u;
        const bool in_parts = dev->listed_split == 1 || (dev->listed_split == 0 && *(volatile uint32_t*)dev->listed_seen_host != 0u);
        if (in_parts) {
            // the parts: subtrees of the nodes at the deepest depth up to 6 that the walk's steps pass through (every second
            // one, from the trunk or from its children) and that has records of four grandchildren below it
            uint32_t depth = ref_root_level & 1u;
            while (depth + 2u <= 6u && depth + 2u + 2u <= ref_root_level) depth += 2u;
            split.cap = RT_SPLIT_RAYS; split.depth = depth;
            split.cand = dev->split_cand; split.cand_count = dev->split_count;
            const uint32_t part_groups = min(groups, (RT_SPLIT_RAYS << depth) / 256u);
            hipLaunchKernelGGL((k_traverse_quad<true>), dim3(part_groups), block, (size_t)kFifo * 512u * sizeof(uint32_t), s, p, ray_in, nodes,
                               pairs, quads, st, dev->hit_records, dev->hit_ci, dev->queue_counter + 192, (const uint2*)dev->listed_rays,
                               (const uint32_t*)listed, split);
            hipLaunchKernelGGL(k_listed_merge, dim3(RT_SPLIT_RAYS / 256u), block, 0, s, ray_in, (const uint2*)dev->listed_rays,
                               (const uint32_t*)listed, split, dev->hit_records, dev->hit_ci, dev->listed_again, dev->queue_counter + 131);
            split.again = dev->listed_again; split.again_count = dev->queue_counter + 131;
            split.seen = nullptr;                 // (the merge kernel has reported the length)
        }
        hipLaunchKernelGGL((k_traverse_quad<false>), dim3(groups), block, (size_t)kFifo * 512u * sizeof(uint32_t), s, p, ray_in, nodes, pairs,
                           quads, st, dev->hit_records, dev->hit_ci, dev->queue_counter + 64, (const uint2*)dev->listed_rays,
                           (const uint32_t*)listed, split);
    } else if (quad_kernel && offsets_fit && p.lv.count >= 3u && quads)
        hipLaunchKernelGGL((k_traverse_quad<false>), dim3(groups), block, (size_t)kFifo * 512u * sizeof(uint32_t), s, p, ray_in, nodes, pairs,
                           quads, st, dev->hit_records, dev->hit_ci, dev->queue_counter, order, (const uint32_t*)live, ListedSplit{});
    else RT_LAUNCH_TRAVERSE(false, false);
#undef RT_LAUNCH_TRAVERSE
    if (timing) {
        RT_HIP(hipEventRecord(timing[1], s));
        dev->timing_count++;          // only a complete pair counts
    }
    rc = rt_launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(k_shade, grid, block, 0, s, p, b, (const float4*)dev->hit_records, (const uint32_t*)dev->hit_ci, missed,
                       leave_keys ? dev->ray_keys_a : (uint2*)nullptr, leave_keys ? (const float4*)accel->sp_nodes : (const float4*)nullptr,
                       leave_cost_bits, leave_key_bits);
    rc = rt_launch_status();
    if (rc) return rc;
    if (leave_keys) {
        dev->keys_ready = true;
        dev->keys_rays = rays->ptr; dev->keys_accel = (const void*)accel; dev->keys_count = nr;
        dev->keys_cost_bits = leave_cost_bits; dev->keys_bits = leave_key_bits;
    }
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_device_set_trace_stats(rt_device* dev, rt_buffer* stats) {
    if (!dev || (stats && stats->bytes < 32u)) return RT_ERR_INVALID_ARGUMENT;
    dev->trace_stats = stats ? (unsigned long long*)stats->ptr : nullptr;
    return RT_OK;
}

int rt_device_set_walk_stats(rt_device* dev, rt_buffer* stats) {
    if (!dev || (stats && stats->bytes < 32u)) return RT_ERR_INVALID_ARGUMENT;
    dev->walk_stats = stats ? (unsigned long long*)stats->ptr : nullptr;
    dev->walk_stats_hist = stats && stats->bytes >= 160u;      // room for 16 more counters: steps by depth
    return RT_OK;
}

int rt_device_set_trace_kernel(rt_device* dev, int kernel) {
    if (!dev || kernel < RT_TRACE_KERNEL_QUAD || kernel > RT_TRACE_KERNEL_SPATIAL) return RT_ERR_INVALID_ARGUMENT;
    dev->trace_kernel = kernel;
    return RT_OK;
}

int rt_device_set_drop_missed_rays(rt_device* dev, int enable) {
    if (!dev) return RT_ERR_INVALID_ARGUMENT;
    dev->drop_missed = enable != 0;
    dev->missed_valid = false;
    dev->keys_ready = false;
    return RT_OK;
}

int rt_device_set_slab_division(rt_device* dev, int enable) {
    if (!dev) return RT_ERR_INVALID_ARGUMENT;
    dev->slab_division = enable != 0;
    return RT_OK;
}

int rt_device_set_walk_margins(rt_device* dev, float entry_margin, float sliver_pad) {
    if (!dev || !(entry_margin >= 1.0f && entry_margin <= 2.0f) || !(sliver_pad >= 0.0f && sliver_pad <= 1.0f)) return RT_ERR_INVALID_ARGUMENT;
    dev->walk_entry_margin = entry_margin;
    dev->walk_sliver_pad = sliver_pad;
    return RT_OK;
}

int rt_device_get_trace_tuning(rt_device* dev, rt_trace_tuning* out) {
    if (!dev || !out) return RT_ERR_INVALID_ARGUMENT;
    *out = dev->tuning;
    return RT_OK;
}

int rt_device_set_trace_tuning(rt_device* dev, const rt_trace_tuning* tuning) {
    if (!dev || !tuning) return RT_ERR_INVALID_ARGUMENT;
    dev->tuning = rt_clamp_tuning(*tuning);
    return RT_OK;
}

int rt_device_set_traverse_timing(rt_device* dev, int enable) {
    if (!dev) return RT_ERR_INVALID_ARGUMENT;
    dev->time_traverse = enable != 0;
    dev->timing_count = 0;
    return RT_OK;
}

int rt_device_traverse_timing(rt_device* dev, double* total_ms, uint32_t* launches) {
    if (!dev || !total_ms || !launches) return RT_ERR_INVALID_ARGUMENT;
    int rc = rt_set_device(dev);
    if (rc) return rc;
    double sum = 0.0;
    for (uint32_t e = 0; e < dev->timing_count; e++) {
        float ms = 0.0f;
        RT_HIP(hipEventSynchronize(dev->timing_events[2u * e + 1u]));
        RT_HIP(hipEventElapsedTime(&ms, dev->timing_events[2u * e], dev->timing_events[2u * e + 1u]));
        sum += (double)ms;
    }
    *total_ms = sum;
    *launches = dev->timing_count;
    dev->timing_count = 0;
    return RT_OK;
}

int rt_device_listed_rays(rt_device* dev, uint32_t* count) {
    if (!dev || !count) return RT_ERR_INVALID_ARGUMENT;
    *count = 0u;
    if (!dev->queue_counter || !dev->listed_valid) return RT_OK;        // nothing traced yet, or not by the SPATIAL kernel
    int rc = rt_set_device(dev);
    if (rc) return rc;
    RT_HIP(hipMemcpyAsync(count, dev->queue_counter + 128, sizeof(uint32_t), hipMemcpyDeviceToHost, dev->stream));
    RT_HIP(hipStreamSynchronize(dev->stream));
    return RT_OK;
}

int rt_device_set_listed_split(rt_device* dev, int mode) {
    if (!dev || mode < 0 || mode > 2) return RT_ERR_INVALID_ARGUMENT;
    dev->listed_split = mode;
    return RT_OK;
}

int rt_device_listed_rays_detail(rt_device* dev, uint32_t* total, uint32_t* during_walk, uint32_t* whole_after_parts) {
    if (!dev || !total || !during_walk || !whole_after_parts) return RT_ERR_INVALID_ARGUMENT;
    *total = *during_walk = *whole_after_parts = 0u;
    if (!dev->queue_counter || !dev->listed_valid) return RT_OK;
    int rc = rt_set_device(dev);
    if (rc) return rc;
    // list length, live count of drop-missed-rays mode, listed during the walk, rays whose parts ran out of room
    uint32_t words[4] = {0u, 0u, 0u, 0u};
    RT_HIP(hipMemcpyAsync(words, dev->queue_counter + 128, sizeof(words), hipMemcpyDeviceToHost, dev->stream));
    RT_HIP(hipStreamSynchronize(dev->stream));
    *total = words[0]; *during_walk = words[2]; *whole_after_parts = words[3];
    return RT_OK;
}

int rt_trace_rays(rt_device* dev, const rt_trace_rays_info* info, int use_bvh, const rt_buffer* indices,
                  const rt_buffer* vertices, const rt_buffer* materials, const rt_buffer* texture_ids,
                  const rt_buffer* texture_atlas, uint32_t atlas_elements, rt_buffer* rays, rt_buffer* old_rays,
                  const rt_buffer* ray_pixels, rt_buffer* scattered_light, rt_buffer* emitted_light, const rt_buffer* bvh,
                  const rt_image_rows* rows) {
    return trace_common(dev, info, use_bvh, nullptr, indices, vertices, materials, texture_ids, texture_atlas, atlas_elements,
                        rays, old_rays, ray_pixels, scattered_light, emitted_light, bvh, rows);
}

int rt_trace_rays_accel(rt_device* dev, const rt_trace_rays_info* info, const rt_accel* accel, const rt_buffer* indices,
                        const rt_buffer* vertices, const rt_buffer* materials, const rt_buffer* texture_ids,
                        const rt_buffer* texture_atlas, uint32_t atlas_elements, rt_buffer* rays, rt_buffer* old_rays,
                        const rt_buffer* ray_pixels, rt_buffer* scattered_light, rt_buffer* emitted_light, const rt_buffer* bvh,
                        const rt_image_rows* rows) {
    if (!accel) return RT_ERR_INVALID_ARGUMENT;
    return trace_common(dev, info, 1, accel, indices, vertices, materials, texture_ids, texture_atlas, atlas_elements, rays,
                        old_rays, ray_pixels, scattered_light, emitted_light, bvh, rows);
}

}  // extern "C"
