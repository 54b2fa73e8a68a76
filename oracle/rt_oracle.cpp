// CPU oracle for the LBVH-build + ray-trace path.
//
// TEST INFRASTRUCTURE ONLY -- never linked into or called from the product library.
// PARITY UNPINNED by reference tests (the reference has none); see rt_oracle.h.
//
// Scalar restatement of the reference's HLSL kernels: one function per HLSL `main`,
// one loop iteration per GPU thread, the same buffer element layouts, and the host
// sequencing of src/RaytracerPipeline.cpp on top. Built with
//   g++ -O2 -fopenmp -ffp-contract=off -fno-fast-math
// so that every float expression rounds once per written operation.
//
// HLSL -> C++ rules used throughout:
//   saturate(NaN) = 0; min/max = IEEE minNum/maxNum with -0 < +0 (hmin/hmax below); round = rintf
//   (ties to even); float->uint truncates, negatives and NaN give 0, overflow
//   saturates; rcp(x) = 1.0f/x; normalize(v) = v / sqrt(dot(v,v)); length = sqrt(dot);
//   dot(a,b) = ((a.x*b.x + a.y*b.y) + a.z*b.z); lerp(a,b,t) = a + t*(b-a);
//   pow(x,5) = ((x*x)*(x*x))*x; sin/cos = the 11/10-degree minimax polynomials of
//   orc_sincos; constant-buffer float4x4 is read column-major (M[r][c] = mem[4c+r]).

#include "rt_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <random>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

struct f3 { float x, y, z; };

inline f3 mk(float x, float y, float z) { f3 r = {x, y, z}; return r; }
inline f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
inline f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
inline f3 operator*(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
inline f3 operator*(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
inline f3 operator*(float s, f3 a) { return mk(s * a.x, s * a.y, s * a.z); }
inline f3 operator/(f3 a, f3 b) { return mk(a.x / b.x, a.y / b.y, a.z / b.z); }
inline f3 operator/(f3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
inline float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline f3 cross(f3 a, f3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline f3 normalize(f3 a) { return a / sqrtf(dot(a, a)); }
// HLSL min/max: the non-NaN operand if one is NaN, and -0 orders below +0 (what AMD's v_min_f32 /
// v_max_f32 return; libm's fminf/fmaxf leave the zero case open)
inline float hmin(float a, float b) {
    if (a != a) return b;
    if (b != b) return a;
    if (a == b) return std::signbit(a) ? a : b;
    return a < b ? a : b;
}
inline float hmax(float a, float b) {
    if (a != a) return b;
    if (b != b) return a;
    if (a == b) return std::signbit(a) ? b : a;
    return a > b ? a : b;
}
inline f3 vmin(f3 a, f3 b) { return mk(hmin(a.x, b.x), hmin(a.y, b.y), hmin(a.z, b.z)); }
inline f3 vmax(f3 a, f3 b) { return mk(hmax(a.x, b.x), hmax(a.y, b.y), hmax(a.z, b.z)); }
inline float saturate(float x) { return hmin(hmax(x, 0.0f), 1.0f); }   // NaN -> 0
inline f3 saturate(f3 a) { return mk(saturate(a.x), saturate(a.y), saturate(a.z)); }
inline float pow5(float x) { float x2 = x * x; return (x2 * x2) * x; }
inline uint32_t f2u(float x) {
    if (!(x > 0.0f)) return 0u;                  // negatives, -0, NaN
    if (x >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)x;
}
inline float asfloat(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
inline uint32_t asuint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline f3 pos(const orc_vertex& v) { return mk(v.px, v.py, v.pz); }

// shader/Random.hlsli:3-9
inline uint32_t xorshift(uint32_t& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }
// shader/Random.hlsli:63-66
inline float rnd(uint32_t& s) { return saturate((float)xorshift(s) * 2.3283064e-10f); }

// Deterministic sine/cosine: range reduction to [-pi/2, pi/2] followed by the
// 11-degree (sin) / 10-degree (cos) minimax polynomials, plain mul/add only.
inline void sincos_poly(float x, float* s, float* c) {
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

// shader/CS_GenerateMortonCodes.hlsl:30-38
inline uint32_t left_shift3(uint32_t v) {
    v |= (v << 16) & 0x030000ffu;
    v |= (v << 8) & 0x0300f00fu;
    v |= (v << 4) & 0x030c30c3u;
    v |= (v << 2) & 0x09249249u;
    return v;
}

// ---------------------------------------------------------------- tracing ----

struct Hit { float t, u, v, w; };

// shader/CS_TraceRays.hlsl:46-77 (Moeller-Trumbore, two-sided)
inline Hit intersect(const orc_ray& r, f3 v1, f3 v2, f3 v3) {
    f3 o = mk(r.ox, r.oy, r.oz), d = mk(r.dx, r.dy, r.dz);
    f3 e1 = v2 - v1, e2 = v3 - v1, tv = o - v1;
    f3 pv = cross(d, e2), qv = cross(tv, e1);
    float det = dot(e1, pv);
    float inv = (fabsf(det) < 1e-6f) ? 0.0f : 1.0f / det;
    float cw = rintf(fabsf(det) * inv);
    float t = dot(e2, qv) * inv, u = dot(tv, pv) * inv, v = dot(d, qv) * inv;
    Hit h;
    if (t < 1e-6f || u < 0.0f || v < 0.0f || u + v > 1.0f) { h.t = -1.0f; h.u = -1.0f; h.v = -1.0f; }
    else { h.t = t; h.u = u; h.v = v; }
    h.w = cw;
    return h;
}

// shader/CS_TraceRays.hlsl:81-97
inline float intersect_aabb(const orc_ray& r, const orc_aabb& b, bool true_div, f3 inv_d) {
    f3 o = mk(r.ox, r.oy, r.oz), d = mk(r.dx, r.dy, r.dz);
    f3 bmin = mk(b.mn[0], b.mn[1], b.mn[2]), bmax = mk(b.mx[0], b.mx[1], b.mx[2]);
    f3 t1, t2;
    if (true_div) { t1 = (bmin - o) / d; t2 = (bmax - o) / d; }
    else { t1 = (bmin - o) * inv_d; t2 = (bmax - o) * inv_d; }
    f3 lo = vmin(t1, t2), hi = vmax(t1, t2);
    float tmin = hmax(lo.x, hmax(lo.y, lo.z));
    float tmax = hmin(hi.x, hmin(hi.y, hi.z));
    if ((tmax < tmin) || (tmax <= 0.0f) || (tmax < r.tmin) || (tmin > r.tmax)) tmin = 1e30f;
    return tmin;
}

struct TriIdx { uint32_t a, b, c; };

// shader/CS_TraceRays.hlsl:100-113
inline void check_intersection(const orc_ray& r, uint32_t idx, const uint32_t* indices, const orc_vertex* verts,
                               Hit& best, TriIdx& bi) {
    uint32_t i1 = indices[idx], i2 = indices[idx + 1], i3 = indices[idx + 2];
    Hit h = intersect(r, pos(verts[i1]), pos(verts[i2]), pos(verts[i3]));
    bool use = (r.tmin <= h.t) && (r.tmax > h.t) && (h.t < best.t);
    if (use) { best = h; bi.a = i1; bi.b = i2; bi.c = i3; }
}

// shader/PerRayShading.hlsli:60-74
inline f3 sample_texture(const orc_texture_id& id, float u, float v, const orc_uint4* atlas, uint32_t atlas_elems) {
    uint32_t sx = f2u(rintf(u * (float)(id.width - 1u)));
    uint32_t sy = f2u(rintf(v * (float)(id.height - 1u)));
    uint32_t loc = sx + id.row_pitch * sy + id.offset;
    uint32_t e = loc / 2u;
    orc_uint4 px = {0, 0, 0, 0};                       // out-of-range structured loads return 0
    if (e < atlas_elems) px = atlas[e];
    uint32_t a = px.x, b = px.y;
    if (loc & 1u) { a = px.z; b = px.w; }
    f3 c = mk((float)(a & 0xffffu), (float)(a >> 16), (float)(b & 0xffffu));
    c = c * 1.5259022e-5f;
    return c * c;
}

// shader/PerRayShading.hlsli:79-106
inline f3 reflected_color(float rough, f3 f0, float ndl, float ndv, float ndh, float vdh) {
    float alpha = rough * rough;
    float sf = pow5(saturate(1.0f - vdh));
    f3 F = mk((f0.x + sf) - f0.x * sf, (f0.y + sf) - f0.y * sf, (f0.z + sf) - f0.z * sf);
    float a2 = alpha * alpha;
    float ndh2 = ndh * ndh;
    float sq = (ndh2 * a2 - ndh2) + 1.0f;
    float dden = (3.141592654f * sq) * sq;
    float k = alpha * 0.5f;
    float gx = (ndv - k * ndv) + k;
    float gy = (ndl - k * ndl) + k;
    float gden = gx * gy;
    float den = dden * gden + 1e-6f;
    return mk(((0.25f * F.x) * a2) / den, ((0.25f * F.y) * a2) / den, ((0.25f * F.z) * a2) / den);
}

// shader/PerRayShading.hlsli:110-133
inline f3 refracted_color(float rough, float metal, f3 albedo, float ndl, float ndv, float vdh, float ldv) {
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
inline float interp(float a1, float a2, float a3, float u, float v) {
    return (a1 * ((1.0f - u) - v) + a2 * u) + a3 * v;
}

void trace_one(uint32_t tid, const orc_trace_info* info, int use_bvh, bool true_div,
               const uint32_t* indices, const orc_vertex* verts, const orc_material* mats,
               const orc_texture_id* tex, const orc_uint4* atlas, uint32_t atlas_elems,
               orc_ray* rays, orc_ray* old_rays, const uint32_t* ray_pixels,
               orc_float4* scattered, orc_float4* emitted, const orc_aabb* bvh,
               uint64_t& nodes, uint64_t& tris, uint64_t& hits, uint64_t& max_stack) {
    orc_ray cur = rays[tid];
    orc_ray old = old_rays[tid];
    Hit best = {cur.tmax, 0.0f, 0.0f, 0.0f};
    TriIdx bi = {0, 0, 0};
    f3 inv_d = mk(1.0f / cur.dx, 1.0f / cur.dy, 1.0f / cur.dz);

    if (!use_bvh) {
        // shader/CS_TraceRays.hlsl:160-175
        for (uint32_t i = 0; i < info->num_indices; i += 3) {
            check_intersection(cur, i, indices, verts, best, bi);
            tris++;
        }
    } else {
        // shader/CS_TraceRays.hlsl:179-237. The reference's private stack has 48 entries, which
        // the balanced tree outgrows beyond ~2^24 leaves; 64 entries cover every 32-bit count.
        uint32_t stack[64];
        uint32_t n = 0;
        const orc_aabb& trunk = bvh[0];
        nodes++;
        if (intersect_aabb(cur, trunk, true_div, inv_d) != 1e30f) {
            stack[0] = trunk.link[1];
            stack[1] = trunk.link[0];
            n = 2;
        }
        while (n > 0) {
            if (n > max_stack) max_stack = n;
            const orc_aabb& box = bvh[stack[n - 1]];
            nodes++;
            float bt = intersect_aabb(cur, box, true_div, inv_d);
            bool remove = false;
            if ((bt != 1e30f) && (bt < best.t)) {
                stack[n - 1] |= 0x80000000u;
                if (box.link[0] & 0x80000000u) {
                    check_intersection(cur, box.link[0] & 0x7fffffffu, indices, verts, best, bi);
                    tris++;
                    if (box.link[1] != 0xffffffffu) {
                        check_intersection(cur, box.link[1] & 0x7fffffffu, indices, verts, best, bi);
                        tris++;
                    }
                    remove = true;
                } else {
                    if (box.link[1] == 0xffffffffu) {
                        stack[n] = box.link[0];
                        n++;
                    } else {
                        stack[n] = box.link[1];
                        stack[n + 1] = box.link[0];
                        n += 2;
                    }
                }
            } else {
                remove = true;
            }
            if (remove) {
                n--;
                while ((n > 0) && (stack[n - 1] & 0x80000000u)) n--;
            }
        }
    }

    // shader/CS_TraceRays.hlsl:116-126, 241-246
    uint32_t pv = ray_pixels[tid];
    uint32_t px = pv >> 16, py = pv & 0xffffu;
    uint32_t off = asuint(old.tmax);
    uint32_t idx = ((uint32_t)info->screen[0] * py + px) * info->max_rays_per_pixel + off;
    orc_float4 S = scattered[idx];
    orc_float4 E = emitted[idx];

    if (best.t != cur.tmax) {
        hits++;
        const orc_vertex& A = verts[bi.a];
        const orc_vertex& B = verts[bi.b];
        const orc_vertex& C = verts[bi.c];
        uint32_t s0 = tid * (tid + 17u);
        xorshift(s0);
        uint32_t sx = s0 + info->rng_seed[0], sy = s0 + info->rng_seed[1], sz = s0 + info->rng_seed[2];

        float u = best.u, v = best.v;
        float tu = interp(A.u, B.u, C.u, u, v), tv = interp(A.v, B.v, C.v, u, v);
        f3 N = mk(interp(A.nx, B.nx, C.nx, u, v), interp(A.ny, B.ny, C.ny, u, v), interp(A.nz, B.nz, C.nz, u, v));
        // (the interpolated tangent is computed by the reference but never consumed)

        // shader/Random.hlsli:100-126
        f3 p1 = normalize(mk(-N.z, 0.0f, N.x));
        if (N.z == 0.0f && N.x == 0.0f) p1 = mk(1.0f, 0.0f, 0.0f);
        f3 p2 = normalize(cross(N, p1));
        p1 = normalize(cross(N, p2));
        float r1 = rnd(sx), r2 = rnd(sy);
        (void)rnd(sz);
        float hy = sqrtf(1.0f - r2);
        float hr = sqrtf(r2);
        float phi = 6.2831853f * r1;
        float sn, cs;
        sincos_poly(phi, &sn, &cs);
        f3 base = mk(cs * hr, hy, sn * hr);
        f3 L = ((base.x * p1) + (base.y * N)) + (base.z * p2);

        if (dot(mk(old.dx, old.dy, old.dz), mk(old.dx, old.dy, old.dz)) == 0.0f) {
            S.x = 1.0f; S.y = 1.0f; S.z = 1.0f;
            E.x = 0.0f; E.y = 0.0f; E.z = 0.0f;
        }

        // shader/PerRayShading.hlsli:138-191
        f3 V = mk(-cur.dx, -cur.dy, -cur.dz);
        f3 H = normalize(V + L);
        orc_material m = mats[A.mat];
        f3 ta = sample_texture(tex[m.tex_albedo], tu, tv, atlas, atlas_elems);
        f3 tr = sample_texture(tex[m.tex_roughness], tu, tv, atlas, atlas_elems);
        f3 tf = sample_texture(tex[m.tex_f0], tu, tv, atlas, atlas_elems);
        f3 tm = sample_texture(tex[m.tex_metallic], tu, tv, atlas, atlas_elems);
        f3 te = sample_texture(tex[m.tex_emissive], tu, tv, atlas, atlas_elems);
        f3 albedo = mk(m.albedo[0], m.albedo[1], m.albedo[2]) * ta;
        float rough = m.roughness * tr.x;
        f3 f0 = mk(m.f0[0], m.f0[1], m.f0[2]) * tf;
        float metal = m.metallic * tm.x;
        f3 emis = mk(m.emissive[0], m.emissive[1], m.emissive[2]) * te;

        float vdh = saturate(dot(V, H)), ndh = saturate(dot(N, H)), ndv = saturate(dot(N, V));
        float ndl = saturate(dot(N, L)), ldv = saturate(dot(L, V));
        f3 Fr = reflected_color(rough, f0, ndl, ndv, ndh, vdh);
        f3 Fd = refracted_color(rough, metal, albedo, ndl, ndv, vdh, ldv);
        f3 Sc = mk(S.x, S.y, S.z), Em = mk(E.x, E.y, E.z);
        f3 k = (Fr + Fd) * ndl;
        f3 newS = Sc * mk(hmax(0.0f, k.x), hmax(0.0f, k.y), hmax(0.0f, k.z));
        f3 se = Sc * emis;
        f3 newE = mk(hmax(0.0f, se.x), hmax(0.0f, se.y), hmax(0.0f, se.z)) + Em;
        S.x = newS.x; S.y = newS.y; S.z = newS.z;
        E.x = newE.x; E.y = newE.y; E.z = newE.z;

        orc_ray nr;
        nr.dx = L.x; nr.dy = L.y; nr.dz = L.z;
        nr.ox = cur.ox + cur.dx * best.t;
        nr.oy = cur.oy + cur.dy * best.t;
        nr.oz = cur.oz + cur.dz * best.t;
        nr.tmin = cur.tmin;
        nr.tmax = cur.tmax;
        cur.tmax = old.tmax;
        rays[tid] = nr;
        old_rays[tid] = cur;
    }
    scattered[idx] = S;
    emitted[idx] = E;
}

// ---------------------------------------------------- small matrix helpers ----
// Row-vector, left-handed conventions of the math library the reference host uses.

struct M4 { float m[4][4]; };

M4 transpose(const M4& a) { M4 r; for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.m[i][j] = a.m[j][i]; return r; }

// general 4x4 inverse by cofactors, evaluated in binary32
M4 inverse(const M4& a) {
    const float* m = &a.m[0][0];
    float inv[16];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    float id = 1.0f / det;
    M4 r;
    for (int i = 0; i < 16; i++) (&r.m[0][0])[i] = inv[i] * id;
    return r;
}

// left-handed perspective, row-vector convention; half-angle sine/cosine from the same
// minimax polynomials the reference's math library uses for scalar sin/cos
M4 perspective_fov_lh(float fov, float aspect, float zn, float zf) {
    float s, c;
    sincos_poly(0.5f * fov, &s, &c);
    float h = c / s, w = h / aspect, range = zf / (zf - zn);
    M4 r; std::memset(&r, 0, sizeof r);
    r.m[0][0] = w; r.m[1][1] = h; r.m[2][2] = range; r.m[2][3] = 1.0f; r.m[3][2] = -range * zn;
    return r;
}

f3 normalize_or_zero(f3 v) {
    float l = sqrtf(dot(v, v));
    if (l == 0.0f) return mk(0.0f, 0.0f, 0.0f);
    if (std::isinf(l)) { float q = asfloat(0x7fc00000u); return mk(q, q, q); }
    return v / l;
}

M4 look_at_lh(f3 eye, f3 focus, f3 up) {
    f3 r2 = normalize_or_zero(focus - eye);
    f3 r0 = normalize_or_zero(cross(up, r2));
    f3 r1 = cross(r2, r0);
    f3 ne = mk(-eye.x, -eye.y, -eye.z);
    float d0 = dot(r0, ne), d1 = dot(r1, ne), d2 = dot(r2, ne);
    M4 r;
    r.m[0][0] = r0.x; r.m[0][1] = r1.x; r.m[0][2] = r2.x; r.m[0][3] = 0.0f;
    r.m[1][0] = r0.y; r.m[1][1] = r1.y; r.m[1][2] = r2.y; r.m[1][3] = 0.0f;
    r.m[2][0] = r0.z; r.m[2][1] = r1.z; r.m[2][2] = r2.z; r.m[2][3] = 0.0f;
    r.m[3][0] = d0; r.m[3][1] = d1; r.m[3][2] = d2; r.m[3][3] = 1.0f;
    return r;
}

inline float dot4(const float a[4], const float b[4]) { return ((a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]) + a[3] * b[3]; }

}  // namespace

extern "C" {

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

uint32_t orc_xorshift(uint32_t seed) { return xorshift(seed); }
uint32_t orc_thread_seed(uint32_t t) { uint32_t s = t * (t + 17u); xorshift(s); return s; }
uint32_t orc_left_shift3(uint32_t v) { return left_shift3(v); }
uint32_t orc_morton(uint32_t x, uint32_t y, uint32_t z) { return (left_shift3(z) << 2) | (left_shift3(y) << 1) | left_shift3(x); }
float orc_random_float(uint32_t* seed) { return rnd(*seed); }
void orc_sincos(float x, float* s, float* c) { sincos_poly(x, s, c); }

// shader/CS_GenerateMortonCodes.hlsl:42-77
void orc_generate_morton_codes(const orc_morton_info* info, const uint32_t* indices, const orc_vertex* verts,
                               orc_uint4* morton, orc_uint4* freq) {
    const uint32_t n = info->num_primitives;
    const uint32_t threads = ((n + 255u) / 256u) * 256u;
    f3 smin = mk(info->scene_min[0], info->scene_min[1], info->scene_min[2]);
    f3 smax = mk(info->scene_max[0], info->scene_max[1], info->scene_max[2]);
#pragma omp parallel for schedule(static)
    for (int64_t ti = 0; ti < (int64_t)threads; ti++) {
        uint32_t t = (uint32_t)ti;
        if (t < n) {
            uint32_t ci = t * 3u;
            f3 p1 = pos(verts[indices[ci]]), p2 = pos(verts[indices[ci + 1]]), p3 = pos(verts[indices[ci + 2]]);
            f3 c = 0.333333f * ((p1 + p2) + p3);
            f3 q = saturate((c - smin) / (smax - smin)) * 1023.0f;
            uint32_t code = (left_shift3(f2u(q.z)) << 2) | (left_shift3(f2u(q.y)) << 1) | left_shift3(f2u(q.x));
            orc_uint4 o = {code, ci, 0u, 0u};
            morton[t] = o;
        }
        if (t < 256u) freq[t].x = 0u;
    }
}

// shader/CS_SortCount.hlsl:21-42
void orc_sort_count(const orc_sort_info* info, const orc_uint4* morton, orc_uint4* temp, orc_uint4* freq) {
    const uint32_t n = info->num_elements, shift = 8u * info->pass_index;
    uint32_t total[256] = {0};
#pragma omp parallel
    {
        uint32_t local[256] = {0};
#pragma omp for schedule(static) nowait
        for (int64_t i = 0; i < (int64_t)n; i++) {
            temp[i].x = morton[i].x;
            temp[i].y = morton[i].y;
            local[(morton[i].x >> shift) & 0xffu]++;
        }
#pragma omp critical
        for (int d = 0; d < 256; d++) total[d] += local[d];
    }
    for (int d = 0; d < 256; d++) freq[d].x += total[d];
}

// shader/CS_SortPrefixSum.hlsl:20-56 (Hillis-Steele over 256 counters; the comparison is
// "thread > 2^i", not ">=", exactly as written)
void orc_sort_prefix_sum(const orc_sort_info*, orc_uint4* freq) {
    uint32_t buf[256];
    buf[0] = 0u;
    for (uint32_t t = 1; t < 256u; t++) buf[t] = freq[t - 1].x;
    for (uint32_t i = 0; i < 8u; i++) {
        uint32_t e = 1u << i;
        uint32_t cur[256];
        for (uint32_t t = 0; t < 256u; t++) cur[t] = (t > e) ? buf[t - e] : 0u;
        for (uint32_t t = 0; t < 256u; t++) if (t > e) buf[t] += cur[t];
    }
    for (uint32_t t = 0; t < 256u; t++) freq[t].y = buf[t];
}

// shader/CS_SortSort.hlsl:18-35 (thread = digit, each scanning all elements in order)
void orc_sort_sort(const orc_sort_info* info, orc_uint4* morton, const orc_uint4* temp, orc_uint4* freq) {
    const uint32_t n = info->num_elements, shift = 8u * info->pass_index;
#pragma omp parallel for schedule(dynamic, 1)
    for (int d = 0; d < 256; d++) {
        uint32_t next = freq[d].y;
        for (uint32_t i = 0; i < n; i++) {
            uint32_t key = temp[i].x;
            if (((key >> shift) & 0xffu) == (uint32_t)d) {
                morton[next].x = key;
                morton[next].y = temp[i].y;
                next++;
            }
        }
        freq[d].x = 0u;
    }
}

// shader/CS_BVHBuildLeaves.hlsl:28-63
void orc_bvh_build_leaves(const orc_bvh_info* info, const uint32_t* indices, const orc_vertex* verts,
                          const orc_uint4* morton, orc_aabb* bvh) {
    const uint32_t leaves = (info->num_children + 1u) / 2u;
#pragma omp parallel for schedule(static)
    for (int64_t ti = 0; ti < (int64_t)leaves; ti++) {
        uint32_t t = (uint32_t)ti;
        uint32_t ci[2];
        ci[0] = morton[2u * t].y;
        ci[1] = 0xffffffffu;
        uint32_t iters = 3;
        if ((2u * t + 1u) < info->num_children) { iters = 6; ci[1] = morton[2u * t + 1u].y; }
        f3 mn = mk(1e30f, 1e30f, 1e30f), mx = mk(-1e30f, -1e30f, -1e30f);
        for (uint32_t i = 0; i < iters; i++) {
            f3 p = pos(verts[indices[ci[i / 3u] + (i % 3u)]]);
            mn = vmin(mn, p);
            mx = vmax(mx, p);
        }
        orc_aabb o;
        o.mn[0] = mn.x; o.mn[1] = mn.y; o.mn[2] = mn.z;
        o.mx[0] = mx.x; o.mx[1] = mx.y; o.mx[2] = mx.z;
        o.link[0] = ci[0] | 0x80000000u;
        o.link[1] = ci[1] | 0x80000000u;
        bvh[t + 1u] = o;
    }
}

// shader/CS_BVHBuild.hlsl:25-48
void orc_bvh_build(const orc_bvh_info* info, orc_aabb* bvh) {
    const uint32_t parents = (info->num_children + 1u) / 2u;
#pragma omp parallel for schedule(static)
    for (int64_t ti = 0; ti < (int64_t)parents; ti++) {
        uint32_t t = (uint32_t)ti;
        uint32_t i1 = 2u * t + info->previous_index, i2 = i1 + 1u;
        orc_aabb a = bvh[i1], b = a;
        if ((i2 - info->previous_index) < info->num_children) b = bvh[i2];
        else i2 = 0xffffffffu;
        for (int k = 0; k < 3; k++) { a.mn[k] = hmin(a.mn[k], b.mn[k]); a.mx[k] = hmax(a.mx[k], b.mx[k]); }
        a.link[0] = i1;
        a.link[1] = i2;
        bvh[info->current_index + t] = a;
    }
}

// shader/CS_CameraRayGeneration.hlsl:56-101
void orc_camera_ray_generation(const orc_raygen_info* info, orc_ray* rays, orc_ray* old_rays, uint32_t* ray_pixels) {
    const int W = info->screen[0], Hh = info->screen[1];
    // constant-buffer matrices are read column-major: M[r][c] = mem[4c + r]
    float IV[4][4], IP[4][4], IVP[4][4];
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) {
        IV[r][c] = info->inverse_view[4 * c + r];
        IP[r][c] = info->inverse_projection[4 * c + r];
    }
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
        float col[4] = {IV[0][j], IV[1][j], IV[2][j], IV[3][j]};
        IVP[i][j] = dot4(IP[i], col);
    }
    const float unit_w[4] = {0.0f, 0.0f, 0.0f, 1.0f};
    f3 cam_add = mk(dot4(IV[0], unit_w), dot4(IV[1], unit_w), dot4(IV[2], unit_w));
    const float isx = 1.0f / (float)W, isy = 1.0f / (float)Hh;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < Hh; y++) {
        for (int x = 0; x < W; x++) {
            uint32_t s0 = (uint32_t)x * ((uint32_t)x + 17u);
            xorshift(s0);
            uint32_t sx = s0 + info->rng_seed[0], sy = s0 + info->rng_seed[1];
            for (uint32_t i = 0; i < info->max_rays_per_pixel; i++) {
                float scx = (float)x * isx, scy = (float)y * isy;
                float ndx = -2.0f * scx + 1.0f, ndy = -2.0f * scy + 1.0f;
                float r1x = rnd(sx), r1y = rnd(sy);
                float nnx = ndx + ((r1x - 0.5f) * isx) * info->dof_spread;
                float nny = ndy + ((r1y - 0.5f) * isy) * info->dof_spread;
                float r2x = rnd(sx), r2y = rnd(sy);
                float fnx = ndx + ((r2x - 0.5f) * isx) * info->aa_spread;
                float fny = ndy + ((r2y - 0.5f) * isy) * info->aa_spread;
                float nv[4] = {nnx, nny, 0.0f, 1.0f}, fv[4] = {fnx, fny, 1.0f, 1.0f};
                float np[4], fp[4];
                for (int j = 0; j < 4; j++) {
                    float col[4] = {IVP[0][j], IVP[1][j], IVP[2][j], IVP[3][j]};
                    np[j] = dot4(nv, col);
                    fp[j] = dot4(fv, col);
                }
                f3 nearp = mk(np[0] / np[3], np[1] / np[3], np[2] / np[3]);
                f3 farp = mk(fp[0] / fp[3], fp[1] / fp[3], fp[2] / fp[3]);
                f3 dir = normalize(farp - nearp);
                f3 org = nearp + cam_add;
                orc_ray r;
                r.ox = org.x; r.oy = org.y; r.oz = org.z;
                r.dx = dir.x; r.dy = dir.y; r.dz = dir.z;
                r.tmin = 0.0f;
                r.tmax = sqrtf(dot(farp, farp));
                orc_ray o;
                o.ox = o.oy = o.oz = o.dx = o.dy = o.dz = 0.0f;
                o.tmin = 0.0f;
                o.tmax = asfloat(i);
                uint32_t fi = ((uint32_t)y * (uint32_t)W + (uint32_t)x) * info->max_rays_per_pixel + i;
                rays[fi] = r;
                old_rays[fi] = o;
                ray_pixels[fi] = ((uint32_t)x << 16) | ((uint32_t)y & 0xffffu);
            }
        }
    }
}

// shader/CS_TraceRays.hlsl:148-293
void orc_trace_rays(const orc_trace_info* info, int use_bvh, int true_division,
                    const uint32_t* indices, const orc_vertex* verts,
                    const orc_material* mats, const orc_texture_id* tex,
                    const orc_uint4* atlas, uint32_t atlas_elems,
                    orc_ray* rays, orc_ray* old_rays, const uint32_t* ray_pixels,
                    orc_float4* scattered, orc_float4* emitted, const orc_aabb* bvh,
                    orc_trace_stats* stats) {
    uint64_t nodes = 0, tris = 0, hits = 0, max_stack = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : nodes, tris, hits) reduction(max : max_stack)
    for (int64_t t = 0; t < (int64_t)info->num_rays; t++) {
        trace_one((uint32_t)t, info, use_bvh, true_division != 0, indices, verts, mats, tex, atlas, atlas_elems,
                  rays, old_rays, ray_pixels, scattered, emitted, bvh, nodes, tris, hits, max_stack);
    }
    if (stats) {
        stats->nodes_visited = nodes; stats->tris_tested = tris; stats->rays_hit = hits;
        stats->rays = info->num_rays; stats->max_stack = max_stack;
    }
}

// shader/CS_GenerateFinalImage.hlsl:28-60
void orc_generate_final_image(const orc_image_info* info, const orc_float4* emitted, orc_float4* result,
                              orc_float4* output) {
    const int W = info->screen[0], Hh = info->screen[1];
    const uint32_t rpp = info->max_rays_per_pixel;
    const uint32_t ns = info->num_samples & 0x7fffffffu;
    const float inv = 1.0f / (float)ns;
    const float keep = (float)(ns - 1u) * inv;
    const bool apply = (info->num_samples & 0x80000000u) != 0u;
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < (int64_t)W * Hh; p++) {
        orc_float4 total = result[p];
        float nc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (uint32_t i = 0; i < rpp; i++) {
            const orc_float4& e = emitted[(uint64_t)p * rpp + i];
            nc[0] += e.x; nc[1] += e.y; nc[2] += e.z; nc[3] += e.w;
        }
        float tc[4] = {total.x, total.y, total.z, total.w}, dc[4], oc[4];
        for (int k = 0; k < 4; k++) {
            nc[k] = nc[k] / (float)rpp;
            dc[k] = keep * tc[k] + inv * nc[k];
            oc[k] = dc[k] / (1.0f + dc[k]);
        }
        if (apply) { total.x = dc[0]; total.y = dc[1]; total.z = dc[2]; total.w = dc[3]; }
        orc_float4 o = {oc[0], oc[1], oc[2], oc[3]};
        output[p] = o;
        result[p] = total;
    }
}

// ------------------------------------------------------------- host logic ----

// src/RaytracerPipeline.cpp:276-383
void orc_sort_primitives(const float scene_min[3], const float scene_max[3], uint32_t n,
                         const uint32_t* indices, const orc_vertex* verts,
                         orc_uint4* morton, orc_uint4* temp, orc_uint4* freq, int as_written) {
    orc_morton_info mi;
    std::memset(&mi, 0, sizeof mi);
    for (int k = 0; k < 3; k++) { mi.scene_min[k] = scene_min[k]; mi.scene_max[k] = scene_max[k]; }
    mi.num_primitives = n;
    orc_generate_morton_codes(&mi, indices, verts, morton, freq);
    for (uint32_t p = 0; p < 4u; p++) {
        orc_sort_info si = {n, p};
        if (as_written) {
            orc_sort_count(&si, morton, temp, freq);
            orc_sort_prefix_sum(&si, freq);
            orc_sort_sort(&si, morton, temp, freq);
        } else {
            const uint32_t shift = 8u * p;
            std::stable_sort(morton, morton + n, [shift](const orc_uint4& a, const orc_uint4& b) {
                return ((a.x >> shift) & 0xffu) < ((b.x >> shift) & 0xffu);
            });
        }
    }
}

uint32_t orc_bvh_node_count(uint32_t n) {
    if (n < 3u) return 0u;
    uint32_t c = (n + 1u) / 2u, total = 1u + c;
    while (c != 2u) { c = (c + 1u) / 2u; total += c; }
    return total;
}

// src/RaytracerPipeline.cpp:465-467, 475-562
uint32_t orc_build_bvh(uint32_t n, const uint32_t* indices, const orc_vertex* verts,
                       const orc_uint4* morton, orc_aabb* bvh) {
    if (n < 3u) return 0u;
    orc_bvh_info bi;
    bi.num_children = n;
    bi.previous_index = 1u;
    bi.current_index = ((n + 1u) / 2u) + 1u;
    bi.pad = 0u;
    orc_bvh_build_leaves(&bi, indices, verts, morton, bvh);
    bi.num_children = (bi.num_children + 1u) / 2u;
    while (bi.num_children != 2u) {
        orc_bvh_info snap = bi;
        bi.num_children = (bi.num_children + 1u) / 2u;
        orc_bvh_build(&snap, bvh);
        bi.previous_index = bi.current_index;
        bi.current_index += bi.num_children;
    }
    uint32_t total = bi.current_index;
    bi.num_children = 2u;
    bi.current_index = 0u;
    orc_bvh_build(&bi, bvh);
    return total;
}

// src/RaytracerPipeline.cpp:81-89
void orc_camera_matrices(const float position[3], const float focus[3], const float up[3],
                         float fov, float aspect, float near_z, float far_z,
                         float inverse_view[16], float inverse_projection[16]) {
    M4 ip = inverse(transpose(perspective_fov_lh(fov, aspect, near_z, far_z)));
    M4 iv = inverse(transpose(look_at_lh(mk(position[0], position[1], position[2]), mk(focus[0], focus[1], focus[2]),
                                         mk(up[0], up[1], up[2]))));
    std::memcpy(inverse_view, &iv.m[0][0], 64);
    std::memcpy(inverse_projection, &ip.m[0][0], 64);
}

// src/RaytracerMesh.cpp:342-364
void orc_scene_aabb(uint64_t index_count, const uint32_t* indices, const orc_vertex* verts,
                    float scene_min[3], float scene_max[3]) {
    const float third = 1.0f / 3.0f;
    f3 mn = mk(3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f);
    f3 mx = mk(-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f);
    for (uint64_t i = 0; i + 2 < index_count; i += 3) {
        f3 c = ((pos(verts[indices[i]]) + pos(verts[indices[i + 1]])) + pos(verts[indices[i + 2]])) * third;
        // the SSE min/max of the reference return the second operand when either is NaN
        mn = mk(mn.x < c.x ? mn.x : c.x, mn.y < c.y ? mn.y : c.y, mn.z < c.z ? mn.z : c.z);
        mx = mk(mx.x > c.x ? mx.x : c.x, mx.y > c.y ? mx.y : c.y, mx.z > c.z ? mx.z : c.z);
    }
    scene_min[0] = mn.x; scene_min[1] = mn.y; scene_min[2] = mn.z;
    scene_max[0] = mx.x; scene_max[1] = mx.y; scene_max[2] = mx.z;
}

// The synthetic dynamic workload (not part of the reference, whose scenes are static): the CPU twin of the
// product's animation kernel. Triangle i (vertices 3i .. 3i + 2) moves as a whole by
// d = 0.25 * sin(2 pi * frac(frame / 32 + 0.618 * i)) along each axis.
void orc_synthetic_animate(uint32_t num_primitives, uint32_t frame, const orc_vertex* base, orc_vertex* out) {
#pragma omp parallel for schedule(static)
    for (int64_t ti = 0; ti < (int64_t)num_primitives; ti++) {
        const uint32_t t = (uint32_t)ti;
        float ph = (float)frame * 0.03125f + 0.618f * (float)t;
        ph = ph - std::floor(ph);
        float s, c;
        sincos_poly(6.2831853f * ph, &s, &c);
        const float d = 0.25f * s;
        for (int k = 0; k < 3; k++) {
            orc_vertex v = base[3u * t + k];
            v.px += d; v.py += d; v.pz += d;
            out[3u * t + k] = v;
        }
    }
}

// src/RaytracerMesh.cpp:99-155
void orc_calculate_tangents(uint64_t index_count, const uint32_t* indices, orc_vertex* verts) {
    std::vector<f3> acc(index_count, mk(0.0f, 0.0f, 0.0f));
    const uint64_t tris = index_count / 3;
    for (uint64_t i = 0; i < tris; i++) {
        uint32_t i1 = indices[i * 3], i2 = indices[i * 3 + 1], i3 = indices[i * 3 + 2];
        f3 e1 = pos(verts[i2]) - pos(verts[i1]), e2 = pos(verts[i3]) - pos(verts[i1]);
        float t1x = verts[i2].u - verts[i1].u, t1y = verts[i2].v - verts[i1].v;
        float t2x = verts[i3].u - verts[i1].u, t2y = verts[i3].v - verts[i1].v;
        float den = t1x * t2y - t1y * t2x;
        f3 nom = (t2y * e1) - (t1y * e2);
        const float eps = 1.192092896e-7f;
        f3 t = (den <= eps && den >= -eps) ? mk(0.0f, 0.0f, 0.0f) : nom / den;
        acc[i1] = acc[i1] + t; acc[i2] = acc[i2] + t; acc[i3] = acc[i3] + t;
    }
    for (uint64_t i = 0; i < index_count; i++) {
        uint32_t c = indices[i];
        f3 t = normalize_or_zero(acc[c]);
        verts[c].tx = t.x; verts[c].ty = t.y; verts[c].tz = t.z;
    }
}

// src/RaytracerMesh.cpp:59-97
void orc_calculate_normals(uint64_t index_count, const uint32_t* indices, orc_vertex* verts) {
    std::vector<f3> acc(index_count, mk(0.0f, 0.0f, 0.0f));
    const uint64_t tris = index_count / 3;
    for (uint64_t i = 0; i < tris; i++) {
        uint32_t i1 = indices[i * 3], i2 = indices[i * 3 + 1], i3 = indices[i * 3 + 2];
        f3 e1 = pos(verts[i3]) - pos(verts[i1]), e2 = pos(verts[i2]) - pos(verts[i1]);
        f3 n = cross(e2, e1);
        acc[i1] = acc[i1] + n; acc[i2] = acc[i2] + n; acc[i3] = acc[i3] + n;
    }
    for (uint64_t i = 0; i < index_count; i++) {
        uint32_t c = indices[i];
        f3 n = normalize_or_zero(acc[c]);
        verts[c].nx = n.x; verts[c].ny = n.y; verts[c].nz = n.z;
    }
}

void orc_mt19937_draw(uint32_t seed, uint32_t skip, uint32_t count, uint32_t* out) {
    std::mt19937 g(seed);
    for (uint32_t i = 0; i < skip; i++) (void)g();
    for (uint32_t i = 0; i < count; i++) out[i] = (uint32_t)g();
}

// src/RaytracerMesh.cpp:290-299
float orc_roughness_from_mtl(float shininess, float pr) {
    if (shininess == 1.0f) return pr;
    return 1.0f - (log2f(fminf(fmaxf(shininess, 1.0f), 1448.15f)) / 10.5f);
}

// --------------------------------------------------------------- pipeline ----

struct orc_pipeline {
    orc_settings s;
    std::vector<uint32_t> indices;
    std::vector<orc_vertex> vertices;
    std::vector<orc_material> materials;
    std::vector<orc_texture_id> tex_ids;
    std::vector<orc_uint4> atlas;
    float scene_min[3], scene_max[3];
    uint32_t num_prims, max_rays;
    std::vector<orc_ray> rays, old_rays;
    std::vector<uint32_t> ray_pixels;
    std::vector<orc_float4> scattered, emitted, result, output;
    std::vector<orc_uint4> morton, temp, freq;
    std::vector<orc_aabb> bvh;
    orc_raygen_info rg;
    orc_trace_info tr;
    orc_image_info im;
    std::mt19937 prng_raygen, prng_trace;
    bool built;
    uint32_t iteration;
    orc_trace_stats last, total;
};

// src/RaytracerPipeline.cpp:888-935 and the stage Initialize() functions
orc_pipeline* orc_pipeline_create(const orc_settings* s, uint64_t index_count, const uint32_t* indices,
                                  uint64_t vertex_count, const orc_vertex* vertices,
                                  uint64_t material_count, const orc_material* materials,
                                  uint32_t texture_count, const orc_texture_id* texture_ids,
                                  const orc_uint4* atlas, uint32_t atlas_elements,
                                  const float scene_min[3], const float scene_max[3]) {
    orc_pipeline* p = new orc_pipeline();
    p->s = *s;
    p->indices.assign(indices, indices + index_count);
    p->vertices.assign(vertices, vertices + vertex_count);
    p->materials.assign(materials, materials + material_count);
    p->tex_ids.assign(texture_ids, texture_ids + texture_count);
    p->atlas.assign(atlas, atlas + atlas_elements);
    for (int k = 0; k < 3; k++) { p->scene_min[k] = scene_min[k]; p->scene_max[k] = scene_max[k]; }
    p->num_prims = (uint32_t)(index_count / 3);
    p->max_rays = (uint32_t)s->width * (uint32_t)s->height * s->max_rays_per_pixel;
    orc_ray zr; std::memset(&zr, 0, sizeof zr);
    orc_float4 zf = {0, 0, 0, 0};
    orc_uint4 zu = {0, 0, 0, 0};
    p->rays.assign(p->max_rays, zr);
    p->old_rays.assign(p->max_rays, zr);
    p->ray_pixels.assign(p->max_rays, 0u);
    p->scattered.assign(p->max_rays, zf);
    p->emitted.assign(p->max_rays, zf);
    p->result.assign((size_t)s->width * s->height, zf);
    p->output.assign((size_t)s->width * s->height, zf);
    p->morton.assign(p->num_prims, zu);
    p->temp.assign(p->num_prims, zu);
    p->freq.assign(256, zu);
    orc_aabb za; std::memset(&za, 0, sizeof za);
    p->bvh.assign((size_t)p->num_prims * 4u, za);
    std::memset(&p->rg, 0, sizeof p->rg);
    orc_camera_matrices(s->cam_position, s->cam_focus, s->cam_up, s->fov, (float)s->width / (float)s->height,
                        s->near_z, s->far_z, p->rg.inverse_view, p->rg.inverse_projection);
    p->rg.screen[0] = s->width; p->rg.screen[1] = s->height;
    p->rg.aa_spread = s->aa_spread; p->rg.dof_spread = s->dof_spread;
    p->rg.max_rays_per_pixel = s->max_rays_per_pixel;
    std::memset(&p->tr, 0, sizeof p->tr);
    p->tr.screen[0] = s->width; p->tr.screen[1] = s->height;
    p->tr.num_indices = (uint32_t)index_count;
    p->tr.num_rays = p->max_rays;
    p->tr.max_rays_per_pixel = s->max_rays_per_pixel;
    p->im.screen[0] = s->width; p->im.screen[1] = s->height;
    p->im.max_rays_per_pixel = s->max_rays_per_pixel;
    p->im.num_samples = 1u;
    p->prng_raygen.seed(s->seed_raygen);
    p->prng_trace.seed(s->seed_trace);
    p->built = false;
    p->iteration = 0;
    std::memset(&p->last, 0, sizeof p->last);
    std::memset(&p->total, 0, sizeof p->total);
    return p;
}

// src/RaytracerPipeline.cpp:939-991
int orc_pipeline_render(orc_pipeline* p) {
    if (p->s.use_bvh) {
        bool rebuild = !p->built || (p->s.rebuild_every_cycle && p->iteration == 0);
        if (rebuild) {
            // rebuild_every_cycle = 2: after the first build the sorted order is kept and only CS_BVHBuildLeaves and
            // CS_BVHBuild run again over the new vertices (a refit; the reference has no such mode)
            if (!p->built || p->s.rebuild_every_cycle != 2)
                orc_sort_primitives(p->scene_min, p->scene_max, p->num_prims, p->indices.data(), p->vertices.data(),
                                    p->morton.data(), p->temp.data(), p->freq.data(), 0);
            if (!orc_build_bvh(p->num_prims, p->indices.data(), p->vertices.data(), p->morton.data(), p->bvh.data()))
                return 0;
            p->built = true;
        }
    }
    if (p->iteration == 0) {
        p->rg.rng_seed[0] = (uint32_t)p->prng_raygen();
        p->rg.rng_seed[1] = (uint32_t)p->prng_raygen();
        p->rg.rng_seed[2] = (uint32_t)p->prng_raygen();
        orc_camera_ray_generation(&p->rg, p->rays.data(), p->old_rays.data(), p->ray_pixels.data());
    }
    p->iteration++;
    if (p->iteration == p->s.max_ray_depth) p->iteration = 0;

    p->tr.rng_seed[0] = (uint32_t)p->prng_trace();
    p->tr.rng_seed[1] = (uint32_t)p->prng_trace();
    p->tr.rng_seed[2] = (uint32_t)p->prng_trace();
    orc_trace_rays(&p->tr, p->s.use_bvh, p->s.true_division, p->indices.data(), p->vertices.data(),
                   p->materials.data(), p->tex_ids.data(), p->atlas.data(), (uint32_t)p->atlas.size(),
                   p->rays.data(), p->old_rays.data(), p->ray_pixels.data(), p->scattered.data(), p->emitted.data(),
                   p->bvh.data(), &p->last);
    p->total.nodes_visited += p->last.nodes_visited;
    p->total.tris_tested += p->last.tris_tested;
    p->total.rays_hit += p->last.rays_hit;
    p->total.rays += p->last.rays;
    if (p->last.max_stack > p->total.max_stack) p->total.max_stack = p->last.max_stack;

    // src/RaytracerPipeline.cpp:809-826
    if (p->iteration == 0) { p->im.num_samples |= 0x80000000u; p->im.num_samples++; }
    else p->im.num_samples &= 0x7fffffffu;
    orc_generate_final_image(&p->im, p->emitted.data(), p->result.data(), p->output.data());
    return 1;
}

// A mesh that moved (dynamic workloads; the reference never changes its mesh): new vertex data of the same
// count, scene bounds recomputed as the loader does. The hierarchy is rebuilt at the start of the next sample
// cycle when rebuild_every_cycle is set.
int orc_pipeline_set_vertices(orc_pipeline* p, uint64_t vertex_count, const orc_vertex* vertices) {
    if (vertex_count != p->vertices.size()) return 0;
    p->vertices.assign(vertices, vertices + vertex_count);
    orc_scene_aabb(p->indices.size(), p->indices.data(), p->vertices.data(), p->scene_min, p->scene_max);
    return 1;
}

const orc_float4* orc_pipeline_output(orc_pipeline* p) { return p->output.data(); }
const orc_float4* orc_pipeline_result(orc_pipeline* p) { return p->result.data(); }
const orc_aabb* orc_pipeline_bvh(orc_pipeline* p) { return p->bvh.data(); }
const orc_uint4* orc_pipeline_morton(orc_pipeline* p) { return p->morton.data(); }
const orc_ray* orc_pipeline_rays(orc_pipeline* p) { return p->rays.data(); }
const orc_float4* orc_pipeline_emitted(orc_pipeline* p) { return p->emitted.data(); }
const orc_float4* orc_pipeline_scattered(orc_pipeline* p) { return p->scattered.data(); }
void orc_pipeline_stats(orc_pipeline* p, orc_trace_stats* last, orc_trace_stats* total) {
    if (last) *last = p->last;
    if (total) *total = p->total;
}
void orc_pipeline_destroy(orc_pipeline* p) { delete p; }

}  // extern "C"
