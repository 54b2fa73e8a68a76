// Per-frame LBVH build for gfx950: Morton codes, stable LSD radix sort, leaf and level emission.
//
// Results are bit-identical to the reference's compute shaders
//   shader/CS_GenerateMortonCodes.hlsl, CS_SortCount.hlsl, CS_SortPrefixSum.hlsl, CS_SortSort.hlsl,
//   CS_BVHBuildLeaves.hlsl, CS_BVHBuild.hlsl
// as sequenced by src/RaytracerPipeline.cpp:276-383 and 475-562, but the work decomposition is
// new: the reference scatters with ONE workgroup whose 256 threads each scan all N keys; here
// every pass is tile-parallel (per-tile digit counts -> per-digit scan over tiles -> ranked scatter
// with wave64 ballots and an LDS-staged reorder), and the ~log2(N) per-level dispatches are fused
// eight levels at a time through LDS.
//
// Built with -ffp-contract=off: the Morton quantisation must round once per written operation.

#include <atomic>
#include <new>

#include "rt_internal.h"
#include "rt_math.h"

// Scratch of rt_scene_bounds_static, the entry point that allocates nothing: the per-workgroup partial bounds of the
// centroid reduction (k_bounds_partial -> k_bounds_finish), RT_BOUNDS_BLOCKS x 6 floats = 24 KiB, resident on every GPU
// the library runs on until the process ends. In host code the array's name is only a stand-in for the symbol; its
// address on a device comes from hipGetSymbolAddress with that device current (rt_static_scratch_init, called by
// rt_device_create) and is what the kernels are given.
#define RT_BOUNDS_BLOCKS 1024u
static __device__ __align__(256) float scratch[RT_BOUNDS_BLOCKS * 6u];
constexpr int kStaticScratchDevices = 64;
static std::atomic<float*> g_scratch_ptr[kStaticScratchDevices];

int rt_static_scratch_init(int gpu) {
    if (gpu < 0 || gpu >= kStaticScratchDevices) return RT_OK;      // such a device has no static scratch; its entry point says so
    if (g_scratch_ptr[gpu].load(std::memory_order_acquire)) return RT_OK;
    float* p = nullptr;
    RT_HIP(hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch)));   // the current device's copy of the array
    if (!p) return RT_ERR_HIP_BASE + (int)hipErrorInvalidSymbol;
    g_scratch_ptr[gpu].store(p, std::memory_order_release);
    return RT_OK;
}

namespace {

constexpr int kItems = RT_SORT_ITEMS;
constexpr int kTile = RT_SORT_TILE;

__device__ __forceinline__ uint32_t left_shift3(uint32_t v) {
    // shader/CS_GenerateMortonCodes.hlsl:30-38 -- the "|=" (rather than "= (v | v << s) & m") is the
    // reference's own and is kept: it leaves stray low bits in place.
    v |= (v << 16) & 0x030000ffu;
    v |= (v << 8) & 0x0300f00fu;
    v |= (v << 4) & 0x030c30c3u;
    v |= (v << 2) & 0x09249249u;
    return v;
}

__device__ __forceinline__ float saturatef(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

__device__ __forceinline__ uint32_t morton_code(float p1x, float p1y, float p1z, float p2x, float p2y, float p2z,
                                                 float p3x, float p3y, float p3z, const float* smin, const float* smax) {
    // shader/CS_GenerateMortonCodes.hlsl:58-66
    float cx = 0.333333f * ((p1x + p2x) + p3x);
    float cy = 0.333333f * ((p1y + p2y) + p3y);
    float cz = 0.333333f * ((p1z + p2z) + p3z);
    float qx = saturatef((cx - smin[0]) / (smax[0] - smin[0])) * 1023.0f;
    float qy = saturatef((cy - smin[1]) / (smax[1] - smin[1])) * 1023.0f;
    float qz = saturatef((cz - smin[2]) / (smax[2] - smin[2])) * 1023.0f;
    return (left_shift3((uint32_t)qz) << 2) | (left_shift3((uint32_t)qy) << 1) | left_shift3((uint32_t)qx);
}

struct SceneBounds { float mn[3]; float mx[3]; };

// reference layout: Indices -> Vertices (48-byte stride) -> uint4(code, 3*tri, 0, 0)
__global__ __launch_bounds__(256) void k_morton_ref(SceneBounds sb, uint32_t n, const uint32_t* __restrict__ indices,
                                                    const rt_vertex* __restrict__ verts, uint4* __restrict__ morton,
                                                    uint4* __restrict__ freq) {
    uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t < n) {
        uint32_t ci = t * 3u;
        const float* a = verts[indices[ci]].position;
        const float* b = verts[indices[ci + 1]].position;
        const float* c = verts[indices[ci + 2]].position;
        uint32_t code = morton_code(a[0], a[1], a[2], b[0], b[1], b[2], c[0], c[1], c[2], sb.mn, sb.mx);
        morton[t] = make_uint4(code, ci, 0u, 0u);
    }
    if (t < 256u) freq[t].x = 0u;
}

// gathers positions into 3 x float4 per triangle (original order) for 128-bit loads
__global__ __launch_bounds__(256) void k_pack(uint32_t n, const uint32_t* __restrict__ indices,
                                              const rt_vertex* __restrict__ verts, float4* __restrict__ tris) {
    uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n) return;
    uint32_t ci = t * 3u;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float* p = verts[indices[ci + k]].position;
        tris[ci + k] = make_float4(p[0], p[1], p[2], 0.0f);
    }
}

// Centroid bounds of the mesh as the reference's loader computes them on the host (src/RaytracerMesh.cpp:342-364):
// min / max over ((p1 + p2) + p3) * (1/3). min and max are exact, so the order of the reduction does not matter
// for finite positions. Stage one: every workgroup strides over the triangles and leaves its bounds in
// partials[block][6]; PACK also writes the packed triangle array (k_pack). Stage two reduces the partials.
constexpr uint32_t kBoundsBlocks = RT_BOUNDS_BLOCKS;

template <bool PACK>
__global__ __launch_bounds__(256) void k_bounds_partial(uint32_t n, const uint32_t* __restrict__ indices,
                                                        const rt_vertex* __restrict__ verts, float4* __restrict__ tris,
                                                        float* __restrict__ partials) {
    __shared__ float s_red[4][6];
    const float third = 1.0f / 3.0f;
    float mn[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
    float mx[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
    for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < n; t += gridDim.x * 256u) {
        const uint32_t ci = t * 3u;
        float p[3][3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float* q = verts[indices[ci + k]].position;
            p[k][0] = q[0]; p[k][1] = q[1]; p[k][2] = q[2];
            if (PACK) tris[ci + k] = make_float4(q[0], q[1], q[2], 0.0f);
        }
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float c = ((p[0][a] + p[1][a]) + p[2][a]) * third;
            mn[a] = fminf(mn[a], c);
            mx[a] = fmaxf(mx[a], c);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], d, 64));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], d, 64));
        }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0u)
        for (int a = 0; a < 3; a++) { s_red[wave][a] = mn[a]; s_red[wave][3 + a] = mx[a]; }
    __syncthreads();
    if (threadIdx.x < 6u) {
        const uint32_t a = threadIdx.x;
        float v = s_red[0][a];
        for (int w = 1; w < 4; w++) v = (a < 3u) ? fminf(v, s_red[w][a]) : fmaxf(v, s_red[w][a]);
        partials[blockIdx.x * 6u + a] = v;
    }
}

// bounds: (min.xyz, 0, max.xyz, 0), the head of the Morton-code constant buffer
__global__ __launch_bounds__(256) void k_bounds_finish(uint32_t blocks, const float* __restrict__ partials,
                                                       float* __restrict__ bounds) {
    __shared__ float s_red[4][6];
    float v[6] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
    for (uint32_t b = threadIdx.x; b < blocks; b += 256u)
#pragma unroll
        for (int a = 0; a < 6; a++) v[a] = (a < 3) ? fminf(v[a], partials[b * 6u + a]) : fmaxf(v[a], partials[b * 6u + a]);
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const float o = __shfl_xor(v[a], d, 64);
            v[a] = (a < 3) ? fminf(v[a], o) : fmaxf(v[a], o);
        }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0u)
        for (int a = 0; a < 6; a++) s_red[wave][a] = v[a];
    __syncthreads();
    if (threadIdx.x < 8u) {
        const uint32_t j = threadIdx.x;              // output slot: 0..2 min, 3 pad, 4..6 max, 7 pad
        float r = 0.0f;
        if ((j & 3u) != 3u) {
            const uint32_t a = (j < 4u) ? j : j - 1u;
            r = s_red[0][a];
            for (int w = 1; w < 4; w++) r = (a < 3u) ? fminf(r, s_red[w][a]) : fmaxf(r, s_red[w][a]);
        }
        bounds[j] = r;
    }
}

// The synthetic dynamic workload (no counterpart in the reference, whose scenes are static): triangle i of the
// base mesh (vertices 3i .. 3i + 2) is moved as a whole by d = 0.25 * sin(2 pi * frac(frame / 32 + 0.618 * i)) along
// each axis. Everything else of the vertex is copied.
// One thread per 16 bytes of the vertex array (a triangle = 144 bytes = nine of them; the first of a vertex's three holds the
// position and one texture coordinate): every wave reads and writes one contiguous kilobyte, which a thread per triangle
// copying 48-byte structs 144 bytes apart does not (measured at 50 M triangles: 4.4 ms, 3.3 TB/s).
__global__ __launch_bounds__(256) void k_animate(uint32_t n, uint32_t frame, const float4* __restrict__ base, float4* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= (uint64_t)n * 9u) return;
    const uint32_t t = (uint32_t)(i / 9u), part = (uint32_t)(i - (uint64_t)t * 9u);
    float4 v = base[i];
    if (part % 3u == 0u) {
        float ph = (float)frame * 0.03125f + 0.618f * (float)t;
        ph = ph - floorf(ph);
        float s, c;
        rtm::sincos_poly(6.2831853f * ph, &s, &c);
        const float d = 0.25f * s;
        v.x += d; v.y += d; v.z += d;
    }
    out[i] = v;
}

// Keys of the library's own hierarchy, and every triangle's place in the reference's order: entry r of the
// sorted reference pairs names triangle ci / 3; its key is the interleave of the same ten bits per axis the
// reference quantises the centroid to (CS_GenerateMortonCodes.hlsl:58-66), without the overlap of the reference's
// bit spreading.
// The keys' coordinates are not the reference's equal steps of the scene bounds but steps that hold equal numbers of
// centroids (per axis): a scene that fills a corner of its bounds - or whose bounds a few far triangles stretch -
// still uses all 30 bits, and its Morton cells hold similar numbers of triangles, which k_cell_choose needs.
__device__ __forceinline__ uint32_t axis_step(float c, float mn, float mx) {
    return (uint32_t)(saturatef((c - mn) / (mx - mn)) * (float)(RT_AXIS_STEPS - 1u));
}
// (a sample of the triangles is enough for steps of equal population: every `stride`-th triangle is counted)
__global__ __launch_bounds__(256) void k_axis_hist(SceneBounds sb, const float* __restrict__ dbounds, uint32_t n, uint32_t stride,
                                                   const float4* __restrict__ tris, uint32_t* __restrict__ cells) {
    __shared__ uint32_t s_hist[3u * RT_AXIS_STEPS];
    for (uint32_t i = threadIdx.x; i < 3u * RT_AXIS_STEPS; i += 256u) s_hist[i] = 0u;
    __syncthreads();
    if (dbounds) {
        sb.mn[0] = dbounds[0]; sb.mn[1] = dbounds[1]; sb.mn[2] = dbounds[2];
        sb.mx[0] = dbounds[4]; sb.mx[1] = dbounds[5]; sb.mx[2] = dbounds[6];
    }
    const uint32_t samples = (n + stride - 1u) / stride;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < samples; i += gridDim.x * 256u) {
        const uint32_t t = i * stride;
        const float4 a = tris[3u * t], b = tris[3u * t + 1u], c = tris[3u * t + 2u];
        const float cx = 0.333333f * ((a.x + b.x) + c.x), cy = 0.333333f * ((a.y + b.y) + c.y), cz = 0.333333f * ((a.z + b.z) + c.z);
        atomicAdd(&s_hist[axis_step(cx, sb.mn[0], sb.mx[0])], 1u);
        atomicAdd(&s_hist[RT_AXIS_STEPS + axis_step(cy, sb.mn[1], sb.mx[1])], 1u);
        atomicAdd(&s_hist[2u * RT_AXIS_STEPS + axis_step(cz, sb.mn[2], sb.mx[2])], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 3u * RT_AXIS_STEPS; i += 256u)
        if (s_hist[i]) atomicAdd(cells + RT_CELLS_AXIS + i, s_hist[i]);
}
// counts -> coordinates: step i of an axis becomes floor(1024 * (centroids in the steps before it) / n); three
// workgroups (one per axis) of 1024 threads, four steps each
__global__ __launch_bounds__(1024) void k_axis_map(uint32_t n, uint32_t* __restrict__ cells) {
    __shared__ uint32_t s_sum[1024];
    uint32_t* h = cells + RT_CELLS_AXIS + blockIdx.x * RT_AXIS_STEPS;
    const uint32_t t = threadIdx.x;
    const uint32_t v0 = h[4u * t], v1 = h[4u * t + 1u], v2 = h[4u * t + 2u], v3 = h[4u * t + 3u];
    s_sum[t] = v0 + v1 + v2 + v3;
    __syncthreads();
    for (uint32_t d = 1u; d < 1024u; d <<= 1) {            // inclusive scan of the per-thread sums
        const uint32_t add = t >= d ? s_sum[t - d] : 0u;
        __syncthreads();
        s_sum[t] += add;
        __syncthreads();
    }
    uint32_t before = s_sum[t] - (v0 + v1 + v2 + v3);
    const uint32_t c[4] = {v0, v1, v2, v3};
#pragma unroll
    for (int j = 0; j < 4; j++) {
        h[4u * t + j] = min(1023u, (uint32_t)(((unsigned long long)before << 10) / n));
        before += c[j];
    }
}
__device__ __forceinline__ uint32_t spread_bits3(uint32_t v) {
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
// key of the library's own hierarchy for a triangle with centroid (cx, cy, cz); map: k_axis_map's three tables
__device__ __forceinline__ uint32_t spatial_key(float cx, float cy, float cz, const SceneBounds& sb, const uint32_t* __restrict__ map) {
    const uint32_t qx = map[axis_step(cx, sb.mn[0], sb.mx[0])];
    const uint32_t qy = map[RT_AXIS_STEPS + axis_step(cy, sb.mn[1], sb.mx[1])];
    const uint32_t qz = map[2u * RT_AXIS_STEPS + axis_step(cz, sb.mn[2], sb.mx[2])];
    return (spread_bits3(qz) << 2) | (spread_bits3(qy) << 1) | spread_bits3(qx);
}

// ---- leaf slots of the library's own hierarchy by Morton cell ----
//
// A balanced tree over the sorted order has, on every level, nodes whose run of leaves crosses a major boundary
// of the Morton curve: the run ends in one octant and goes on in the next, and the node's box spans both. Rays meet
// those boxes on every level. With more leaf slots than leaves (rt_spatial_slots) the leaves of every Morton cell of
// 2^-T of the curve can start on a slot that is a multiple of slots / 2^T: then every node above that size IS a
// cell (or half, or a quarter of one), and only inside a cell are leaves laid out by rank. T is the largest of
// 21, 18, ... 3 whose fullest cell still fits its slots (two triangles per slot); 0 = by rank, as the reference.
// cells: see RT_CELLS_* in rt_internal.h; all of it is zero before k_cell_hist.
// (outliers - see RT_CELLS_OUTLIERS - carry the largest key, sort behind every other triangle and belong to no cell)
__device__ __forceinline__ uint32_t without_outliers(uint32_t n, const uint32_t* __restrict__ cells) {
    return n - min(min(cells[RT_CELLS_OUTLIERS], RT_OUTLIERS_MAX), n);
}
__global__ __launch_bounds__(256) void k_cell_hist(uint32_t n, const uint2* __restrict__ sorted, uint32_t* __restrict__ cells) {
    const uint32_t r = blockIdx.x * 256u + threadIdx.x, lane = threadIdx.x & 63u;
    n = without_outliers(n, cells);
    // the keys are sorted: neighbouring lanes mostly fall into the same cell, and one lane per run of equal cells adds
    // the run's length (one atomic per triangle on a word its neighbours want too costs ten times as much)
    const uint32_t cell = r < n ? sorted[r].x >> (30u - RT_CELL_BITS_MAX) : 0xffffffffu;
    const uint32_t before = (uint32_t)__shfl_up((int)cell, 1, 64);
    const bool first = lane == 0u || cell != before;
    const unsigned long long firsts = __builtin_amdgcn_ballot_w64(first);
    if (first && r < n) {
        const unsigned long long later = lane == 63u ? 0ull : firsts >> (lane + 1u);
        const uint32_t run = later ? (uint32_t)__builtin_ctzll(later) + 1u : 64u - lane;
        atomicAdd(cells + RT_CELLS_HIST + cell, run);
    }
}
__device__ __forceinline__ uint32_t wave_max(uint32_t v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d, 64));
    return v;
}
// With F = RT_CELL_BITS_MAX: a workgroup of 512 sums the counts of 512 finest cells, i.e. one cell of F - 9 bits, and
// leaves that count and the fullest cell it has seen at T = F, F - 3 and F - 6; one thread per finest cell
constexpr uint32_t kCoarse = 1u << (RT_CELL_BITS_MAX - 9u);        // cells of the level one workgroup sums up
__global__ __launch_bounds__(512) void k_cell_sums(uint32_t* __restrict__ cells) {
    __shared__ uint32_t s_part[8][3];
    const uint32_t i = blockIdx.x * 512u + threadIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t v = cells[RT_CELLS_HIST + i];
    const uint32_t m0 = wave_max(v);
    v += (uint32_t)__shfl_xor((int)v, 1, 64); v += (uint32_t)__shfl_xor((int)v, 2, 64); v += (uint32_t)__shfl_xor((int)v, 4, 64);
    const uint32_t m1 = wave_max(v);
    v += (uint32_t)__shfl_xor((int)v, 8, 64); v += (uint32_t)__shfl_xor((int)v, 16, 64); v += (uint32_t)__shfl_xor((int)v, 32, 64);
    if (lane == 0u) { s_part[wave][0] = m0; s_part[wave][1] = m1; s_part[wave][2] = v; }
    __syncthreads();
    if (threadIdx.x == 0u) {
        uint32_t a0 = 0u, a1 = 0u, a2 = 0u, sum = 0u;
        for (int w = 0; w < 8; w++) { a0 = max(a0, s_part[w][0]); a1 = max(a1, s_part[w][1]); a2 = max(a2, s_part[w][2]); sum += s_part[w][2]; }
        // the maxima go to the (still unused) table of cell starts; k_cell_choose gathers them: thousands of
        // atomics on three words would take longer than the rest of the build's bookkeeping together
        cells[RT_CELLS_HIST12 + blockIdx.x] = sum;
        cells[RT_CELLS_HIST12 + kCoarse + blockIdx.x] = a2;
        cells[RT_CELLS_START + blockIdx.x] = a0;
        cells[RT_CELLS_START + kCoarse + blockIdx.x] = a1;
    }
}
// fullest cells of the coarser levels (each the sum of eight of the level below), and the choice of T; one workgroup
// of 1024
__global__ __launch_bounds__(1024) void k_cell_choose(uint32_t* __restrict__ cells, uint32_t slots_log2, uint32_t t_cap) {
    static_assert(RT_CELL_BITS_MAX == 21u && kCoarse == 4096u, "the sums below go 12 -> 9 -> 6 -> 3 bits");
    __shared__ uint32_t s12[4096], s9[512], s6[64], s3[8], s_max[7];
    const uint32_t t = threadIdx.x;
    if (t < 7u) s_max[t] = 0u;
    __syncthreads();
    {   // fullest cells of T = 21, 18, 15, 12 from the per-workgroup values of k_cell_sums
        uint32_t m0 = 0u, m1 = 0u, m2 = 0u, m3 = 0u;
        for (uint32_t j = t; j < kCoarse; j += 1024u) {
            m0 = max(m0, cells[RT_CELLS_START + j]);
            m1 = max(m1, cells[RT_CELLS_START + kCoarse + j]);
            m2 = max(m2, cells[RT_CELLS_HIST12 + kCoarse + j]);
            const uint32_t c = cells[RT_CELLS_HIST12 + j];
            s12[j] = c;
            m3 = max(m3, c);
        }
        m0 = wave_max(m0); m1 = wave_max(m1); m2 = wave_max(m2); m3 = wave_max(m3);
        if ((t & 63u) == 0u) { atomicMax(&s_max[0], m0); atomicMax(&s_max[1], m1); atomicMax(&s_max[2], m2); atomicMax(&s_max[3], m3); }
    }
    __syncthreads();
    if (t < 512u) { uint32_t v = 0u; for (uint32_t j = 0; j < 8u; j++) v += s12[8u * t + j]; s9[t] = v; atomicMax(&s_max[4], v); }
    __syncthreads();
    if (t < 64u) { uint32_t v = 0u; for (uint32_t j = 0; j < 8u; j++) v += s9[8u * t + j]; s6[t] = v; atomicMax(&s_max[5], v); }
    __syncthreads();
    if (t < 8u) { uint32_t v = 0u; for (uint32_t j = 0; j < 8u; j++) v += s6[8u * t + j]; s3[t] = v; atomicMax(&s_max[6], v); }
    __syncthreads();
    if (t == 0u) {
        uint32_t chosen = 0u;
        for (uint32_t k = 0; k < 7u; k++) {
            cells[1u + k] = s_max[k];
            const uint32_t T = RT_CELL_BITS_MAX - 3u * k;
            // at least two slots per cell, and room for the fullest cell's triangles in its slots
            if (!chosen && T <= t_cap && T + 1u <= slots_log2 && s_max[k] <= 2u << (slots_log2 - T)) chosen = T;
        }
        cells[0] = chosen;
        cells[RT_CELLS_HIST] = 0u;          // the per-cell counts are done with: their table becomes k_cell_starts' list of long gaps
    }
}
// first rank of every cell at the chosen T (cells without triangles included), and the total behind the last.
// The lane of a cell's first triangle also writes the cells without triangles before it; a run of kGapInline or more of
// them goes on a list that k_cell_fill works off with whole workgroups (a scene whose triangles sit in a few of many
// cells would otherwise have single lanes write thousands of entries each).
constexpr uint32_t kGapInline = 32u;
__device__ __forceinline__ void cell_run(uint32_t* __restrict__ cells, uint32_t from, uint32_t to, uint32_t value) {     // start[from .. to] = value
    uint32_t* start = cells + RT_CELLS_START;
    if (to - from < kGapInline) {
        for (uint32_t k = from; k <= to; k++) start[k] = value;
    } else {
        uint32_t* list = cells + RT_CELLS_HIST;               // [0] = runs listed, from [4] on (from, to, value) each
        const uint32_t i = atomicAdd(list, 1u);               // at most 2^RT_CELL_BITS_MAX / kGapInline runs: 3 words each fit the table
        list[4u + 3u * i] = from; list[5u + 3u * i] = to; list[6u + 3u * i] = value;
    }
}
__global__ __launch_bounds__(256) void k_cell_starts(uint32_t n, const uint2* __restrict__ sorted, uint32_t* __restrict__ cells) {
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    n = without_outliers(n, cells);
    if (n == 0u && r == 0u) cell_run(cells, 0u, 1u << cells[0], 0u);      // nothing but outliers: every cell is empty
    if (r >= n) return;
    const uint32_t T = cells[0], shift = 30u - T;          // codes have 30 bits: T = 0 puts everything into cell 0
    const uint32_t c = sorted[r].x >> shift;
    uint32_t from = 0u;                                     // cells from..c start at r
    bool boundary = true;
    if (r) {
        const uint32_t before = sorted[r - 1u].x >> shift;
        boundary = before != c;
        from = before + 1u;
    }
    if (boundary) cell_run(cells, from, c, r);
    if (r == n - 1u) cell_run(cells, c + 1u, 1u << T, n);
}
__global__ __launch_bounds__(256) void k_cell_fill(uint32_t* __restrict__ cells) {
    const uint32_t* list = cells + RT_CELLS_HIST;
    uint32_t* start = cells + RT_CELLS_START;
    const uint32_t runs = list[0];
    for (uint32_t i = blockIdx.x; i < runs; i += gridDim.x) {
        const uint32_t from = list[4u + 3u * i], to = list[5u + 3u * i], value = list[6u + 3u * i];
        for (uint32_t k = from + threadIdx.x; k <= to; k += 256u) start[k] = value;
    }
}

// rt_accel_refit: the sorted (code, 3 * triangle) pairs of the last build, as far as k_leaves reads them (the triangle),
// from the leaf records that build wrote
__global__ __launch_bounds__(256) void k_order_from_leaves(uint32_t n, const float4* __restrict__ sorted_tris, uint2* __restrict__ keys) {
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    if (2u * g >= n) return;
    const float4 l4 = sorted_tris[RT_LEAF_STRIDE * (size_t)g + 4u];
    keys[2u * g] = make_uint2(0u, __float_as_uint(l4.z));
    if (2u * g + 1u < n) keys[2u * g + 1u] = make_uint2(0u, __float_as_uint(l4.w));
}

__global__ __launch_bounds__(256) void k_expand_keys(uint32_t n, const uint2* __restrict__ keys, uint4* __restrict__ morton) {
    uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t < n) { uint2 k = keys[t]; morton[t] = make_uint4(k.x, k.y, 0u, 0u); }
}

// ------------------------------------------------------------------ radix sort ----

// wave64 inclusive scan by lane shuffles
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(v, d, 64);
        if (lane >= (uint32_t)d) v += o;
    }
    return v;
}

// exclusive scan of one value per thread over a 256-thread workgroup; *total receives the sum.
// tmp: 4 words of LDS. Contains two barriers.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* tmp, uint32_t* total) {
    uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = wave_incl_scan(v, lane);
    __syncthreads();                       // tmp may still be read from a previous call
    if (lane == 63u) tmp[wave] = incl;
    __syncthreads();
    uint32_t w0 = tmp[0], w1 = tmp[1], w2 = tmp[2], w3 = tmp[3];
    uint32_t before = (wave > 0 ? w0 : 0u) + (wave > 1 ? w1 : 0u) + (wave > 2 ? w2 : 0u);
    *total = w0 + w1 + w2 + w3;
    return before + incl - v;
}

// Per-tile digit counts. STRIDE = 32-bit words per element (4: reference uint4 records, 2: packed
// pairs). REF additionally performs the reference count shader's side effects: copy .xy to the
// temp buffer and add the counts to the global counters' .x.
template <int STRIDE, bool REF>
__global__ __launch_bounds__(256) void k_sort_hist(const uint32_t* __restrict__ in, uint32_t* __restrict__ temp, uint32_t n,
                                                   uint32_t shift, uint32_t* __restrict__ table, uint32_t tiles,
                                                   uint4* __restrict__ freq) {
    __shared__ uint32_t hist[4][256];
    const uint32_t tid = threadIdx.x, wave = tid >> 6;
    hist[0][tid] = 0u; hist[1][tid] = 0u; hist[2][tid] = 0u; hist[3][tid] = 0u;
    __syncthreads();
    const uint32_t base = blockIdx.x * (uint32_t)kTile;
#pragma unroll
    for (int j = 0; j < kItems; j++) {
        uint32_t i = base + (uint32_t)j * 256u + tid;
        if (i < n) {
            uint2 kv = *reinterpret_cast<const uint2*>(in + (size_t)i * STRIDE);
            if (REF) *reinterpret_cast<uint2*>(temp + (size_t)i * STRIDE) = kv;
            atomicAdd(&hist[wave][(kv.x >> shift) & 0xffu], 1u);
        }
    }
    __syncthreads();
    uint32_t c = hist[0][tid] + hist[1][tid] + hist[2][tid] + hist[3][tid];
    table[(size_t)tid * tiles + blockIdx.x] = c;
    if (REF && c) atomicAdd(&freq[tid].x, c);
}

// One workgroup per digit: exclusive scan of that digit's counts over the tiles, in place.
__global__ __launch_bounds__(256) void k_sort_scan(uint32_t* __restrict__ table, uint32_t tiles, uint32_t* __restrict__ totals) {
    __shared__ uint32_t tmp[4];
    uint32_t* row = table + (size_t)blockIdx.x * tiles;
    const uint32_t tid = threadIdx.x;
    uint32_t carry = 0u;
    for (uint32_t base = 0; base < tiles; base += 1024u) {
        // four consecutive entries per thread
        uint32_t i0 = base + tid * 4u;
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = (i0 + k < tiles) ? row[i0 + k] : 0u;
        uint32_t sum = v[0] + v[1] + v[2] + v[3], total;
        uint32_t ex = block_excl_scan(sum, tmp, &total) + carry;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (i0 + k < tiles) row[i0 + k] = ex;
            ex += v[k];
        }
        carry += total;
    }
    if (tid == 0) totals[blockIdx.x] = carry;
}

// shader/CS_SortPrefixSum.hlsl: .y = exclusive prefix sum of .x over the 256 counters
__global__ __launch_bounds__(256) void k_freq_scan(uint4* __restrict__ freq) {
    __shared__ uint32_t tmp[4];
    uint32_t total;
    uint32_t ex = block_excl_scan(freq[threadIdx.x].x, tmp, &total);
    freq[threadIdx.x].y = ex;
}

// Stable scatter of one tile. Keys are held wave-striped (wave w owns elements
// [w*64*kItems, (w+1)*64*kItems) of the tile; item j of lane l is element j*64 + l), so the order
// (wave, item, lane) is the input order. Rank inside the wave comes from a ballot match on the 8
// digit bits; per-wave running digit counts live in LDS. Keys are then reordered through LDS so
// that the global writes are runs of consecutive addresses per digit.
template <int STRIDE_IN, int STRIDE_OUT, bool REF>
__global__ __launch_bounds__(256) void k_sort_scatter(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n,
                                                      uint32_t shift, const uint32_t* __restrict__ table, uint32_t tiles,
                                                      const uint32_t* __restrict__ totals, uint4* __restrict__ freq) {
    __shared__ uint32_t s_keys[kTile];
    __shared__ uint32_t s_vals[kTile];
    __shared__ uint32_t wave_hist[4][256];
    __shared__ uint32_t digit_start[256];
    __shared__ uint32_t gbase[256];
    __shared__ uint32_t tmp[4];

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t tile = blockIdx.x;
    const uint32_t tile_base = tile * (uint32_t)kTile;
    const uint32_t tile_count = min((uint32_t)kTile, n - tile_base);

    wave_hist[0][tid] = 0u; wave_hist[1][tid] = 0u; wave_hist[2][tid] = 0u; wave_hist[3][tid] = 0u;

    uint32_t key[kItems], val[kItems], rank[kItems];
    const uint32_t wbase = wave * 64u * (uint32_t)kItems;
#pragma unroll
    for (int j = 0; j < kItems; j++) {
        uint32_t e = wbase + (uint32_t)j * 64u + lane;
        key[j] = 0u; val[j] = 0u;
        if (e < tile_count) {
            uint2 kv = *reinterpret_cast<const uint2*>(in + (size_t)(tile_base + e) * STRIDE_IN);
            key[j] = kv.x; val[j] = kv.y;
        }
    }
    __syncthreads();

#pragma unroll
    for (int j = 0; j < kItems; j++) {
        const uint32_t e = wbase + (uint32_t)j * 64u + lane;
        const bool valid = e < tile_count;
        const uint32_t d = (key[j] >> shift) & 0xffu;
        unsigned long long peers = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const bool bit = (d >> b) & 1u;
            unsigned long long bb = __builtin_amdgcn_ballot_w64(valid && bit);
            peers &= bit ? bb : ~bb;
        }
        const uint32_t lower = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
        const uint32_t cnt = (uint32_t)__popcll(peers);
        uint32_t pre = 0u;
        if (valid) pre = wave_hist[wave][d];
        __builtin_amdgcn_wave_barrier();
        if (valid && lower == 0u) wave_hist[wave][d] = pre + cnt;
        __builtin_amdgcn_wave_barrier();
        rank[j] = pre + lower;
    }
    __syncthreads();

    {   // thread = digit: digit starts inside the tile, per-wave offsets, global bases
        uint32_t c0 = wave_hist[0][tid], c1 = wave_hist[1][tid], c2 = wave_hist[2][tid], c3 = wave_hist[3][tid];
        uint32_t total;
        uint32_t start = block_excl_scan(c0 + c1 + c2 + c3, tmp, &total);
        uint32_t dbase;
        if (REF) {
            dbase = freq[tid].y;
        } else {
            uint32_t t2;
            dbase = block_excl_scan(totals[tid], tmp, &t2);
        }
        digit_start[tid] = start;
        gbase[tid] = dbase + table[(size_t)tid * tiles + tile];
        wave_hist[0][tid] = start;
        wave_hist[1][tid] = start + c0;
        wave_hist[2][tid] = start + c0 + c1;
        wave_hist[3][tid] = start + c0 + c1 + c2;
    }
    __syncthreads();

#pragma unroll
    for (int j = 0; j < kItems; j++) {
        const uint32_t e = wbase + (uint32_t)j * 64u + lane;
        if (e < tile_count) {
            uint32_t p = wave_hist[wave][(key[j] >> shift) & 0xffu] + rank[j];
            s_keys[p] = key[j];
            s_vals[p] = val[j];
        }
    }
    __syncthreads();

#pragma unroll
    for (int j = 0; j < kItems; j++) {
        uint32_t p = (uint32_t)j * 256u + tid;
        if (p < tile_count) {
            uint32_t k = s_keys[p];
            uint32_t d = (k >> shift) & 0xffu;
            uint32_t g = gbase[d] + (p - digit_start[d]);
            *reinterpret_cast<uint2*>(out + (size_t)g * STRIDE_OUT) = make_uint2(k, s_vals[p]);
        }
    }
    if (REF && tile == 0u) freq[tid].x = 0u;      // shader/CS_SortSort.hlsl:34
}

// Morton codes of the packed triangles (CS_GenerateMortonCodes.hlsl:58-66) and the first sorting pass's k_sort_hist in
// one: a workgroup makes the (code, 3 * triangle) keys of one sorting tile (the 256 x kItems layout k_sort_hist reads)
// and counts their lowest digits while it has them.
// dbounds (optional): scene bounds in device memory, (min.xyz, pad, max.xyz, pad); overrides sb
__global__ __launch_bounds__(256) void k_morton_hist(SceneBounds sb, const float* __restrict__ dbounds, uint32_t n,
                                                     const float4* __restrict__ tris, uint2* __restrict__ keys,
                                                     uint32_t* __restrict__ table, uint32_t tiles) {
    __shared__ uint32_t hist[4][256];
    const uint32_t tid = threadIdx.x, wave = tid >> 6;
    hist[0][tid] = 0u; hist[1][tid] = 0u; hist[2][tid] = 0u; hist[3][tid] = 0u;
    if (dbounds) {
        sb.mn[0] = dbounds[0]; sb.mn[1] = dbounds[1]; sb.mn[2] = dbounds[2];
        sb.mx[0] = dbounds[4]; sb.mx[1] = dbounds[5]; sb.mx[2] = dbounds[6];
    }
    __syncthreads();
    const uint32_t base = blockIdx.x * (uint32_t)kTile;
#pragma unroll
    for (int j = 0; j < kItems; j++) {
        const uint32_t t = base + (uint32_t)j * 256u + tid;
        if (t < n) {
            const uint32_t ci = t * 3u;
            const float4 a = tris[ci], b = tris[ci + 1], c = tris[ci + 2];
            const uint32_t code = morton_code(a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z, sb.mn, sb.mx);
            keys[t] = make_uint2(code, ci);
            atomicAdd(&hist[wave][code & 0xffu], 1u);
        }
    }
    __syncthreads();
    table[(size_t)tid * tiles + blockIdx.x] = hist[0][tid] + hist[1][tid] + hist[2][tid] + hist[3][tid];
}

// ------------------------------------------------------------------- hierarchy ----

struct Box { float mn[3]; float mx[3]; };

__device__ __forceinline__ void store_node(rt_aabb* nodes, uint32_t idx, const Box& b, uint32_t l0, uint32_t l1) {
    float4* p = reinterpret_cast<float4*>(nodes + idx);
    p[0] = make_float4(b.mn[0], b.mn[1], b.mn[2], b.mx[0]);
    p[1] = make_float4(b.mx[1], b.mx[2], __uint_as_float(l0), __uint_as_float(l1));
}

// The two triangles of a leaf, interleaved for the packed two-triangle test (80 bytes, one leaf per 128-byte line):
//   (a0.x a1.x a0.y a1.y) (a0.z a1.z b0.x b1.x) (b0.y b1.y b0.z b1.z) (c0.x c1.x c0.y c1.y) (c0.z c1.z ci0 ci1)
// p[0..2] = corners of triangle 0, p[3..5] of triangle 1 (a copy of triangle 0, ci1 = 0xffffffff, if there is none)
__device__ __forceinline__ void leaf_triangle_record(float4* o, const float (*p)[3], uint32_t ci0, uint32_t ci1) {
    o[0] = make_float4(p[0][0], p[3][0], p[0][1], p[3][1]);
    o[1] = make_float4(p[0][2], p[3][2], p[1][0], p[4][0]);
    o[2] = make_float4(p[1][1], p[4][1], p[1][2], p[4][2]);
    o[3] = make_float4(p[2][0], p[5][0], p[2][1], p[5][1]);
    o[4] = make_float4(p[2][2], p[5][2], __uint_as_float(ci0), __uint_as_float(ci1));
}
__device__ __forceinline__ void store_leaf_triangles(float4* sorted_tris, uint32_t leaf, const float (*p)[3], uint32_t ci0,
                                                     uint32_t ci1) {
    float4* o = sorted_tris + RT_LEAF_STRIDE * (size_t)leaf;
    leaf_triangle_record(o, p, ci0, ci1);
    // the rest of the line is written as well: whole 128-byte lines go to memory faster than 80 bytes of each
    o[5] = o[6] = o[7] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// The accel's private copy of the hierarchy holds sibling pairs: children 2j and 2j+1 of every node share one
// 64-byte line, components interleaved so that one packed operation covers both boxes:
//   (min.x0, min.x1, min.y0, min.y1) (min.z0, min.z1, max.x0, max.x1) (max.y0, max.y1, max.z0, max.z1) (links0, links1)
// The pair of nodes 2j, 2j+1 of level k lives in 32-byte slots aoff[k] + 2j, aoff[k] + 2j + 1.
__device__ __forceinline__ void store_pair(rt_aabb* pairs, uint32_t slot, const Box& a, const Box& b, uint32_t a0, uint32_t a1,
                                           uint32_t b0, uint32_t b1) {
    float4* p = reinterpret_cast<float4*>(pairs + slot);
    p[0] = make_float4(a.mn[0], b.mn[0], a.mn[1], b.mn[1]);
    p[1] = make_float4(a.mn[2], b.mn[2], a.mx[0], b.mx[0]);
    p[2] = make_float4(a.mx[1], b.mx[1], a.mx[2], b.mx[2]);
    p[3] = make_float4(__uint_as_float(a0), __uint_as_float(a1), __uint_as_float(b0), __uint_as_float(b1));
}
// one node's half of its pair (used where the sibling is produced by another workgroup)
__device__ __forceinline__ void store_pair_half(rt_aabb* pairs, uint32_t pair_slot, uint32_t child, const Box& a, uint32_t a0,
                                                uint32_t a1) {
    float* p = reinterpret_cast<float*>(pairs + pair_slot);
    p[0 + child] = a.mn[0]; p[2 + child] = a.mn[1]; p[4 + child] = a.mn[2];
    p[6 + child] = a.mx[0]; p[8 + child] = a.mx[1]; p[10 + child] = a.mx[2];
    p[12 + 2u * child] = __uint_as_float(a0); p[13 + 2u * child] = __uint_as_float(a1);
}
// The four-grandchildren records (rt_quad_records) are written along with the pair records: the pair of the children
// of node c (index ci of level k + 1) is one half of the 64-byte record of c's parent, which exists where level k + 2
// does and k is even. A record holds boxes that CONTAIN the true ones, in binary16: every minimum rounded down, every
// maximum rounded up. Half p of record h is six dwords at dword 6p, one per component (min x y z, max x y z), each a
// pair (first node, second node) of binary16 values; dwords 12 - 15 are padding. The last node of a level with an odd
// count has no sibling: its parent's second half is NaNs.
__device__ __forceinline__ uint32_t half_bits_below(float x) {      // the largest binary16 value <= x
    _Float16 hf = (_Float16)x;                                      // round to nearest
    uint32_t b = (uint32_t)__builtin_bit_cast(unsigned short, hf);
    if ((float)hf > x) b = (b == 0u) ? 0x8001u : ((b & 0x8000u) ? b + 1u : b - 1u);      // one step towards -infinity
    // no denormals in the records (whatever an instruction does with them, the boxes stay valid): down to the next
    // normal value or to zero
    if (!(b & 0x7c00u) && (b & 0x03ffu)) b = (b & 0x8000u) ? 0x8400u : 0x0000u;
    return b;                                                       // NaN stays NaN; beyond the range: -inf or 65504
}
__device__ __forceinline__ uint32_t half_bits_above(float x) { return half_bits_below(-x) ^ 0x8000u; }
__device__ __forceinline__ void quad_half_dwords(const Box& a, const Box& b, uint32_t* d) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
        d[c] = half_bits_below(a.mn[c]) | (half_bits_below(b.mn[c]) << 16);
        d[3 + c] = half_bits_above(a.mx[c]) | (half_bits_above(b.mx[c]) << 16);
    }
}
__device__ __forceinline__ uint32_t* quad_half(float4* quads, const rt_levels& lv, uint32_t k, uint32_t ci) {
    const uint32_t c = lv.aoff[k + 1u] + ci;                   // slot of node c
    return reinterpret_cast<uint32_t*>(quads + RT_QUAD_STRIDE * (size_t)(c >> 1)) + 6u * (c & 1u);
}
__device__ __forceinline__ bool quad_level(const rt_levels& lv, uint32_t k) { return !(k & 1u) && k + 2u < lv.count; }
__device__ __forceinline__ void store_quad_pair(float4* quads, const rt_levels& lv, uint32_t k, uint32_t ci, const Box& a,
                                                const Box& b) {
    uint32_t* o = quad_half(quads, lv, k, ci);
    uint32_t d[6];
    quad_half_dwords(a, b, d);
    uint2* o2 = reinterpret_cast<uint2*>(o);                   // halves start at multiples of 8 bytes
    o2[0] = make_uint2(d[0], d[1]); o2[1] = make_uint2(d[2], d[3]); o2[2] = make_uint2(d[4], d[5]);
    if (ci & 1u) {
        o2[3] = make_uint2(0u, 0u); o2[4] = make_uint2(0u, 0u);                 // padding
    } else if (ci + 1u >= lv.n[k + 1u]) {
        const uint2 nan = make_uint2(0x7e007e00u, 0x7e007e00u);
        o2[3] = nan; o2[4] = nan; o2[5] = nan;
        o2[6] = make_uint2(0u, 0u); o2[7] = make_uint2(0u, 0u);
    }
}
// one node's part of its pair (where the sibling is produced by another workgroup); child = 0 or 1
__device__ __forceinline__ void store_quad_pair_half(float4* quads, const rt_levels& lv, uint32_t k, uint32_t ci, uint32_t child,
                                                     const Box& a) {
    uint32_t* o = quad_half(quads, lv, k, ci);
    unsigned short* p = reinterpret_cast<unsigned short*>(o) + child;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        p[2 * c] = (unsigned short)half_bits_below(a.mn[c]);
        p[2 * (3 + c)] = (unsigned short)half_bits_above(a.mx[c]);
    }
    if (child == 0u && !(ci & 1u) && ci + 1u >= lv.n[k + 1u])
        for (int c = 0; c < 6; c++) o[6 + c] = 0x7e007e00u;
}

// the next lane's box, or - where the next node does not exist - a box of NaNs, which no box built from
// vertices can be (min/max drop NaN operands); the walk recognises a missing child by it
__device__ __forceinline__ Box next_lane_box(const Box& b, bool exists) {
    Box o;
    const float nan = __uint_as_float(0x7fc00000u);
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float mn = __shfl_down(b.mn[a], 1, 64), mx = __shfl_down(b.mx[a], 1, 64);     // by every lane
        o.mn[a] = exists ? mn : nan;
        o.mx[a] = exists ? mx : nan;
    }
    return o;
}

// Builds up to eight levels above level k0 from the (up to) 256 level-k0 boxes the workgroup holds in
// LDS slot order (slot s <-> node blockIdx.x*256 + s of level k0). Same arithmetic as
// shader/CS_BVHBuild.hlsl:28-47: min/max union, links (first child, second child or 0xffffffff).
// sparse: of the levels made here only those a later launch or a kernel reads are stored as nodes - every eighth (where
// k_upsweep takes over) and the two at the top (the trunk and its children); the records of the walk are complete either way.
__device__ __forceinline__ bool node_level_kept(const rt_levels& lv, uint32_t k) { return !(k & 7u) || k + 2u >= lv.count; }
__device__ __forceinline__ void upsweep_block(Box mine, uint32_t k0, const rt_levels& lv, rt_aabb* __restrict__ nodes,
                                              rt_aabb* __restrict__ pairs, float4* __restrict__ quads, float (*s_box)[6],
                                              bool sparse = false) {
    const uint32_t tid = threadIdx.x;
    uint32_t cnt = 256u;
    uint32_t k = k0;
#pragma unroll 1
    for (int step = 0; step < 8; step++) {
        if (k + 1u >= lv.count) break;
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 3; a++) { s_box[tid][a] = mine.mn[a]; s_box[tid][3 + a] = mine.mx[a]; }
        __syncthreads();
        k++;
        cnt >>= 1;
        const uint32_t gi = blockIdx.x * cnt + tid;
        const bool have = tid < cnt && gi < lv.n[k];
        uint32_t l0 = 0u, l1 = 0u;
        if (have) {
            const uint32_t c0 = 2u * gi;
            const bool has2 = (c0 + 1u) < lv.n[k - 1];
            const uint32_t s0 = 2u * tid, s1 = has2 ? s0 + 1u : s0;
#pragma unroll
            for (int a = 0; a < 3; a++) {
                mine.mn[a] = fminf(s_box[s0][a], s_box[s1][a]);
                mine.mx[a] = fmaxf(s_box[s0][3 + a], s_box[s1][3 + a]);
            }
            l0 = lv.off[k - 1] + c0;
            l1 = has2 ? l0 + 1u : 0xffffffffu;
            if (!sparse || node_level_kept(lv, k)) store_node(nodes, lv.off[k] + gi, mine, l0, l1);
        }
        if ((pairs || quads) && k + 1u < lv.count) {   // same for the whole grid; the trunk is in nobody's record
            if (cnt >= 2u) {
                const Box other = next_lane_box(mine, gi + 1u < lv.n[k]);
                const uint32_t o0 = __shfl_down(l0, 1, 64), o1 = __shfl_down(l1, 1, 64);
                if (have && !(gi & 1u)) {
                    if (pairs) store_pair(pairs, lv.aoff[k] + gi, mine, other, l0, l1, o0, o1);
                    if (quads && quad_level(lv, k)) store_quad_pair(quads, lv, k, gi >> 1, mine, other);
                }
            } else if (have) {
                const bool quad = quads && quad_level(lv, k);
                if (pairs) store_pair_half(pairs, lv.aoff[k] + (gi & ~1u), gi & 1u, mine, l0, l1);
                if (quad) store_quad_pair_half(quads, lv, k, gi >> 1, gi & 1u, mine);
                if (!(gi & 1u) && gi + 1u >= lv.n[k]) {        // an only child: mark the other half as missing
                    Box none;
                    for (int a = 0; a < 3; a++) none.mn[a] = none.mx[a] = __uint_as_float(0x7fc00000u);
                    if (pairs) store_pair_half(pairs, lv.aoff[k] + gi, 1u, none, 0xffffffffu, 0xffffffffu);
                    if (quad) store_quad_pair_half(quads, lv, k, gi >> 1, 1u, none);
                }
            }
        }
    }
}

// ---- which triangles share a leaf of the library's own hierarchy ----
//
// The walk tests the four leaves under a node two levels above them with one record, so the eight triangles of those four
// leaf slots may be dealt to the slots in any way: the boxes above stay what they are. Neighbours in the order of the
// keys make poor pairs - the box around two triangles that merely lie in the same small cell is several times the size of
// either - so the pairs are chosen by surface area: of the 105 ways to split eight triangles into four pairs, the one whose
// four boxes have the smallest total surface (what the number of rays that meet a box is proportional to).
struct PairTable {
    unsigned char item[105][8];      // matching k: pairs (item[k][0], item[k][1]), (item[k][2], item[k][3]), ...
    uint32_t code[105];              // the same, three bits per position
};
constexpr PairTable make_pair_table() {
    PairTable t{};
    int k = 0;
    // pairs in ascending order of their lowest member, which is always the first of the pair: matching 0 is the identity
    for (int a = 1; a < 8; a++) {
        int r6[6] = {0, 0, 0, 0, 0, 0}, n6 = 0;
        for (int i = 1; i < 8; i++) if (i != a) r6[n6++] = i;
        for (int b = 1; b < 6; b++) {
            int r4[4] = {0, 0, 0, 0}, n4 = 0;
            for (int i = 1; i < 6; i++) if (i != b) r4[n4++] = r6[i];
            for (int c = 1; c < 4; c++) {
                int r2[2] = {0, 0}, n2 = 0;
                for (int i = 1; i < 4; i++) if (i != c) r2[n2++] = r4[i];
                const int m[8] = {0, a, r6[0], r6[b], r4[0], r4[c], r2[0], r2[1]};
                uint32_t code = 0u;
                for (int i = 0; i < 8; i++) { t.item[k][i] = (unsigned char)m[i]; code |= (uint32_t)m[i] << (3 * i); }
                t.code[k] = code;
                k++;
            }
        }
    }
    return t;
}
constexpr PairTable kPairs = make_pair_table();
static_assert(kPairs.code[0] == 0xfac688u, "the first matching keeps every triangle where it is");
__device__ const uint32_t d_pair_code[105] = {
#define RT_PC(i) kPairs.code[i]
#define RT_PC5(i) RT_PC(i), RT_PC(i + 1), RT_PC(i + 2), RT_PC(i + 3), RT_PC(i + 4)
#define RT_PC15(i) RT_PC5(i), RT_PC5(i + 5), RT_PC5(i + 10)
    RT_PC15(0), RT_PC15(15), RT_PC15(30), RT_PC15(45), RT_PC15(60), RT_PC15(75), RT_PC15(90)
#undef RT_PC15
#undef RT_PC5
#undef RT_PC
};
// index of the pair (a, b), a < b, among the 28
constexpr int pair_index(int a, int b) { return a * 7 - a * (a - 1) / 2 + (b - a - 1); }

// Leaves (shader/CS_BVHBuildLeaves.hlsl:28-63). MODE: kLeavesShader reads the reference layout (uint4 records,
// Indices -> Vertices); kLeavesPacked reads the packed pair array and the packed triangles and also emits the
// Morton-ordered triangle copy; kLeavesOwn is the packed form for the library's own hierarchy (ref_leaves and cells are
// given). FUSED continues with up to eight levels in the same launch.
// sparse_nodes (own hierarchy): leaf nodes and the levels nothing reads are not stored (node_level_kept).
constexpr int kLeavesShader = 0, kLeavesPacked = 1, kLeavesOwn = 2;
template <int MODE, bool FUSED>
__global__ __launch_bounds__(256) void k_leaves(uint32_t num_prims, rt_levels lv, const uint32_t* __restrict__ sorted,
                                                const uint32_t* __restrict__ indices, const rt_vertex* __restrict__ verts,
                                                const float4* __restrict__ tris, float4* __restrict__ sorted_tris,
                                                rt_aabb* __restrict__ nodes, rt_aabb* __restrict__ pairs, float4* __restrict__ quads,
                                                const uint32_t* __restrict__ cells, uint32_t slots_log2,
                                                const float4* __restrict__ ref_leaves, float sliver_pad,
                                                uint2* __restrict__ own_keys, SceneBounds sb, const float* __restrict__ dbounds,
                                                const uint32_t* __restrict__ axis_map, uint32_t* __restrict__ outliers,
                                                uint32_t sparse_nodes) {
    constexpr bool PACKED = MODE != kLeavesShader, OWN = MODE == kLeavesOwn;
    // Two builds use the extras. The reference's hierarchy, when the library's own is to follow: own_keys receives, for
    // the sorted entries 2g and 2g + 1, (key of the own hierarchy, place in the reference's order) - the corners are in
    // registers here, so the second sort needs no pass of its own over the triangles. The library's own hierarchy:
    // ref_leaves are the reference's leaf records, sorted[] holds (key, place in the reference's order), and a triangle's
    // corners and index slot come from the record of its leaf there (place / 2), one 128-byte line per triangle.
    // One block of LDS, used twice: first the workgroup's 256 leaf records are handed over through it so that they leave as
    // runs of whole lines (a lane storing its own record writes 16 bytes to each of 64 different lines per instruction:
    // half the rate) - five float4 per lane, six while the triangles of a group change places (own hierarchy) -, then the
    // up-sweep's boxes (its first barrier comes after the last read of the records). Less LDS = more workgroups per CU,
    // which is what hides the latency of the gathers.
    constexpr uint32_t kGroupStride = 25u;
    constexpr uint32_t kTriWords = OWN ? 64u * kGroupStride : (PACKED ? 256u * 5u : 1u);
    __shared__ float4 s_raw[kTriWords * 16u > 256u * 24u ? kTriWords : 256u * 24u / 16u];
    float4* s_tri = s_raw;
    float (*s_box)[6] = reinterpret_cast<float (*)[6]>(s_raw);
    // own hierarchy: the sixth 16 bytes of a leaf record hold the places of its two triangles in the reference's order
    uint2* s_rank = reinterpret_cast<uint2*>(s_raw + 256u * 5u);      // (behind the records; the groups of the pairing search are done with by then)
    static_assert(!OWN || 256u * 5u + 256u / 2u <= kTriWords, "records and ranks fit where the groups were");
    __shared__ uint32_t s_group[OWN ? 64 : 1];          // per group of four slots: its triangles, then the pairing chosen for them
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    const uint32_t leaves = (num_prims + 1u) / 2u;
    Box b;
#pragma unroll
    for (int a = 0; a < 3; a++) { b.mn[a] = 1e30f; b.mx[a] = -1e30f; }
    uint32_t ci0 = 0u, ci1 = 0xffffffffu;
    // leaf g holds the sorted triangles 2g and 2g + 1; with a cell layout (k_cell_choose) it is slot g, holds the
    // triangles first + 2j and first + 2j + 1 of its cell, j = its place among the cell's slots, or nothing
    uint32_t r0 = 2u * g;
    bool have = g < leaves, two = (2u * g + 1u) < num_prims;
    if (OWN && g < leaves) {
        const uint32_t wshift = slots_log2 - cells[0];
        const uint32_t c = g >> wshift, j = g & ((1u << wshift) - 1u);
        const uint32_t first = cells[RT_CELLS_START + c], count = cells[RT_CELLS_START + c + 1u] - first;
        r0 = first + 2u * j;
        have = 2u * j < count;
        two = 2u * j + 1u < count;
    }
    uint32_t place0 = 0u, place1 = 0xffffffffu;        // with ref_leaves: the triangles' places in the reference's order
    float corner[6][3];
#pragma unroll
    for (int i = 0; i < 6; i++) corner[i][0] = corner[i][1] = corner[i][2] = 0.0f;
    if (have) {
        constexpr int STRIDE = PACKED ? 2 : 4;
        ci0 = sorted[(size_t)r0 * STRIDE + 1];
        if (two) ci1 = sorted[(size_t)(r0 + 1u) * STRIDE + 1];
        if (OWN) {
            place0 = ci0; place1 = two ? ci1 : 0xffffffffu;
            // one half of an interleaved record (leaf_triangle_record): corners into corner[K .. K + 2], index slot into CI
#define RT_REF_TRIANGLE(PLACE, K, CI)                                                                                          \
            {                                                                                                                  \
                const float4* l = ref_leaves + RT_LEAF_STRIDE * (size_t)((PLACE) >> 1);                                          \
                const float4 l0 = l[0], l1 = l[1], l2 = l[2], l3 = l[3], l4 = l[4];                                              \
                const bool second = ((PLACE) & 1u) != 0u;                                                                        \
                corner[K][0] = second ? l0.y : l0.x; corner[K][1] = second ? l0.w : l0.z; corner[K][2] = second ? l1.y : l1.x;   \
                corner[K + 1][0] = second ? l1.w : l1.z; corner[K + 1][1] = second ? l2.y : l2.x; corner[K + 1][2] = second ? l2.w : l2.z; \
                corner[K + 2][0] = second ? l3.y : l3.x; corner[K + 2][1] = second ? l3.w : l3.z; corner[K + 2][2] = second ? l4.y : l4.x; \
                CI = __float_as_uint(second ? l4.w : l4.z);                                                                    \
            }
            RT_REF_TRIANGLE(place0, 0, ci0)
            if (two) RT_REF_TRIANGLE(place1, 3, ci1)
#undef RT_REF_TRIANGLE
        }
        if (!OWN) {
#pragma unroll
            for (int i = 0; i < 6; i++) {
                if (i < 3 || two) {
                    if (PACKED) {
                        const float4 p = tris[((i < 3) ? ci0 : ci1) + (uint32_t)(i % 3)];
                        corner[i][0] = p.x; corner[i][1] = p.y; corner[i][2] = p.z;
                    } else {
                        const float* p = verts[indices[((i < 3) ? ci0 : ci1) + (uint32_t)(i % 3)]].position;
                        corner[i][0] = p[0]; corner[i][1] = p[1]; corner[i][2] = p[2];
                    }
                }
            }
        }
    }
    if (OWN && slots_log2 - cells[0] >= 2u) {       // same for the whole grid
        // The library's own hierarchy: the eight triangles of four neighbouring leaf slots (one record of the walk) are
        // paired anew by surface area (see PairTable). The lanes of the four slots hand their triangles over through LDS,
        // each works out the best matching of the eight, and takes the two triangles of its place in it.
        // A group's triangles fill its places from the front; the last one of an odd number stays alone where it is.
        // The search runs in ONE of the workgroup's four waves, a lane per group of eight triangles: the instruction sequence
        // (a thousand vector instructions) is then issued once per 64 groups instead of once per 16, and the wave that runs
        // it changes with the workgroup, so that the four SIMDs of a CU share it. (Groups are kGroupStride float4 apart:
        // 24 hold the triangles, the 25th spreads the lanes' reads over the LDS banks.)
        float4* mine = s_tri + kGroupStride * (threadIdx.x >> 2) + 6u * (threadIdx.x & 3u);
        mine[0] = make_float4(corner[0][0], corner[0][1], corner[0][2], __uint_as_float(ci0));
        mine[1] = make_float4(corner[1][0], corner[1][1], corner[1][2], __uint_as_float(place0));
        mine[2] = make_float4(corner[2][0], corner[2][1], corner[2][2], 0.0f);
        mine[3] = make_float4(corner[3][0], corner[3][1], corner[3][2], __uint_as_float(ci1));
        mine[4] = make_float4(corner[4][0], corner[4][1], corner[4][2], __uint_as_float(place1));
        mine[5] = make_float4(corner[5][0], corner[5][1], corner[5][2], 0.0f);
        const unsigned long long first_mask = __builtin_amdgcn_ballot_w64(have), second_mask = __builtin_amdgcn_ballot_w64(have && two);
        const uint32_t quad = threadIdx.x & 60u;
        if (!(threadIdx.x & 3u))            // triangles of the group
            s_group[threadIdx.x >> 2] = (uint32_t)__popcll((first_mask >> quad) & 15ull) + (uint32_t)__popcll((second_mask >> quad) & 15ull);
        __syncthreads();
        if ((threadIdx.x >> 6) == (blockIdx.x & 3u)) {
            const uint32_t lane = threadIdx.x & 63u;
            const uint32_t paired = s_group[lane] & ~1u;           // triangles that take part
            uint32_t code = kPairs.code[0];
            if (paired >= 4u) {
                const float4* group = s_tri + kGroupStride * lane;
                float area[28];
                {
                    float mn[8][3], mx[8][3];
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const float4 a = group[3 * i], b2 = group[3 * i + 1], c = group[3 * i + 2];
                        mn[i][0] = fminf(fminf(a.x, b2.x), c.x); mn[i][1] = fminf(fminf(a.y, b2.y), c.y); mn[i][2] = fminf(fminf(a.z, b2.z), c.z);
                        mx[i][0] = fmaxf(fmaxf(a.x, b2.x), c.x); mx[i][1] = fmaxf(fmaxf(a.y, b2.y), c.y); mx[i][2] = fmaxf(fmaxf(a.z, b2.z), c.z);
                    }
#pragma unroll
                    for (int i = 0; i < 8; i++)
#pragma unroll
                        for (int j = i + 1; j < 8; j++) {
                            const float ex = fmaxf(mx[i][0], mx[j][0]) - fminf(mn[i][0], mn[j][0]);
                            const float ey = fmaxf(mx[i][1], mx[j][1]) - fminf(mn[i][1], mn[j][1]);
                            const float ez = fmaxf(mx[i][2], mx[j][2]) - fminf(mn[i][2], mn[j][2]);
                            // a pair of triangles that take part costs its box's surface; pairs of the others cost nothing, so
                            // that they stay together behind; mixed pairs are out
                            area[pair_index(i, j)] = (uint32_t)j < paired ? (ex * ey + ey * ez) + ez * ex
                                                                          : ((uint32_t)i < paired ? __builtin_inff() : 0.0f);
                        }
                }
                float best = __builtin_inff();
                uint32_t best_k = 0u;
#pragma unroll
                for (int k = 0; k < 105; k++) {
                    const float cost = (area[pair_index(kPairs.item[k][0], kPairs.item[k][1])] + area[pair_index(kPairs.item[k][2], kPairs.item[k][3])]) +
                                       (area[pair_index(kPairs.item[k][4], kPairs.item[k][5])] + area[pair_index(kPairs.item[k][6], kPairs.item[k][7])]);
                    if (cost < best) { best = cost; best_k = (uint32_t)k; }
                }
                code = d_pair_code[best_k];
            }
            s_group[lane] = code;
        }
        __syncthreads();
        const uint32_t code = s_group[threadIdx.x >> 2];
        {   // (every lane: a slot without triangles reads back the zeros it wrote, which costs no branch and lets the
            // corners leave their registers while the search runs)
            const uint32_t at = 2u * (threadIdx.x & 3u);
            const float4* t0 = s_tri + kGroupStride * (threadIdx.x >> 2) + 3u * ((code >> (3u * at)) & 7u);
            const float4* t1 = s_tri + kGroupStride * (threadIdx.x >> 2) + 3u * ((code >> (3u * at + 3u)) & 7u);
            const float4 a0 = t0[0], b0 = t0[1], c0 = t0[2], a1 = t1[0], b1 = t1[1], c1 = t1[2];
            corner[0][0] = a0.x; corner[0][1] = a0.y; corner[0][2] = a0.z; ci0 = __float_as_uint(a0.w);
            corner[1][0] = b0.x; corner[1][1] = b0.y; corner[1][2] = b0.z; place0 = __float_as_uint(b0.w);
            corner[2][0] = c0.x; corner[2][1] = c0.y; corner[2][2] = c0.z;
            corner[3][0] = a1.x; corner[3][1] = a1.y; corner[3][2] = a1.z; ci1 = __float_as_uint(a1.w);
            corner[4][0] = b1.x; corner[4][1] = b1.y; corner[4][2] = b1.z; place1 = __float_as_uint(b1.w);
            corner[5][0] = c1.x; corner[5][1] = c1.y; corner[5][2] = c1.z;
        }
        __syncthreads();                               // the leaf records below go to the same LDS
    }
    if (g < leaves && !have) {                         // an empty slot: a box no ray enters, a triangle no ray hits
        if (PACKED) {
            float none[6][3];
#pragma unroll
            for (int i = 0; i < 6; i++) none[i][0] = none[i][1] = none[i][2] = 0.0f;
            leaf_triangle_record(s_tri + 5u * threadIdx.x, none, ci0, ci1);
        }
        if (!sparse_nodes || lv.count <= 2u) store_node(nodes, 1u + g, b, ci0 | 0x80000000u, ci1 | 0x80000000u);
    }
    if (have) {
        if (!two) {                                     // no second triangle: repeat the first
#pragma unroll
            for (int i = 0; i < 3; i++) { corner[3 + i][0] = corner[i][0]; corner[3 + i][1] = corner[i][1]; corner[3 + i][2] = corner[i][2]; }
        }
#pragma unroll
        for (int i = 0; i < 6; i++) {                   // (the repeated corners change no minimum or maximum)
            b.mn[0] = fminf(b.mn[0], corner[i][0]); b.mn[1] = fminf(b.mn[1], corner[i][1]); b.mn[2] = fminf(b.mn[2], corner[i][2]);
            b.mx[0] = fmaxf(b.mx[0], corner[i][0]); b.mx[1] = fmaxf(b.mx[1], corner[i][1]); b.mx[2] = fmaxf(b.mx[2], corner[i][2]);
        }
        if (MODE == kLeavesPacked && own_keys) {
            if (dbounds) {
                sb.mn[0] = dbounds[0]; sb.mn[1] = dbounds[1]; sb.mn[2] = dbounds[2];
                sb.mx[0] = dbounds[4]; sb.mx[1] = dbounds[5]; sb.mx[2] = dbounds[6];
            }
            float cen[2][3];
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int a = 0; a < 3; a++) cen[t][a] = 0.333333f * ((corner[3 * t][a] + corner[3 * t + 1][a]) + corner[3 * t + 2][a]);
            uint32_t key[2] = {spatial_key(cen[0][0], cen[0][1], cen[0][2], sb, axis_map), spatial_key(cen[1][0], cen[1][1], cen[1][2], sb, axis_map)};
            // An outlier: a triangle whose box has a sixteenth or more of the surface of the box of all centroids. It goes on
            // the list (while there is room) and gets the largest key, behind every triangle of the hierarchy.
            const float sx = sb.mx[0] - sb.mn[0], sy = sb.mx[1] - sb.mn[1], sz = sb.mx[2] - sb.mn[2];
            const float scene_area = (sx * sy + sy * sz) + sz * sx;
#pragma unroll
            for (int t = 0; t < 2; t++) {
                if (t == 1 && !two) break;
                const float* c0 = corner[3 * t]; const float* c1 = corner[3 * t + 1]; const float* c2 = corner[3 * t + 2];
                const float ex = fmaxf(fmaxf(c0[0], c1[0]), c2[0]) - fminf(fminf(c0[0], c1[0]), c2[0]);
                const float ey = fmaxf(fmaxf(c0[1], c1[1]), c2[1]) - fminf(fminf(c0[1], c1[1]), c2[1]);
                const float ez = fmaxf(fmaxf(c0[2], c1[2]), c2[2]) - fminf(fminf(c0[2], c1[2]), c2[2]);
                const float area = (ex * ey + ey * ez) + ez * ex;
                if (scene_area > 0.0f && area * 16.0f >= scene_area) {
                    const uint32_t at = atomicAdd(outliers, 1u);
                    if (at < RT_OUTLIERS_MAX) { outliers[1u + at] = r0 + (uint32_t)t; key[t] = 0xffffffffu; }
                }
            }
            if (two) *reinterpret_cast<uint4*>(own_keys + r0) = make_uint4(key[0], r0, key[1], r0 + 1u);
            else own_keys[r0] = make_uint2(key[0], r0);
        }
        if (PACKED) leaf_triangle_record(s_tri + 5u * threadIdx.x, corner, ci0, ci1);
        if (OWN) {
            // The library's own hierarchy only: room around a sliver. The hit test of a triangle whose altitude is a
            // tiny part of its longest edge is dominated by rounding near its long edges (its determinant is just
            // above the reference's cut-off there), and accepts rays that pass beside the triangle's box; the
            // reference finds such hits, because its leaf boxes hold an unrelated second triangle and are large. A box
            // that is sliver_pad (1/32) of the longest edge wider on every side lets those rays test the triangle here too.
            float pad = 0.0f;
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const float* c0 = corner[3 * t]; const float* c1 = corner[3 * t + 1]; const float* c2 = corner[3 * t + 2];
                const float e1[3] = {c1[0] - c0[0], c1[1] - c0[1], c1[2] - c0[2]}, e2[3] = {c2[0] - c0[0], c2[1] - c0[1], c2[2] - c0[2]};
                const float e3[3] = {c2[0] - c1[0], c2[1] - c1[1], c2[2] - c1[2]};
                const float n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
                const float longest = fmaxf(fmaxf(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2], e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]),
                                            e3[0] * e3[0] + e3[1] * e3[1] + e3[2] * e3[2]);
                const float twice_area2 = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
                if (twice_area2 < longest * longest * (1.0f / 4096.0f)) pad = fmaxf(pad, sqrtf(longest) * sliver_pad);      // altitude < longest / 64
            }
#pragma unroll
            for (int a = 0; a < 3; a++) { b.mn[a] -= pad; b.mx[a] += pad; }
        }
        if (!sparse_nodes || lv.count <= 2u) store_node(nodes, 1u + g, b, ci0 | 0x80000000u, ci1 | 0x80000000u);
    }
    if (PACKED) {
        // (an empty slot - cell layout only - is marked: its record is never read, because no ray enters its box,
        // and is not written)
        if (OWN) s_rank[threadIdx.x] = have ? make_uint2(place0, place1) : make_uint2(0xffffffffu, 0u);
        __syncthreads();
        const uint32_t first = blockIdx.x * 256u;
#pragma unroll
        for (uint32_t i = 0; i < 8u; i++) {
            const uint32_t e = i * 256u + threadIdx.x;             // float4 e of the workgroup's 256 x 128 bytes
            const uint32_t leaf = e >> 3, j = e & 7u;
            const uint2 rank = OWN ? s_rank[leaf] : make_uint2(0u, 0u);
            if (first + leaf < leaves && rank.x != 0xffffffffu)
                sorted_tris[RT_LEAF_STRIDE * (size_t)first + e] =
                    j < 5u ? s_tri[5u * leaf + j]
                           : (j == 5u ? make_float4(__uint_as_float(rank.x), __uint_as_float(rank.y), 0.0f, 0.0f)
                                      : make_float4(0.0f, 0.0f, 0.0f, 0.0f));
        }
    }
    if (PACKED && (pairs || quads)) {                  // same for the whole grid
        const Box other = next_lane_box(b, g + 1u < leaves);
        const uint32_t o0 = __shfl_down(ci0, 1, 64), o1 = __shfl_down(ci1, 1, 64);
        if (g < leaves && !(g & 1u)) {
            if (pairs) store_pair(pairs, lv.aoff[0] + g, b, other, ci0 | 0x80000000u, ci1 | 0x80000000u, o0 | 0x80000000u, o1 | 0x80000000u);
            if (quads && quad_level(lv, 0u)) store_quad_pair(quads, lv, 0u, g >> 1, b, other);
        }
    }
    if (FUSED) upsweep_block(b, 0u, lv, nodes, pairs, quads, s_box, sparse_nodes != 0u);
}

// The outliers of a build (RT_CELLS_OUTLIERS) as leaf records for k_traverse_spatial: in ascending order of their places in
// the reference's order, two per record, corners and index slots from the reference's leaf records. One wave.
__global__ __launch_bounds__(64) void k_outlier_leaves(const uint32_t* __restrict__ cells, const float4* __restrict__ ref_leaves,
                                                       float4* __restrict__ out) {
    __shared__ uint32_t s_place[RT_OUTLIERS_MAX + 1u];
    static_assert(RT_OUTLIERS_MAX == 64u, "one lane per outlier");
    const uint32_t lane = threadIdx.x;
    const uint32_t count = min(cells[RT_CELLS_OUTLIERS], RT_OUTLIERS_MAX);
    const uint32_t mine = lane < count ? cells[RT_CELLS_OUTLIERS + 1u + lane] : 0xffffffffu;
    uint32_t rank = 0u;                                   // places are distinct: the rank is a permutation
    for (int j = 0; j < 64; j++) rank += ((uint32_t)__shfl((int)mine, j, 64) < mine) ? 1u : 0u;
    s_place[lane] = 0xffffffffu;
    if (lane == 0u) s_place[RT_OUTLIERS_MAX] = 0xffffffffu;
    __syncthreads();
    if (lane < count) s_place[rank] = mine;
    __syncthreads();
    if (2u * lane >= count) return;
    const uint32_t place0 = s_place[2u * lane], place1 = s_place[2u * lane + 1u];
    float corner[6][3];
    uint32_t ci0 = 0u, ci1 = 0xffffffffu;
#define RT_REF_TRIANGLE(PLACE, K, CI)                                                                                          \
    {                                                                                                                          \
        const float4* l = ref_leaves + RT_LEAF_STRIDE * (size_t)((PLACE) >> 1);                                                \
        const float4 l0 = l[0], l1 = l[1], l2 = l[2], l3 = l[3], l4 = l[4];                                                    \
        const bool second = ((PLACE) & 1u) != 0u;                                                                              \
        corner[K][0] = second ? l0.y : l0.x; corner[K][1] = second ? l0.w : l0.z; corner[K][2] = second ? l1.y : l1.x;         \
        corner[K + 1][0] = second ? l1.w : l1.z; corner[K + 1][1] = second ? l2.y : l2.x; corner[K + 1][2] = second ? l2.w : l2.z; \
        corner[K + 2][0] = second ? l3.y : l3.x; corner[K + 2][1] = second ? l3.w : l3.z; corner[K + 2][2] = second ? l4.y : l4.x; \
        CI = __float_as_uint(second ? l4.w : l4.z);                                                                            \
    }
    RT_REF_TRIANGLE(place0, 0, ci0)
    if (place1 != 0xffffffffu) RT_REF_TRIANGLE(place1, 3, ci1)
#undef RT_REF_TRIANGLE
    if (place1 == 0xffffffffu)
        for (int i = 0; i < 3; i++) { corner[3 + i][0] = corner[i][0]; corner[3 + i][1] = corner[i][1]; corner[3 + i][2] = corner[i][2]; }
    float4* o = out + RT_LEAF_STRIDE * (size_t)lane;
    leaf_triangle_record(o, corner, ci0, ci1);
    o[5] = make_float4(__uint_as_float(place0), __uint_as_float(place1), 0.0f, 0.0f);
    o[6] = o[7] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// continues the up-sweep from level k0 (already in memory), eight levels per launch
__global__ __launch_bounds__(256) void k_upsweep(uint32_t k0, rt_levels lv, rt_aabb* __restrict__ nodes,
                                                 rt_aabb* __restrict__ pairs, float4* __restrict__ quads, uint32_t sparse_nodes) {
    __shared__ float s_box[256][6];
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    Box b;
#pragma unroll
    for (int a = 0; a < 3; a++) { b.mn[a] = 1e30f; b.mx[a] = -1e30f; }
    if (g < lv.n[k0]) {
        const float4* p = reinterpret_cast<const float4*>(nodes + lv.off[k0] + g);
        float4 lo = p[0], hi = p[1];
        b.mn[0] = lo.x; b.mn[1] = lo.y; b.mn[2] = lo.z; b.mx[0] = lo.w; b.mx[1] = hi.x; b.mx[2] = hi.y;
    }
    upsweep_block(b, k0, lv, nodes, pairs, quads, s_box, sparse_nodes != 0u);
}

// one level, reference parameters (shader/CS_BVHBuild.hlsl:25-48)
__global__ __launch_bounds__(256) void k_level(rt_bvh_info info, rt_aabb* __restrict__ nodes) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= (info.num_children + 1u) / 2u) return;
    uint32_t i1 = 2u * t + info.previous_index, i2 = i1 + 1u;
    const float4* p = reinterpret_cast<const float4*>(nodes + i1);
    float4 a0 = p[0], a1 = p[1], b0 = a0, b1 = a1;
    if ((i2 - info.previous_index) < info.num_children) { b0 = p[2]; b1 = p[3]; }
    else i2 = 0xffffffffu;
    Box r;
    r.mn[0] = fminf(a0.x, b0.x); r.mn[1] = fminf(a0.y, b0.y); r.mn[2] = fminf(a0.z, b0.z);
    r.mx[0] = fmaxf(a0.w, b0.w); r.mx[1] = fmaxf(a1.x, b1.x); r.mx[2] = fmaxf(a1.y, b1.y);
    store_node(nodes, info.current_index + t, r, i1, i2);
}

// Private copy from a hierarchy in the reference's layout (for callers that trace without an accel): one
// thread per sibling pair of every level below the trunk
__global__ __launch_bounds__(256) void k_pairs_from_nodes(rt_levels lv, uint32_t total_pairs, const rt_aabb* __restrict__ nodes,
                                                          rt_aabb* __restrict__ pairs) {
    uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= total_pairs) return;
    uint32_t k = 0u;
    for (; k + 1u < lv.count; k++) {
        const uint32_t np = (lv.n[k] + 1u) / 2u;
        if (t < np) break;
        t -= np;
    }
    const uint32_t i = 2u * t;
    const float4* a = reinterpret_cast<const float4*>(nodes + lv.off[k] + i);
    const float4 alo = a[0], ahi = a[1];
    Box ba, bb;
    ba.mn[0] = alo.x; ba.mn[1] = alo.y; ba.mn[2] = alo.z; ba.mx[0] = alo.w; ba.mx[1] = ahi.x; ba.mx[2] = ahi.y;
    uint32_t b0 = 0xffffffffu, b1 = 0xffffffffu;
    if (i + 1u < lv.n[k]) {
        const float4 blo = a[2], bhi = a[3];
        bb.mn[0] = blo.x; bb.mn[1] = blo.y; bb.mn[2] = blo.z; bb.mx[0] = blo.w; bb.mx[1] = bhi.x; bb.mx[2] = bhi.y;
        b0 = __float_as_uint(bhi.z); b1 = __float_as_uint(bhi.w);
    } else {
        for (int c = 0; c < 3; c++) bb.mn[c] = bb.mx[c] = __uint_as_float(0x7fc00000u);
    }
    store_pair(pairs, lv.aoff[k] + i, ba, bb, __float_as_uint(ahi.z), __float_as_uint(ahi.w), b0, b1);
}

// The four-grandchildren records (rt_quad_records) from the pair records: the record of slot h holds the pair records
// of its children 2h and 2h + 1, widened to binary16 (see store_quad_pair). One thread per slot below
// 2^(root level - 1); slots an odd number of levels above the leaves, and slots of nodes that do not exist, have no record.
__device__ __forceinline__ void pair_record_boxes(const float4* r, Box& a, Box& b) {
    const float4 q0 = r[0], q1 = r[1], q2 = r[2];
    a.mn[0] = q0.x; b.mn[0] = q0.y; a.mn[1] = q0.z; b.mn[1] = q0.w;
    a.mn[2] = q1.x; b.mn[2] = q1.y; a.mx[0] = q1.z; b.mx[0] = q1.w;
    a.mx[1] = q2.x; b.mx[1] = q2.y; a.mx[2] = q2.z; b.mx[2] = q2.w;
}
__global__ __launch_bounds__(256) void k_quads_from_pairs(rt_levels lv, uint32_t slots, const float4* __restrict__ pairs,
                                                          float4* __restrict__ quads) {
    const uint32_t h = blockIdx.x * 256u + threadIdx.x;
    if (h == 0u || h >= slots) return;
    const uint32_t root_level = lv.count - 1u;
    const uint32_t depth = 31u - (uint32_t)__builtin_clz(h);
    const uint32_t k = root_level - depth;                    // the node's level above the leaves
    if ((k & 1u) || k < 2u) return;
    const uint32_t i = h - (1u << depth);
    if (i >= lv.n[k]) return;
    uint32_t d[16];
    Box a, b;
    pair_record_boxes(pairs + 4 * (size_t)(2u * h), a, b);
    quad_half_dwords(a, b, d);
    if (2u * i + 1u < lv.n[k - 1u]) {
        pair_record_boxes(pairs + 4 * (size_t)(2u * h + 1u), a, b);
        quad_half_dwords(a, b, d + 6);
    } else {
        for (int c = 0; c < 6; c++) d[6 + c] = 0x7e007e00u;
    }
    d[12] = d[13] = d[14] = d[15] = 0u;
    uint4* o = reinterpret_cast<uint4*>(quads + RT_QUAD_STRIDE * (size_t)h);
    for (int c = 0; c < 4; c++) o[c] = make_uint4(d[4 * c], d[4 * c + 1], d[4 * c + 2], d[4 * c + 3]);
}

// ... and the Morton-ordered packed triangles from the leaves' links
__global__ __launch_bounds__(256) void k_tris_from_leaves(uint32_t leaves, const rt_aabb* __restrict__ nodes,
                                                          const uint32_t* __restrict__ indices, const rt_vertex* __restrict__ verts,
                                                          float4* __restrict__ sorted_tris) {
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    if (g >= leaves) return;
    const uint32_t l0 = nodes[1u + g].link[0], l1 = nodes[1u + g].link[1];
    float corner[6][3];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const uint32_t link = (i < 3 || l1 == 0xffffffffu) ? l0 : l1;
        const uint32_t ci = link & 0x7fffffffu;
        const float* p = verts[indices[ci + (uint32_t)(i % 3)]].position;
        corner[i][0] = p[0]; corner[i][1] = p[1]; corner[i][2] = p[2];
    }
    store_leaf_triangles(sorted_tris, g, corner, l0 & 0x7fffffffu, l1 == 0xffffffffu ? l1 : (l1 & 0x7fffffffu));
}

inline uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1u) / b; }

inline SceneBounds make_bounds(const float* mn, const float* mx) {
    SceneBounds sb;
    for (int k = 0; k < 3; k++) { sb.mn[k] = mn[k]; sb.mx[k] = mx[k]; }
    return sb;
}

}  // namespace

int rt_private_copy_from_reference(rt_device* dev, uint32_t num_prims, const rt_aabb* nodes, const uint32_t* indices,
                                   const rt_vertex* verts, rt_aabb* pairs, float4* sorted_tris, float4* quads) {
    rt_levels lv;
    if (!rt_make_levels(num_prims, &lv)) return RT_ERR_TOO_FEW_PRIMITIVES;
    uint32_t total_pairs = 0u;
    for (uint32_t k = 0; k + 1u < lv.count; k++) total_pairs += (lv.n[k] + 1u) / 2u;
    hipLaunchKernelGGL(k_pairs_from_nodes, dim3(div_up(total_pairs, 256u)), dim3(256), 0, dev->stream, lv, total_pairs, nodes, pairs);
    hipLaunchKernelGGL(k_tris_from_leaves, dim3(div_up(lv.n[0], 256u)), dim3(256), 0, dev->stream, lv.n[0], nodes, indices, verts,
                       sorted_tris);
    if (quads && lv.count >= 3u) {
        const uint32_t slots = (uint32_t)rt_quad_records(&lv);
        hipLaunchKernelGGL(k_quads_from_pairs, dim3(div_up(slots, 256u)), dim3(256), 0, dev->stream, lv, slots, (const float4*)pairs,
                           quads);
    }
    return rt_launch_status();
}

// Leaves and levels of the library's own hierarchy over the slots, cells and sorted (key, place) pairs of the accel's last
// build, corners from the reference's leaf records. all_nodes: every level is also stored as nodes in the reference's
// layout (what the sibling-pair records are derived from); otherwise only the levels something reads (node_level_kept) -
// the walk itself reads the four-grandchildren records and the leaf records, and those are a third less to write.
static void own_leaves_and_levels(rt_accel* a, bool all_nodes) {
    hipStream_t s = a->dev->stream;
    const uint32_t slots = rt_spatial_slots(a->num_prims);
    uint32_t slots_log2 = 0u;
    while ((1u << slots_log2) < slots) slots_log2++;
    const rt_levels& sl = a->sp_lv;
    const uint32_t sparse = all_nodes ? 0u : 1u;
    hipLaunchKernelGGL((k_leaves<kLeavesOwn, true>), dim3(div_up(slots, 256u)), dim3(256), 0, s, 2u * slots, sl, (const uint32_t*)a->sp_sorted,
                       (const uint32_t*)nullptr, (const rt_vertex*)nullptr, a->tris, a->sp_tris, a->sp_nodes, (rt_aabb*)nullptr, a->sp_quads,
                       (const uint32_t*)a->cells, slots_log2, (const float4*)a->sorted_tris, a->dev->walk_sliver_pad, (uint2*)nullptr,
                       SceneBounds{}, (const float*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, sparse);
    for (uint32_t k0 = 8u; k0 + 1u < sl.count; k0 += 8u)
        hipLaunchKernelGGL(k_upsweep, dim3(div_up(sl.n[k0], 256u)), dim3(256), 0, s, k0, sl, a->sp_nodes, (rt_aabb*)nullptr, a->sp_quads, sparse);
    a->sp_nodes_full = all_nodes;
}

int rt_accel_pairs(rt_accel* a, bool own, const rt_aabb* nodes, const float4** pairs) {
    bool& valid = own ? a->sp_pairs_valid : a->pairs_valid;
    rt_aabb* out = own ? a->sp_pairs : a->pairs;
    *pairs = (const float4*)out;
    if (valid) return RT_OK;
    // (a build leaves the own hierarchy's nodes incomplete, because nothing on the default path reads them: here they are needed)
    if (own && !a->sp_nodes_full) {
        if (!a->sp_built || !a->sp_sorted) return RT_ERR_INVALID_ARGUMENT;
        own_leaves_and_levels(a, true);
    }
    rt_levels lv;
    if (own) lv = a->sp_lv;
    else if (!rt_make_levels(a->num_prims, &lv)) return RT_ERR_TOO_FEW_PRIMITIVES;
    uint32_t total_pairs = 0u;
    for (uint32_t k = 0; k + 1u < lv.count; k++) total_pairs += (lv.n[k] + 1u) / 2u;
    hipLaunchKernelGGL(k_pairs_from_nodes, dim3(div_up(total_pairs, 256u)), dim3(256), 0, a->dev->stream, lv, total_pairs, nodes, out);
    valid = true;
    return rt_launch_status();
}

int rt_sort_pairs(rt_device* dev, uint2* a, uint2* b, uint32_t n, uint32_t passes, uint2** sorted, bool first_counted,
                  uint32_t first_shift) {
    int rc = rt_sort_workspace(dev, n);
    if (rc) return rc;
    hipStream_t s = dev->stream;
    const uint32_t tiles = div_up(n, kTile);
    uint2* src = a;
    uint2* dst = b;
    for (uint32_t pass = 0; pass < passes; pass++) {
        const uint32_t shift = first_shift + 8u * pass;
        // (first_counted: the kernel that made the keys has left the first pass's per-tile counts in the table)
        if (pass || !first_counted)
            hipLaunchKernelGGL((k_sort_hist<2, false>), dim3(tiles), dim3(256), 0, s, (const uint32_t*)src, (uint32_t*)nullptr, n,
                               shift, dev->sort_table, tiles, (uint4*)nullptr);
        hipLaunchKernelGGL(k_sort_scan, dim3(256), dim3(256), 0, s, dev->sort_table, tiles, dev->sort_totals);
        hipLaunchKernelGGL((k_sort_scatter<2, 2, false>), dim3(tiles), dim3(256), 0, s, (const uint32_t*)src, (uint32_t*)dst, n,
                           shift, dev->sort_table, tiles, dev->sort_totals, (uint4*)nullptr);
        uint2* t = src; src = dst; dst = t;
    }
    *sorted = src;
    return rt_launch_status();
}

extern "C" {

uint32_t rt_bvh_node_count(uint32_t num_primitives) {
    rt_levels lv;
    if (!rt_make_levels(num_primitives, &lv)) return 0u;
    uint32_t total = 0u;
    for (uint32_t k = 0; k < lv.count; k++) total += lv.n[k];
    return total;
}

int rt_generate_morton_codes(rt_device* dev, const rt_morton_code_info* info, const rt_buffer* indices,
                             const rt_buffer* vertices, rt_buffer* morton_codes, rt_buffer* code_frequencies) {
    if (!dev || !info || !indices || !vertices || !morton_codes || !code_frequencies) return RT_ERR_INVALID_ARGUMENT;
    const uint32_t n = info->num_primitives;
    if (indices->bytes < (size_t)n * 12u || morton_codes->bytes < (size_t)n * 16u || code_frequencies->bytes < 256u * 16u)
        return RT_ERR_BUFFER_TOO_SMALL;
    int rc = rt_set_device(dev);
    if (rc) return rc;
    // the shader's "thread < 256" counter reset needs at least one workgroup
    const uint32_t groups = n ? div_up(n, 256u) : 1u;
    hipLaunchKernelGGL(k_morton_ref, dim3(groups), dim3(256), 0, dev->stream, make_bounds(info->scene_min, info->scene_max), n,
                       (const uint32_t*)indices->ptr, (const rt_vertex*)vertices->ptr, (uint4*)morton_codes->ptr,
                       (uint4*)code_frequencies->ptr);
    return rt_launch_status();
}

int rt_sort_count(rt_device* dev, const rt_sort_info* info, const rt_buffer* morton_codes, rt_buffer* temp_morton_codes,
                  rt_buffer* code_frequencies) {
    if (!dev || !info || !morton_codes || !temp_morton_codes || !code_frequencies) return RT_ERR_INVALID_ARGUMENT;
    const uint32_t n = info->num_elements;
    if (info->sort_pass_index > 3u) return RT_ERR_INVALID_ARGUMENT;
    if (morton_codes->bytes < (size_t)n * 16u || temp_morton_codes->bytes < (size_t)n * 16u || code_frequencies->bytes < 4096u)
        return RT_ERR_BUFFER_TOO_SMALL;
    if (!n) return RT_OK;
    int rc = rt_set_device(dev);
    if (rc) return rc;
    rc = rt_sort_workspace(dev, n);
    if (rc) return rc;
    const uint32_t tiles = div_up(n, kTile);
    hipLaunchKernelGGL((k_sort_hist<4, true>), dim3(tiles), dim3(256), 0, dev->stream, (const uint32_t*)morton_codes->ptr,
                       (uint32_t*)temp_morton_codes->ptr, n, 8u * info->sort_pass_index, dev->sort_table, tiles,
                       (uint4*)code_frequencies->ptr);
    return rt_launch_status();
}

int rt_sort_prefix_sum(rt_device* dev, const rt_sort_info* info, rt_buffer* code_frequencies) {
    if (!dev || !info || !code_frequencies) return RT_ERR_INVALID_ARGUMENT;
    if (code_frequencies->bytes < 4096u) return RT_ERR_BUFFER_TOO_SMALL;
    int rc = rt_set_device(dev);
    if (rc) return rc;
    hipLaunchKernelGGL(k_freq_scan, dim3(1), dim3(256), 0, dev->stream, (uint4*)code_frequencies->ptr);
    rc = rt_launch_status();
    if (rc) return rc;
    const uint32_t n = info->num_elements;
    if (!n) return RT_OK;
    rc = rt_sort_workspace(dev, n);
    if (rc) return rc;
    const uint32_t tiles = div_up(n, kTile);
    hipLaunchKernelGGL(k_sort_scan, dim3(256), dim3(256), 0, dev->stream, dev->sort_table, tiles, dev->sort_totals);
    return rt_launch_status();
}

int rt_sort_sort(rt_device* dev, const rt_sort_info* info, rt_buffer* morton_codes, const rt_buffer* temp_morton_codes,
                 rt_buffer* code_frequencies) {
    if (!dev || !info || !morton_codes || !temp_morton_codes || !code_frequencies) return RT_ERR_INVALID_ARGUMENT;
    const uint32_t n = info->num_elements;
    if (info->sort_pass_index > 3u) return RT_ERR_INVALID_ARGUMENT;
    if (morton_codes->bytes < (size_t)n * 16u || temp_morton_codes->bytes < (size_t)n * 16u || code_frequencies->bytes < 4096u)
        return RT_ERR_BUFFER_TOO_SMALL;
    int rc = rt_set_device(dev);
    if (rc) return rc;
    const uint32_t tiles = div_up(n, kTile);
    if (!n || (size_t)tiles * 256u > dev->sort_table_words) {
        if (n) return RT_ERR_INVALID_ARGUMENT;    // scatter without the count step of this pass
        return RT_OK;
    }
    hipLaunchKernelGGL((k_sort_scatter<4, 4, true>), dim3(tiles), dim3(256), 0, dev->stream,
                       (const uint32_t*)temp_morton_codes->ptr, (uint32_t*)morton_codes->ptr, n, 8u * info->sort_pass_index,
                       dev->sort_table, tiles, dev->sort_totals, (uint4*)code_frequencies->ptr);
    return rt_launch_status();
}

int rt_bvh_build_leaves(rt_device* dev, const rt_bvh_info* info, const rt_buffer* indices, const rt_buffer* vertices,
                        const rt_buffer* morton_codes, rt_buffer* bvh) {
    if (!dev || !info || !indices || !vertices || !morton_codes || !bvh) return RT_ERR_INVALID_ARGUMENT;
    const uint32_t n = info->num_children;
    const uint32_t leaves = (n + 1u) / 2u;
    if (morton_codes->bytes < (size_t)n * 16u || bvh->bytes < ((size_t)leaves + 1u) * 32u) return RT_ERR_BUFFER_TOO_SMALL;
    if (!n) return RT_OK;
    int rc = rt_set_device(dev);
    if (rc) return rc;
    rt_levels lv;
    lv.count = 0;
    dev->missed_valid = false;          // a new hierarchy: rays that missed the old scene may hit this one,
    dev->keys_ready = false;            // and ray keys made for the old trunk's box are not this one's
    hipLaunchKernelGGL((k_leaves<kLeavesShader, false>), dim3(div_up(leaves, 256u)), dim3(256), 0, dev->stream, n, lv,
                       (const uint32_t*)morton_codes->ptr, (const uint32_t*)indices->ptr, (const rt_vertex*)vertices->ptr,
                       (const float4*)nullptr, (float4*)nullptr, (rt_aabb*)bvh->ptr, (rt_aabb*)nullptr, (float4*)nullptr,
                       (const uint32_t*)nullptr, 0u, (const float4*)nullptr, 0.0f, (uint2*)nullptr, SceneBounds{}, (const float*)nullptr,
                       (const uint32_t*)nullptr, (uint32_t*)nullptr, 0u);
    return rt_launch_status();
}

int rt_bvh_build(rt_device* dev, const rt_bvh_info* info, rt_buffer* bvh) {
    if (!dev || !info || !bvh) return RT_ERR_INVALID_ARGUMENT;
    const uint32_t parents = (info->num_children + 1u) / 2u;
    if (!parents) return RT_OK;
    const size_t last_read = (size_t)info->previous_index + info->num_children;      // one past the last child
    const size_t last_write = (size_t)info->current_index + parents;
    if (bvh->bytes < last_read * 32u || bvh->bytes < last_write * 32u) return RT_ERR_BUFFER_TOO_SMALL;
    int rc = rt_set_device(dev);
    if (rc) return rc;
    hipLaunchKernelGGL(k_level, dim3(div_up(parents, 256u)), dim3(256), 0, dev->stream, *info, (rt_aabb*)bvh->ptr);
    return rt_launch_status();
}

int rt_accel_create(rt_device* dev, uint32_t max_primitives, rt_accel** out) {
    if (!out) return RT_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (!dev || max_primitives < 3u || max_primitives > 0x2aaaaaaau) return RT_ERR_INVALID_ARGUMENT;
    int rc = rt_set_device(dev);
    if (rc) return rc;
    rc = rt_sort_workspace(dev, max_primitives);
    if (rc) return rc;
    rt_accel* a = new (std::nothrow) rt_accel();
    if (!a) return RT_ERR_OUT_OF_MEMORY;
    a->dev = dev;
    a->max_prims = max_primitives;
    a->num_prims = 0;
    a->tris = a->sorted_tris = a->quads = a->sp_quads = a->sp_tris = nullptr;
    a->keys_a = a->keys_b = nullptr;
    a->pairs = a->sp_pairs = a->sp_nodes = nullptr;
    a->cells = nullptr;
    a->outlier_leaves = nullptr;
    a->pairs_valid = a->sp_pairs_valid = false;
    a->sp_built = false;
    a->sp_nodes_full = false;
    a->built_prims = 0;
    a->sp_sorted = nullptr;
    a->want_spatial = true;
    a->cell_bits_cap = RT_CELL_BITS_MAX;
    const size_t tri_bytes = (size_t)max_primitives * 48u, key_bytes = (size_t)max_primitives * 8u;
    hipError_t e = hipMalloc((void**)&a->tris, tri_bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&a->sorted_tris, ((size_t)max_primitives + 1u) / 2u * RT_LEAF_STRIDE * sizeof(float4));
    if (e == hipSuccess) e = hipMalloc((void**)&a->keys_a, key_bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&a->keys_b, key_bytes);
    {
        rt_levels lv;
        rt_make_levels(max_primitives, &lv);
        // a smaller scene packed later never needs more slots than the largest one
        if (e == hipSuccess) e = hipMalloc((void**)&a->pairs, (rt_aligned_node_slots(&lv) + 64u) * sizeof(rt_aabb));
        if (e == hipSuccess) e = hipMalloc((void**)&a->quads, rt_quad_records(&lv) * RT_QUAD_STRIDE * sizeof(float4));
    }
    // the library's own hierarchy has rt_spatial_slots() leaf slots and all its levels full; beyond what its kernel can
    // address it is not made, and the walk in the reference's order traces such an accel
    if (rt_spatial_slots(max_primitives) <= RT_SPATIAL_SLOTS_MAX) {
        const uint32_t slots = rt_spatial_slots(max_primitives);
        rt_levels sl;
        rt_make_levels(2u * slots, &sl);
        a->sp_lv = sl;
        if (e == hipSuccess) e = hipMalloc((void**)&a->sp_pairs, (rt_aligned_node_slots(&sl) + 64u) * sizeof(rt_aabb));
        if (e == hipSuccess) e = hipMalloc((void**)&a->sp_quads, rt_quad_records(&sl) * RT_QUAD_STRIDE * sizeof(float4));
        if (e == hipSuccess) e = hipMalloc((void**)&a->sp_nodes, (size_t)rt_bvh_node_count(2u * slots) * sizeof(rt_aabb));
        if (e == hipSuccess) e = hipMalloc((void**)&a->sp_tris, (size_t)slots * RT_LEAF_STRIDE * sizeof(float4));
    }
    if (e == hipSuccess) e = hipMalloc((void**)&a->cells, (size_t)RT_CELLS_WORDS * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void**)&a->outlier_leaves, (size_t)(RT_OUTLIERS_MAX / 2u) * RT_LEAF_STRIDE * sizeof(float4));
    if (e != hipSuccess) {
        rt_accel_destroy(a);
        return e == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_HIP_BASE + (int)e;
    }
    *out = a;
    return RT_OK;
}

int rt_accel_destroy(rt_accel* a) {
    if (!a) return RT_ERR_INVALID_ARGUMENT;
    (void)rt_set_device(a->dev);
    (void)hipStreamSynchronize(a->dev->stream);
    if (a->tris) (void)hipFree(a->tris);
    if (a->sorted_tris) (void)hipFree(a->sorted_tris);
    if (a->quads) (void)hipFree(a->quads);
    if (a->keys_a) (void)hipFree(a->keys_a);
    if (a->keys_b) (void)hipFree(a->keys_b);
    if (a->pairs) (void)hipFree(a->pairs);
    if (a->sp_nodes) (void)hipFree(a->sp_nodes);
    if (a->sp_pairs) (void)hipFree(a->sp_pairs);
    if (a->sp_quads) (void)hipFree(a->sp_quads);
    if (a->sp_tris) (void)hipFree(a->sp_tris);
    if (a->cells) (void)hipFree(a->cells);
    if (a->outlier_leaves) (void)hipFree(a->outlier_leaves);
    delete a;
    return RT_OK;
}

int rt_accel_set_second_hierarchy(rt_accel* a, int enable, uint32_t cell_bits) {
    if (!a) return RT_ERR_INVALID_ARGUMENT;
    a->cell_bits_cap = cell_bits > RT_CELL_BITS_MAX ? RT_CELL_BITS_MAX : cell_bits;
    // (an accel beyond RT_SPATIAL_SLOTS_MAX leaf slots was created without room for it: asking for it says so)
    a->want_spatial = enable != 0 && a->sp_tris != nullptr;
    return enable != 0 && !a->sp_tris ? RT_ERR_LIMIT_EXCEEDED : RT_OK;
}

int rt_accel_outliers(rt_accel* a, uint32_t* found, uint32_t* listed) {
    if (!a || !found || !listed) return RT_ERR_INVALID_ARGUMENT;
    *found = *listed = 0u;
    if (!a->sp_built) return RT_OK;                  // the last build made no second hierarchy, hence no list
    int rc = rt_set_device(a->dev);
    if (rc) return rc;
    RT_HIP(hipMemcpyAsync(found, a->cells + RT_CELLS_OUTLIERS, sizeof(uint32_t), hipMemcpyDeviceToHost, a->dev->stream));
    RT_HIP(hipStreamSynchronize(a->dev->stream));
    *listed = *found < RT_OUTLIERS_MAX ? *found : RT_OUTLIERS_MAX;
    return RT_OK;
}

int rt_accel_pack(rt_accel* a, uint32_t num_primitives, const rt_buffer* indices, const rt_buffer* vertices) {
    if (!a || !indices || !vertices) return RT_ERR_INVALID_ARGUMENT;
    if (num_primitives < 3u) return RT_ERR_TOO_FEW_PRIMITIVES;
    if (num_primitives > a->max_prims) return RT_ERR_INVALID_ARGUMENT;
    if (indices->bytes < (size_t)num_primitives * 12u) return RT_ERR_BUFFER_TOO_SMALL;
    int rc = rt_set_device(a->dev);
    if (rc) return rc;
    a->num_prims = num_primitives;
    hipLaunchKernelGGL(k_pack, dim3(div_up(num_primitives, 256u)), dim3(256), 0, a->dev->stream, num_primitives,
                       (const uint32_t*)indices->ptr, (const rt_vertex*)vertices->ptr, a->tris);
    return rt_launch_status();
}

namespace {

int bounds_workspace(rt_device* dev) {
    if (!dev->bounds_partials) RT_HIP(hipMalloc((void**)&dev->bounds_partials, kBoundsBlocks * 6u * sizeof(float)));
    return RT_OK;
}

// centroid bounds of n triangles into bounds[0..7]; tris != nullptr also writes the packed triangle array.
// partials: kBoundsBlocks x 6 floats of scratch; nullptr = the device's own workspace, allocated on first use
int launch_bounds(rt_device* dev, uint32_t n, const rt_buffer* indices, const rt_buffer* vertices, float4* tris, rt_buffer* bounds,
                  float* partials = nullptr) {
    if (indices->bytes < (size_t)n * 12u || bounds->bytes < 32u) return RT_ERR_BUFFER_TOO_SMALL;
    int rc = rt_set_device(dev);
    if (rc) return rc;
    if (!partials) {
        rc = bounds_workspace(dev);
        if (rc) return rc;
        partials = dev->bounds_partials;
    }
    uint32_t blocks = div_up(n, 256u);
    if (blocks > kBoundsBlocks) blocks = kBoundsBlocks;
    if (blocks) {
        if (tris)
            hipLaunchKernelGGL((k_bounds_partial<true>), dim3(blocks), dim3(256), 0, dev->stream, n, (const uint32_t*)indices->ptr,
                               (const rt_vertex*)vertices->ptr, tris, partials);
        else
            hipLaunchKernelGGL((k_bounds_partial<false>), dim3(blocks), dim3(256), 0, dev->stream, n, (const uint32_t*)indices->ptr,
                               (const rt_vertex*)vertices->ptr, (float4*)nullptr, partials);
    }
    hipLaunchKernelGGL(k_bounds_finish, dim3(1), dim3(256), 0, dev->stream, blocks, (const float*)partials, (float*)bounds->ptr);
    return rt_launch_status();
}

int accel_build(rt_accel* a, const float* scene_min, const float* scene_max, const float* dbounds, rt_buffer* bvh,
                rt_buffer* morton_codes);

}  // namespace

int rt_scene_bounds(rt_device* dev, uint32_t num_primitives, const rt_buffer* indices, const rt_buffer* vertices,
                    rt_buffer* bounds) {
    if (!dev || !indices || !vertices || !bounds) return RT_ERR_INVALID_ARGUMENT;
    if (num_primitives > 0x55555555u) return RT_ERR_INVALID_ARGUMENT;
    return launch_bounds(dev, num_primitives, indices, vertices, nullptr, bounds);
}

int rt_scene_bounds_static(rt_device* dev, uint32_t num_primitives, const rt_buffer* indices, const rt_buffer* vertices,
                           rt_buffer* bounds) {
    if (!dev || !indices || !vertices || !bounds) return RT_ERR_INVALID_ARGUMENT;
    if (num_primitives > RT_SCENE_BOUNDS_STATIC_MAX_PRIMITIVES) return RT_ERR_LIMIT_EXCEEDED;
    // the address of this device's copy of the scratch array, looked up when the device was created
    float* partials = (dev->gpu >= 0 && dev->gpu < kStaticScratchDevices) ? g_scratch_ptr[dev->gpu].load(std::memory_order_acquire) : nullptr;
    if (!partials) return RT_ERR_LIMIT_EXCEEDED;                      // GPU index beyond the table of scratch addresses
    // Nothing has to be cleared: the first launch overwrites partials [0, blocks) and the second reads exactly those,
    // whatever an earlier (larger) call left behind them.
    return launch_bounds(dev, num_primitives, indices, vertices, nullptr, bounds, partials);
}

int rt_accel_pack_bounds(rt_accel* a, uint32_t num_primitives, const rt_buffer* indices, const rt_buffer* vertices,
                         rt_buffer* bounds) {
    if (!a || !indices || !vertices || !bounds) return RT_ERR_INVALID_ARGUMENT;
    if (num_primitives < 3u) return RT_ERR_TOO_FEW_PRIMITIVES;
    if (num_primitives > a->max_prims) return RT_ERR_INVALID_ARGUMENT;
    int rc = launch_bounds(a->dev, num_primitives, indices, vertices, a->tris, bounds);
    if (rc) return rc;
    a->num_prims = num_primitives;
    return RT_OK;
}

int rt_synthetic_animate(rt_device* dev, uint32_t num_primitives, uint32_t frame, const rt_buffer* base_vertices,
                         rt_buffer* vertices) {
    if (!dev || !base_vertices || !vertices) return RT_ERR_INVALID_ARGUMENT;
    if (num_primitives > 0x55555555u) return RT_ERR_INVALID_ARGUMENT;
    if (base_vertices->bytes < (size_t)num_primitives * 144u || vertices->bytes < (size_t)num_primitives * 144u)
        return RT_ERR_BUFFER_TOO_SMALL;
    if (!num_primitives) return RT_OK;
    if (((uintptr_t)base_vertices->ptr | (uintptr_t)vertices->ptr) & 15u) return RT_ERR_INVALID_ARGUMENT;      // moved 16 bytes at a time
    int rc = rt_set_device(dev);
    if (rc) return rc;
    static_assert(sizeof(rt_vertex) == 48u, "three 16-byte parts per vertex");
    const uint64_t parts = (uint64_t)num_primitives * 9u;       // < 2^32 * 3: the grid's x dimension holds the workgroups
    hipLaunchKernelGGL(k_animate, dim3((uint32_t)((parts + 255u) / 256u)), dim3(256), 0, dev->stream, num_primitives, frame,
                       (const float4*)base_vertices->ptr, (float4*)vertices->ptr);
    return rt_launch_status();
}

int rt_accel_build(rt_accel* a, const float scene_min[3], const float scene_max[3], rt_buffer* bvh, rt_buffer* morton_codes) {
    if (!a || !scene_min || !scene_max || !bvh) return RT_ERR_INVALID_ARGUMENT;
    return accel_build(a, scene_min, scene_max, nullptr, bvh, morton_codes);
}

int rt_accel_build_device_bounds(rt_accel* a, const rt_buffer* bounds, rt_buffer* bvh, rt_buffer* morton_codes) {
    if (!a || !bounds || !bvh) return RT_ERR_INVALID_ARGUMENT;
    if (bounds->bytes < 32u) return RT_ERR_BUFFER_TOO_SMALL;
    const float zero[3] = {0.0f, 0.0f, 0.0f};
    return accel_build(a, zero, zero, (const float*)bounds->ptr, bvh, morton_codes);
}

int rt_accel_refit(rt_accel* a, rt_buffer* bvh) {
    if (!a || !bvh) return RT_ERR_INVALID_ARGUMENT;
    const uint32_t n = a->num_prims;
    if (!a->built_prims || a->built_prims != n) return RT_ERR_INVALID_ARGUMENT;       // nothing of this size to refit
    rt_levels lv;
    if (!rt_make_levels(n, &lv)) return RT_ERR_TOO_FEW_PRIMITIVES;
    if (bvh->bytes < (size_t)rt_bvh_node_count(n) * 32u) return RT_ERR_BUFFER_TOO_SMALL;
    rt_device* dev = a->dev;
    int rc = rt_set_device(dev);
    if (rc) return rc;
    hipStream_t s = dev->stream;
    dev->missed_valid = false;
    dev->keys_ready = false;
    const bool own = a->sp_built && a->sp_sorted != nullptr;
    // the order of the last build, read back from its leaf records into the key buffer that is free
    uint2* order = (own && a->sp_sorted == a->keys_b) ? a->keys_a : a->keys_b;
    const uint32_t leaves = lv.n[0];
    const SceneBounds none{};           // (scene bounds only enter the keys, which a refit does not make)
    hipLaunchKernelGGL(k_order_from_leaves, dim3(div_up(leaves, 256u)), dim3(256), 0, s, n, (const float4*)a->sorted_tris, order);
    a->pairs_valid = a->sp_pairs_valid = false;
    hipLaunchKernelGGL((k_leaves<kLeavesPacked, true>), dim3(div_up(leaves, 256u)), dim3(256), 0, s, n, lv, (const uint32_t*)order,
                       (const uint32_t*)nullptr, (const rt_vertex*)nullptr, a->tris, a->sorted_tris, (rt_aabb*)bvh->ptr, (rt_aabb*)nullptr,
                       a->quads, (const uint32_t*)nullptr, 0u, (const float4*)nullptr, 0.0f, (uint2*)nullptr, none, (const float*)nullptr,
                       (const uint32_t*)nullptr, (uint32_t*)nullptr, 0u);
    for (uint32_t k0 = 8u; k0 + 1u < lv.count; k0 += 8u)
        hipLaunchKernelGGL(k_upsweep, dim3(div_up(lv.n[k0], 256u)), dim3(256), 0, s, k0, lv, (rt_aabb*)bvh->ptr, (rt_aabb*)nullptr, a->quads, 0u);
    if (!own) return rt_launch_status();
    // the same outliers as at the last build, from the refitted leaf records
    hipLaunchKernelGGL(k_outlier_leaves, dim3(1), dim3(64), 0, s, (const uint32_t*)a->cells, (const float4*)a->sorted_tris, a->outlier_leaves);
    // the library's own hierarchy: same slots, same pairs of triangles, corners from the leaf records written above
    own_leaves_and_levels(a, false);
    return rt_launch_status();
}

namespace {

int accel_build(rt_accel* a, const float* scene_min, const float* scene_max, const float* dbounds, rt_buffer* bvh,
                rt_buffer* morton_codes) {
    const uint32_t n = a->num_prims;
    rt_levels lv;
    if (!rt_make_levels(n, &lv)) return RT_ERR_TOO_FEW_PRIMITIVES;
    if (bvh->bytes < (size_t)rt_bvh_node_count(n) * 32u) return RT_ERR_BUFFER_TOO_SMALL;
    if (morton_codes && morton_codes->bytes < (size_t)n * 16u) return RT_ERR_BUFFER_TOO_SMALL;
    rt_device* dev = a->dev;
    int rc = rt_set_device(dev);
    if (rc) return rc;
    hipStream_t s = dev->stream;
    dev->missed_valid = false;          // a new hierarchy: rays that missed the old scene may hit this one,
    dev->keys_ready = false;            // and ray keys made for the old trunk's box are not this one's

    rc = rt_sort_workspace(dev, n);
    if (rc) return rc;
    const uint32_t tiles = div_up(n, kTile);
    hipLaunchKernelGGL(k_morton_hist, dim3(tiles), dim3(256), 0, s, make_bounds(scene_min, scene_max), dbounds, n, a->tris, a->keys_a,
                       dev->sort_table, tiles);
    // The library's own hierarchy (rt_accel_set_second_hierarchy): same triangles, same kernels, keys without overlapping
    // bits. Its keys' coordinates come from per-axis histograms of the centroids, which only need the triangles:
    const bool own = a->want_spatial && a->sp_tris != nullptr;
    const SceneBounds spb = make_bounds(scene_min, scene_max);
    if (own) {
        // (the per-axis histograms and, right behind them, the count of outliers)
        static_assert(RT_CELLS_OUTLIERS == RT_CELLS_AXIS + 3u * RT_AXIS_STEPS, "one memset clears both");
        RT_HIP(hipMemsetAsync(a->cells + RT_CELLS_AXIS, 0, (size_t)(3u * RT_AXIS_STEPS + 1u) * sizeof(uint32_t), s));
        const uint32_t stride = n > (1u << 16) ? (n >> 16) + 1u : 1u;         // 65 536 samples: 64 per step of the 1024 coordinates made
        const uint32_t samples = div_up(n, stride);
        hipLaunchKernelGGL(k_axis_hist, dim3(min(div_up(samples, 256u), 512u)), dim3(256), 0, s, spb, dbounds, n, stride, a->tris, a->cells);
        hipLaunchKernelGGL(k_axis_map, dim3(3), dim3(1024), 0, s, samples, a->cells);
    }
    uint2* src = nullptr;
    rc = rt_sort_pairs(dev, a->keys_a, a->keys_b, n, 4u, &src, true);     // four passes: the sorted pairs are back in keys_a
    if (rc) return rc;
    uint2* other = src == a->keys_a ? a->keys_b : a->keys_a;
    const uint32_t leaves = lv.n[0];
    // the pair records are not written here: the traversal kernels that read them have them made on demand
    a->pairs_valid = a->sp_pairs_valid = false;
    hipLaunchKernelGGL((k_leaves<kLeavesPacked, true>), dim3(div_up(leaves, 256u)), dim3(256), 0, s, n, lv, (const uint32_t*)src,
                       (const uint32_t*)nullptr, (const rt_vertex*)nullptr, a->tris, a->sorted_tris, (rt_aabb*)bvh->ptr, (rt_aabb*)nullptr,
                       a->quads, (const uint32_t*)nullptr, 0u, (const float4*)nullptr, 0.0f, own ? other : (uint2*)nullptr, spb, dbounds,
                       (const uint32_t*)(a->cells + RT_CELLS_AXIS), a->cells + RT_CELLS_OUTLIERS, 0u);
    if (own) hipLaunchKernelGGL(k_outlier_leaves, dim3(1), dim3(64), 0, s, (const uint32_t*)a->cells, (const float4*)a->sorted_tris, a->outlier_leaves);
    for (uint32_t k0 = 8u; k0 + 1u < lv.count; k0 += 8u)
        hipLaunchKernelGGL(k_upsweep, dim3(div_up(lv.n[k0], 256u)), dim3(256), 0, s, k0, lv, (rt_aabb*)bvh->ptr, (rt_aabb*)nullptr, a->quads, 0u);
    if (morton_codes)
        hipLaunchKernelGGL(k_expand_keys, dim3(div_up(n, 256u)), dim3(256), 0, s, n, src, (uint4*)morton_codes->ptr);

    a->sp_built = own;
    a->built_prims = n;
    a->sp_sorted = nullptr;
    if (!own) return rt_launch_status();
    // (key, place in the reference's order) of every triangle, written by the leaf kernel above, sorted by key
    uint2* sp_src = nullptr;
    // (by the upper 24 of the key's 32 bits: the lowest eight tell triangles apart that share a cell of 2^-22 of the
    // curve, far below the cells the leaf slots are laid out by, and the pairing inside a record looks at the boxes anyway)
    rc = rt_sort_pairs(dev, other, src, n, 3u, &sp_src, false, 8u);
    if (rc) return rc;
    a->sp_sorted = sp_src;
    // leaf slots by Morton cell where the cells fit (k_cell_choose), then leaves and levels over all the slots
    const uint32_t slots = rt_spatial_slots(n);
    uint32_t slots_log2 = 0u;
    while ((1u << slots_log2) < slots) slots_log2++;
    rt_make_levels(2u * slots, &a->sp_lv);
    RT_HIP(hipMemsetAsync(a->cells, 0, (size_t)RT_CELLS_START * sizeof(uint32_t), s));
    hipLaunchKernelGGL(k_cell_hist, dim3(div_up(n, 256u)), dim3(256), 0, s, n, (const uint2*)sp_src, a->cells);
    hipLaunchKernelGGL(k_cell_sums, dim3((1u << RT_CELL_BITS_MAX) / 512u), dim3(512), 0, s, a->cells);
    hipLaunchKernelGGL(k_cell_choose, dim3(1), dim3(1024), 0, s, a->cells, slots_log2, a->cell_bits_cap);
    hipLaunchKernelGGL(k_cell_starts, dim3(div_up(n, 256u)), dim3(256), 0, s, n, (const uint2*)sp_src, a->cells);
    hipLaunchKernelGGL(k_cell_fill, dim3(256), dim3(256), 0, s, a->cells);
    own_leaves_and_levels(a, false);
    return rt_launch_status();
}

}  // namespace

}  // extern "C"
