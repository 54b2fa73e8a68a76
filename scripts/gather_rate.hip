// Ceiling of the memory system for the traversal kernel's access pattern: every lane follows a dependent chain
// of random 64-byte records and reads 48 bytes (three 16-byte loads) of each, as one step of the tree walk does.
// Swept over the table size (L1 / L2 / Infinity Cache / HBM resident), the workgroups per CU and the share of
// active lanes; variants read the record with one, two or three loads to separate per-access from per-line cost.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/gather_rate scripts/gather_rate.hip && scripts/gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int LOADS>
__global__ __launch_bounds__(256) void k_gather(const float4* __restrict__ t, uint32_t records, uint32_t per_lane,
                                                uint32_t active_mask, float* out) {
    if (!((active_mask >> (threadIdx.x & 31u)) & 1u)) return;
    uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.0f;
    for (uint32_t j = 0; j < per_lane; j++) {
        s ^= s << 13; s ^= s >> 17; s ^= s << 5;
        const float4* r = t + 4 * (size_t)(s & (records - 1u));
        float4 a = r[0];
        acc += a.x;
        if (LOADS > 1) { float4 b = r[1]; acc += b.y; }
        if (LOADS > 2) { float4 c = r[2]; acc += c.z; }
        s += __float_as_uint(acc) & 1u;        // the next address depends on the data, as in a tree walk
    }
    if (acc == 1.2345f) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_gather128(const float4* __restrict__ t, uint32_t records, uint32_t per_lane, float* out) {
    uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.0f;
    for (uint32_t j = 0; j < per_lane; j++) {
        s ^= s << 13; s ^= s >> 17; s ^= s << 5;
        const float4* r = t + 8 * (size_t)(s & (records - 1u));
        float4 a = r[0], b = r[4];
        acc += a.x + b.y;
        s += __float_as_uint(acc) & 1u;
    }
    if (acc == 1.2345f) out[0] = acc;
}

// What a 16-byte load costs when the four lanes of a quad read 64 contiguous bytes of one record (each quad its
// own random record) against every lane reading 16 bytes of its own record: the address/tag rate of the L1 path.
// ... and when all 64 lanes of a wave (SHARE = 6), or groups of 2^SHARE neighbouring lanes, read the very same 16 bytes
template <int SHARE>
__global__ __launch_bounds__(256) void k_same(const float4* __restrict__ t, uint32_t records, uint32_t per_lane, float* out) {
    const uint32_t id = blockIdx.x * 256u + threadIdx.x;
    uint32_t s = (id >> SHARE) * 2654435761u + 12345u;
    float acc = 0.0f;
    for (uint32_t j = 0; j < per_lane; j++) {
        s ^= s << 13; s ^= s >> 17; s ^= s << 5;
        const float4* r = t + 4 * (size_t)(s & (records - 1u));
        float4 a = r[0];
        acc += a.x;
        s += __float_as_uint(acc) & 1u;
    }
    if (acc == 1.2345f) out[0] = acc;
}

template <bool QUAD>
__global__ __launch_bounds__(256) void k_access(const float4* __restrict__ t, uint32_t records, uint32_t per_lane, float* out) {
    const uint32_t id = blockIdx.x * 256u + threadIdx.x;
    uint32_t s = (QUAD ? (id >> 2) : id) * 2654435761u + 12345u;
    float acc = 0.0f;
    for (uint32_t j = 0; j < per_lane; j++) {
        s ^= s << 13; s ^= s >> 17; s ^= s << 5;
        const float4* r = t + 4 * (size_t)(s & (records - 1u)) + (QUAD ? (id & 3u) : 0u);
        float4 a = r[0];
        acc += a.x;
        s += __float_as_uint(acc) & 1u;
    }
    if (acc == 1.2345f) out[0] = acc;
}

int main() {
    const size_t big = (size_t)1 << 30;
    float4* t; float* out;
    hipMalloc(&t, big); hipMalloc(&out, 4);
    hipMemset(t, 0, big);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const size_t sizes[] = {(size_t)16 << 10, (size_t)1 << 20, (size_t)16 << 20, (size_t)64 << 20, big};
    const uint32_t masks[] = {0xffffffffu};
    for (size_t sz : sizes)
        for (int bpc : {3})
            for (uint32_t mask : masks)
                for (int loads = 1; loads <= 3; loads += 2) {
                    const uint32_t records = (uint32_t)(sz / 64), per_lane = 1024, blocks = 256u * bpc;
                    float ms = 0.0f;
                    for (int rep = 0; rep < 2; rep++) {
                        hipEventRecord(a);
                        if (loads == 1) hipLaunchKernelGGL(k_gather<1>, dim3(blocks), dim3(256), 0, 0, t, records, per_lane, mask, out);
                        if (loads == 2) hipLaunchKernelGGL(k_gather<2>, dim3(blocks), dim3(256), 0, 0, t, records, per_lane, mask, out);
                        if (loads == 3) hipLaunchKernelGGL(k_gather<3>, dim3(blocks), dim3(256), 0, 0, t, records, per_lane, mask, out);
                        hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
                    }
                    const double n = (double)blocks * 256 * per_lane * (mask == 0xffffffffu ? 1.0 : 0.5);
                    printf("table %7zu KiB  %d blocks/CU  lanes %3d%%  %d loads/record: %7.3f ms  %6.1f G records/s\n", sz >> 10, bpc,
                           mask == 0xffffffffu ? 100 : 50, loads, ms, n / ms / 1e6);
                }
    // Is the rate past L2 limited per CU or for the whole chip? One workgroup per CU on a growing number of CUs,
    // and records of 128 bytes (both 64-byte halves of a line read) against 64.
    for (size_t sz : {(size_t)64 << 20, big})
        for (uint32_t blocks : {32u, 64u, 128u, 256u, 512u, 1024u}) {
            const uint32_t records = (uint32_t)(sz / 64), per_lane = 1024;
            float ms = 0.0f;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(a);
                hipLaunchKernelGGL(k_gather<3>, dim3(blocks), dim3(256), 0, 0, t, records, per_lane, 0xffffffffu, out);
                hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
            }
            const double n = (double)blocks * 256 * per_lane;
            printf("table %7zu KiB  %4u workgroups in all: %7.3f ms  %6.1f G records/s\n", sz >> 10, blocks, ms, n / ms / 1e6);
        }
    for (size_t sz : {(size_t)16 << 10, (size_t)1 << 20, (size_t)64 << 20})
        for (int quad = 0; quad < 2; quad++) {
            const uint32_t records = (uint32_t)(sz / 64), per_lane = 1024, blocks = 256u * 3u;
            float ms = 0.0f;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(a);
                if (quad) hipLaunchKernelGGL(k_access<true>, dim3(blocks), dim3(256), 0, 0, t, records, per_lane, out);
                else hipLaunchKernelGGL(k_access<false>, dim3(blocks), dim3(256), 0, 0, t, records, per_lane, out);
                hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
            }
            const double n = (double)blocks * 256 * per_lane;
            printf("table %7zu KiB  16-byte loads, %s: %7.3f ms  %6.1f G loads/s  %.2f per clock and CU\n", sz >> 10,
                   quad ? "four lanes on one record's 64 bytes" : "every lane on its own record     ", ms, n / ms / 1e6,
                   n / ms / 1e6 / 256 / 2.4);
        }
    for (int share : {1, 2, 4, 6}) {
        const uint32_t records = (16u << 10) / 64, per_lane = 1024, blocks = 256u * 3u;
        float ms = 0.0f;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(a);
            if (share == 1) hipLaunchKernelGGL(k_same<1>, dim3(blocks), dim3(256), 0, 0, t, records, per_lane, out);
            if (share == 2) hipLaunchKernelGGL(k_same<2>, dim3(blocks), dim3(256), 0, 0, t, records, per_lane, out);
            if (share == 4) hipLaunchKernelGGL(k_same<4>, dim3(blocks), dim3(256), 0, 0, t, records, per_lane, out);
            if (share == 6) hipLaunchKernelGGL(k_same<6>, dim3(blocks), dim3(256), 0, 0, t, records, per_lane, out);
            hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        }
        const double n = (double)blocks * 256 * per_lane;
        printf("table      16 KiB  16-byte loads, %2d neighbouring lanes on the same address: %7.3f ms  %6.1f G loads/s  %.2f per clock and CU\n",
               1 << share, ms, n / ms / 1e6, n / ms / 1e6 / 256 / 2.4);
    }
    for (size_t sz : {(size_t)1 << 20, (size_t)64 << 20, big})
        for (int bpc : {2, 4}) {
            const uint32_t records = (uint32_t)(sz / 128), per_lane = 1024, blocks = 256u * bpc;
            float ms = 0.0f;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(a);
                hipLaunchKernelGGL(k_gather128, dim3(blocks), dim3(256), 0, 0, t, records, per_lane, out);
                hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
            }
            const double n = (double)blocks * 256 * per_lane;
            printf("table %7zu KiB  %d blocks/CU  128-byte records (both halves read): %7.3f ms  %6.1f G records/s  %5.2f TB/s\n",
                   sz >> 10, bpc, ms, n / ms / 1e6, n * 128 / ms / 1e9);
        }
    return 0;
}
