// Calibration of the FETCH_SIZE counter and of the memory system's rate for the traversal kernel's access
// pattern: every lane reads 48 bytes (three 16-byte loads) of a random 64-byte record of a table.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/calib_fetch scripts/calib_fetch.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- scripts/calib_fetch
// Prints, per case, the records read, the bytes they hold (x48 used, x64 per record) and the time; compare
// the kernel's FETCH_SIZE with it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ __launch_bounds__(256) void k_stream(const float4* __restrict__ t, size_t n, float* out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float acc = 0.0f;
    for (; i < n; i += (size_t)gridDim.x * 256) { float4 v = t[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 1.2345f) out[0] = acc;
}

// records: number of 64-byte records in the table (power of two); per_lane: gathers per lane
__global__ __launch_bounds__(256) void k_gather(const float4* __restrict__ t, uint32_t records, uint32_t per_lane, float* out) {
    uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.0f;
    for (uint32_t j = 0; j < per_lane; j++) {
        s ^= s << 13; s ^= s >> 17; s ^= s << 5;
        const float4* r = t + 4 * (size_t)(s & (records - 1u));
        float4 a = r[0], b = r[1], c = r[2];
        acc += a.x + b.y + c.z;
        s += __float_as_uint(acc) & 1u;        // the next address depends on the data, as in a tree walk
    }
    if (acc == 1.2345f) out[0] = acc;
}

int main() {
    const size_t big = (size_t)2 << 30;         // 2 GiB: beyond the 256 MiB Infinity Cache
    float4* t; float* out;
    hipMalloc(&t, big); hipMalloc(&out, 4);
    hipMemset(t, 0, big);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms;
    // 1. streaming read of 1 GiB
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k_stream, dim3(256 * 16), dim3(256), 0, 0, t, ((size_t)1 << 30) / 16, out);
        hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        printf("stream   1 GiB: %.3f ms  %.2f TB/s\n", ms, 1.073741824 / ms);
    }
    // 2. dependent random 48-of-64-byte gathers, tables of 64 MiB (Infinity Cache) and 2 GiB (HBM), 12 waves per CU
    const size_t sizes[2] = {(size_t)64 << 20, big};
    for (int c = 0; c < 2; c++) {
        const uint32_t records = (uint32_t)(sizes[c] / 64);
        const uint32_t per_lane = 2048;
        const uint32_t blocks = 256 * 3;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(a);
            hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, 0, t, records, per_lane, out);
            hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
            const double n = (double)blocks * 256 * per_lane;
            printf("gather table %5zu MiB: %.0f records (%.1f GB at 64 B), %.3f ms, %.2f G records/s, %.2f TB/s at 64 B\n",
                   sizes[c] >> 20, n, n * 64 / 1e9, ms, n / ms / 1e6, n * 64 / ms / 1e9);
        }
    }
    return 0;
}
