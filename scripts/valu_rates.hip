// Issue cost of the vector instructions the traversal step is made of (gfx950): cycles per wave64 instruction, measured with
// one wave per SIMD running 8 independent dependency chains. Build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHAINS 8
#define ITERS 4096

template <int OP>
__global__ __launch_bounds__(256) void k(uint64_t* out, uint32_t seed) {
    uint64_t a[CHAINS];
    uint32_t s = seed + threadIdx.x;
    for (int i = 0; i < CHAINS; i++) a[i] = ((uint64_t)(s * 2654435761u + i) << 32) | (s + i * 7u);
    const uint32_t sh = (seed & 3u) + 1u;      // run-time shift amount
    const long long t0 = clock64();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < CHAINS; i++) {
            uint32_t lo = (uint32_t)a[i], hi = (uint32_t)(a[i] >> 32);
            if (OP == 0) { lo = lo * 1u + hi; }                                            // v_add_u32 (baseline, may fold)
            if (OP == 1) { a[i] = a[i] << sh; continue; }                                   // v_lshlrev_b64
            if (OP == 2) { a[i] = (a[i] >> sh) | (1ull << 63); continue; }                  // v_lshrrev_b64 (+ or)
            if (OP == 3) { a[i] = a[i] & (a[i] - 1ull) | (1ull << 40); continue; }         // 64-bit add + ands
            if (OP == 4) { lo = __builtin_amdgcn_alignbit(hi, lo, sh); }                    // v_alignbit_b32
            if (OP == 5) { lo = (uint32_t)__builtin_ctz(lo | 0x80000000u) + hi; }           // v_ffbl_b32
            if (OP == 6) { lo = (lo << sh) | hi; }                                          // v_lshl_or_b32
            if (OP == 7) { lo = lo > hi ? lo + 1u : hi; }                                   // cmp + cndmask
            if (OP == 8) { float f = __uint_as_float(lo), g = __uint_as_float(hi); lo = __float_as_uint(fminf(fminf(f, g), 1.0f)); }
            if (OP == 9) { a[i] = a[i] + 0x100000001ull; continue; }                        // 64-bit add
            a[i] = ((uint64_t)hi << 32) | lo;
        }
    }
    const long long t1 = clock64();
    uint64_t r = 0;
    for (int i = 0; i < CHAINS; i++) r ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (uint64_t)(t1 - t0);
}

template <int OP>
void run(const char* name, uint64_t* d) {
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256), 0, 0, d, 12345u);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256), 0, 0, d, 12345u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    uint64_t ticks = 0;
    hipMemcpy(&ticks, d + (1 << 20), 8, hipMemcpyDeviceToHost);
    printf("%-28s %8.3f ms  %6.2f clock64 ticks per op per wave  (%.2f ns per op)\n", name, ms, (double)ticks / (ITERS * CHAINS), ms * 1e6 / (ITERS * CHAINS));
}

int main() {
    uint64_t* d;
    hipMalloc(&d, ((1 << 20) + 8) * 8);
    run<0>("v_add_u32", d);
    run<1>("v_lshlrev_b64", d);
    run<2>("v_lshrrev_b64 + or", d);
    run<3>("x & (x - 1) 64-bit", d);
    run<4>("v_alignbit_b32", d);
    run<5>("v_ffbl_b32 + add", d);
    run<6>("v_lshl_or_b32", d);
    run<7>("cmp + cndmask + add", d);
    run<8>("v_min3_f32", d);
    run<9>("64-bit add", d);
    return 0;
}
