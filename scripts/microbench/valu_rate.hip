// Issue rate of the integer instructions Philox4x32 is made of, on one MI355X: N dependent-free chains per lane, enough wavefronts to
// fill every SIMD.  hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed, int iters) {
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
    uint32_t b0 = a0 ^ 0x9E3779B9u, b1 = a1 ^ 0xBB67AE85u;
    for (int i = 0; i < iters; ++i) {
#define STEP(x)                                                                                                                      \
        if (OP == 0) { uint64_t p = (uint64_t)0xD2511F53u * x + b0; x = (uint32_t)p ^ (uint32_t)(p >> 32); }       /* v_mad_u64_u32 + xor */ \
        else if (OP == 1) { x = (x + b0) ^ b1; }                                                                     /* add + xor */   \
        else if (OP == 2) { x = x * 0xCD9E8D57u + b1; }                                                              /* v_mul_lo_u32 (+add) */ \
        else if (OP == 3) { x = __umulhi(x, 0xCD9E8D57u) ^ b1; }                                                     /* v_mul_hi_u32 + xor */ \
        else { x = (x & 0xffffffu) * (b1 & 0xffffffu) + b0; }                                                        /* v_mad_u32_u24 */
        STEP(a0) STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

template <int OP>
static void run(const char *name, uint32_t *d) {
    const int blocks = 256 * 8 * 4, iters = 4096;          // 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 1, 16);
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d, 1, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double wave_steps = (double)blocks * 4 * iters * 8;          // wavefront-level STEPs
    printf("%-28s %8.3f ms   %.2f ns per wavefront step per SIMD (1024 SIMDs)\n", name, ms, ms * 1e6 / (wave_steps / 1024.0));
}

int main() {
    uint32_t *d; hipMalloc(&d, 256 * 8 * 4 * 256 * 4);
    run<1>("add + xor", d);
    run<0>("mad_u64_u32 + xor", d);
    run<2>("mul_lo_u32 + add", d);
    run<3>("mul_hi_u32 + xor", d);
    run<4>("mad_u32_u24 (+2 and)", d);
    return 0;
}
// one MI355X (round 2):  add + xor 2.29 ns per step (two instructions), mad_u64_u32 + xor 4.07, mul_lo_u32 + add 2.10, mul_hi_u32 + xor 3.17,
// and + mad_u32_u24 1.90  ->  ~1.1 ns per wavefront instruction and SIMD; v_mad_u64_u32 ~2.6 x, v_mul_hi_u32 ~1.8 x, v_mul_lo_u32 1 x
