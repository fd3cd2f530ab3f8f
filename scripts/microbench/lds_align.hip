// LDS cost of the access shapes the text-assembling kernels use (k_errlog: rows of ~67 bytes side by side, any byte alignment), on one
// MI355X: every wavefront of a full machine issues the same DS instruction `iters` times on its own 16 KB of LDS.
//   hipcc --offload-arch=gfx950 -O3 -o lds_align lds_align.hip && ./lds_align
// (round 5: k_errlog's LDS pipe was 76 % busy with 19 LDS cycles per DS instruction against 5.5 in the record kernel — which shape costs that?)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

enum { W64_ALIGNED8 = 0, W64_ROW67, W64_ROW72, W8_ROW67, W32_ROW68, R64_STRIDE16_OFF3, R64_STRIDE16_OFF0, R128_STRIDE16, W128_STRIDE16, W64_ROW67_OFF4,
       W32_ROW67_UNALIGNED, R64_BROADCAST, W16_ROW67, N_OPS };
static const char *NAMES[N_OPS] = {"ds_write_b64 lane*8 (aligned)", "ds_write_b64 lane*67 (rows, any alignment)", "ds_write_b64 lane*72 (rows, 8-aligned)",
                                   "ds_write_b8  lane*67", "ds_write_b32 lane*68 (rows, 4-aligned)", "ds_read_b64 lane*16+3 (unaligned)",
                                   "ds_read_b64 lane*16 (aligned)", "ds_read_b128 lane*16", "ds_write_b128 lane*16", "ds_write_b64 lane*68+4 (4-aligned only)",
                                   "ds_write_b32 lane*67 (unaligned)", "ds_read_b64 same address (broadcast)", "ds_write_b16 lane*67 (odd addresses too)"};

template <int OP>
__global__ void __launch_bounds__(64) k(uint32_t *out, int iters) {
    __shared__ __align__(16) uint8_t lds[16384];
    const uint32_t lane = threadIdx.x;
    uint32_t base = (uint32_t)(uintptr_t)lds;          // LDS byte address
    uint32_t a;
    switch (OP) {
        case W64_ALIGNED8: a = lane * 8; break;
        case W64_ROW67: case W8_ROW67: case W32_ROW67_UNALIGNED: case W16_ROW67: a = lane * 67; break;
        case W64_ROW72: a = lane * 72; break;
        case W32_ROW68: a = lane * 68; break;
        case R64_STRIDE16_OFF3: a = lane * 16 + 3; break;
        case W64_ROW67_OFF4: a = lane * 68 + 4; break;
        case R64_BROADCAST: a = 64; break;
        default: a = lane * 16; break;
    }
    a += base;
    uint64_t v = lane * 0x0101010101010101ull, acc = 0;
    uint32_t v32 = lane;
    for (int i = 0; i < iters; ++i) {
        const uint32_t ai = a + ((i & 7) << 3) * (OP == R64_BROADCAST ? 0 : 1) * 0;      // (same address every iteration)
        if (OP == W64_ALIGNED8 || OP == W64_ROW67 || OP == W64_ROW72 || OP == W64_ROW67_OFF4) asm volatile("ds_write_b64 %0, %1" ::"v"(ai), "v"(v) : "memory");
        else if (OP == W8_ROW67) asm volatile("ds_write_b8 %0, %1" ::"v"(ai), "v"(v32) : "memory");
        else if (OP == W16_ROW67) asm volatile("ds_write_b16 %0, %1" ::"v"(ai), "v"(v32) : "memory");
        else if (OP == W32_ROW68 || OP == W32_ROW67_UNALIGNED) asm volatile("ds_write_b32 %0, %1" ::"v"(ai), "v"(v32) : "memory");
        else if (OP == R64_STRIDE16_OFF3 || OP == R64_STRIDE16_OFF0 || OP == R64_BROADCAST) { uint64_t r; asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(ai) : "memory"); acc ^= r; }
        else if (OP == R128_STRIDE16) { v4u r; asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(ai) : "memory"); acc ^= r.x ^ r.w; }
        else if (OP == W128_STRIDE16) { v4u w = {v32, v32, v32, v32}; asm volatile("ds_write_b128 %0, %1" ::"v"(ai), "v"(w) : "memory"); }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    out[blockIdx.x * 64 + lane] = (uint32_t)acc + lds[lane];
}

template <int OP>
static void run(uint32_t *d) {
    const int blocks = 256 * 4 * 8, iters = 2048;          // 8 single-wave workgroups per SIMD (16 KB of LDS each: 8 per CU resident at a time)
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 64>>>(d, 16);
    hipEventRecord(e0);
    k<OP><<<blocks, 64>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double per_cu = (double)blocks / 256.0 * iters;         // DS instructions per CU
    printf("%-46s %8.3f ms   %6.2f ns per DS instruction and CU  (~%.1f cycles at 2.1 GHz)\n", NAMES[OP], ms, ms * 1e6 / per_cu, ms * 1e6 / per_cu * 2.1);
}

int main() {
    uint32_t *d; hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
    run<W64_ALIGNED8>(d); run<W64_ROW72>(d); run<W64_ROW67_OFF4>(d); run<W64_ROW67>(d);
    run<W32_ROW68>(d); run<W32_ROW67_UNALIGNED>(d); run<W16_ROW67>(d); run<W8_ROW67>(d);
    run<W128_STRIDE16>(d);
    run<R64_STRIDE16_OFF0>(d); run<R64_STRIDE16_OFF3>(d); run<R128_STRIDE16>(d); run<R64_BROADCAST>(d);
    return 0;
}
