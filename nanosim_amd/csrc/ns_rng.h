// ns_rng.h — counter-based RNG, exact-operation math and the draw layout.
// Product code (device + host side of the engine).  Everything here uses only + - * / sqrt fma and bit
// moves so that the gfx950 results are bit-identical to a plain C evaluation of the same formulas.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NS_HD __host__ __device__ __forceinline__
// NS_DEV: device code of the thread-per-read event chains (ns_chain.h).  The CPU test suite compiles the SAME source for the host
// (tests/chain_host.hip, -DNS_HOST_TEST, host pass only) and holds it against the oracle without a GPU; in the product build the
// qualifier is the plain device one.
#ifdef NS_HOST_TEST
#define NS_DEV __host__ __device__ __forceinline__
#else
#define NS_DEV __device__ __forceinline__
#endif

// stream ids of the Philox counter (c3 bits 18..23)
enum : uint32_t {
    ST_NSEG = 1, ST_REFLEN = 2, ST_GAPLEN = 3, ST_HT = 4, ST_RATIO = 5, ST_STRAND = 6, ST_EVENT = 7,
    ST_UEVENT = 8, ST_POS = 9, ST_IUPAC = 10, ST_SUB = 11, ST_INS = 12, ST_QUAL = 13, ST_HTQ = 14,
    ST_HEAD = 15, ST_TAIL = 16, ST_HPLEN = 17, ST_HPMIS = 18, ST_HPQ = 19, ST_ULEN = 20, ST_SPECIES = 21, ST_TRX = 22, ST_IR = 23
};
#define NS_GAP_SEG 128u
#define NS_MAX_ATTEMPT 1000u
#define NS_EPOCH_FAILS 64u
#define NS_KDE_RETRY 64u
#define NS_POS_RETRY 64u
#define NS_MAX_SEG 64u
#define NS_TRX_BLOCK 1024u      // transcriptome: read indices per block of the sample-until-repeat walk (k_trx_walk)

struct ns_key {           // per-read part of the counter
    uint32_t k0, k1;      // seed
    uint32_t r_lo, r_hi;  // global read index
};

struct u32x4 { uint32_t x, y, z, w; };

// leading zero bits, 32 for 0 (__clz)
NS_HD uint32_t ns_clz32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__clz((int)x);
#else
    return x ? (uint32_t)__builtin_clz(x) : 32u;
#endif
}

// Philox4x32-10 (Salmon, Moraes, Dror, Shaw 2011)
// The two three-way XORs of a round are ONE v_bitop3_b32 each on gfx950 (truth table 0x96 = a ^ b ^ c; the compiler emits two v_xor_b32):
// 20 of the ~62 vector instructions of an evaluation, in every kernel (round 5, same-box A/B: profiles/r05/ab_chain.log)
#if defined(__HIP_DEVICE_COMPILE__)
#define NS_XOR3(a, b, c) __builtin_amdgcn_bitop3_b32((a), (b), (c), 0x96)
#else
#define NS_XOR3(a, b, c) ((a) ^ (b) ^ (c))
#endif
NS_HD u32x4 philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = NS_XOR3((uint32_t)(p1 >> 32), c1, k0);
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = NS_XOR3((uint32_t)(p0 >> 32), c3, k1);
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return u32x4{c0, c1, c2, c3};
}

// counter = (idx, sub, read_lo, read_hi[8] | stream[6] | seg[8] | attempt[10])
NS_HD u32x4 ns_draw(const ns_key &k, uint32_t stream, uint32_t seg, uint32_t attempt, uint32_t idx, uint32_t sub) {
    uint32_t c3 = (k.r_hi & 0xffu) << 24 | (stream & 0x3fu) << 18 | (seg & 0xffu) << 10 | (attempt & 0x3ffu);
    return philox4x32_10(k.k0, k.k1, idx, sub, k.r_lo, c3);
}
NS_HD uint32_t ns_word(const u32x4 &v, uint32_t i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

NS_HD double u32_to_p(uint32_t x) { return ((double)x + 0.5) * 0x1p-32; }
NS_HD double u53_to_p(uint32_t a, uint32_t b) {
    return ((double)(((uint64_t)(a >> 5) << 26) | (uint64_t)(b >> 6)) + 0.5) * 0x1p-53;
}

NS_HD double ns_bits_to_double(uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __longlong_as_double((long long)b);
#else
    double d; __builtin_memcpy(&d, &b, 8); return d;
#endif
}
NS_HD uint64_t ns_double_to_bits(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint64_t)__double_as_longlong(d);
#else
    uint64_t b; __builtin_memcpy(&b, &d, 8); return b;
#endif
}

// log via atanh series on the reduced mantissa
NS_HD double ns_log(double x) {
    uint64_t b = ns_double_to_bits(x);
    int e = (int)((b >> 52) & 0x7ff) - 1023;
    double m = ns_bits_to_double((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    if (m > 1.4142135623730951) { m *= 0.5; e += 1; }
    double s = (m - 1.0) / (m + 1.0);
    double z = s * s;
    double r = 1.0 / 23.0;
    r = fma(r, z, 1.0 / 21.0); r = fma(r, z, 1.0 / 19.0); r = fma(r, z, 1.0 / 17.0);
    r = fma(r, z, 1.0 / 15.0); r = fma(r, z, 1.0 / 13.0); r = fma(r, z, 1.0 / 11.0);
    r = fma(r, z, 1.0 / 9.0);  r = fma(r, z, 1.0 / 7.0);  r = fma(r, z, 1.0 / 5.0);
    r = fma(r, z, 1.0 / 3.0);  r = fma(r, z, 1.0);
    return fma((double)e, 0.6931471805599453, 2.0 * s * r);
}

NS_HD double ns_exp(double y) {
    if (y > 700.0) y = 700.0;
    if (y < -700.0) y = -700.0;
    double k = floor(fma(y, 1.4426950408889634, 0.5));
    double r = fma(-k, 6.93147180369123816490e-01, y);
    r = fma(-k, 1.90821492927058770002e-10, r);
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0); p = fma(p, r, 1.0 / 39916800.0); p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);    p = fma(p, r, 1.0 / 40320.0);    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);       p = fma(p, r, 1.0 / 120.0);      p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);         p = fma(p, r, 0.5);              p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    double sc = ns_bits_to_double((uint64_t)((int64_t)k + 1023) << 52);
    return p * sc;
}

// inverse normal CDF (P. J. Acklam's rational approximation)
NS_HD double ns_norminv(double p) {
    const double plow = 0.02425;
    if (p < plow || p > 1.0 - plow) {
        double t = (p < plow) ? p : 1.0 - p;
        double q = sqrt(-2.0 * ns_log(t));
        double num = -7.784894002430293e-03;
        num = fma(num, q, -3.223964580411365e-01); num = fma(num, q, -2.400758277161838e+00);
        num = fma(num, q, -2.549732539343734e+00); num = fma(num, q, 4.374664141464968e+00);
        num = fma(num, q, 2.938163982698783e+00);
        double den = 7.784695709041462e-03;
        den = fma(den, q, 3.224671290700398e-01); den = fma(den, q, 2.445134137142996e+00);
        den = fma(den, q, 3.754408661907416e+00); den = fma(den, q, 1.0);
        double x = num / den;
        return (p < plow) ? x : -x;
    }
    double q = p - 0.5, r = q * q;
    double num = -3.969683028665376e+01;
    num = fma(num, r, 2.209460984245205e+02); num = fma(num, r, -2.759285104469687e+02);
    num = fma(num, r, 1.383577518672690e+02); num = fma(num, r, -3.066479806614716e+01);
    num = fma(num, r, 2.506628277459239e+00);
    double den = -5.447609879822406e+01;
    den = fma(den, r, 1.615858368580409e+02); den = fma(den, r, -1.556989798598866e+02);
    den = fma(den, r, 6.680131188771972e+01); den = fma(den, r, -1.328068155288572e+01);
    den = fma(den, r, 1.0);
    return num * q / den;
}

NS_HD double ns_pow10m1(double x) { return ns_exp(x * 2.302585092994046) - 1.0; }

// ---- event record packing: {pos:32 | len:12 | type:2 | shift+2^17:18} ------------------------------------
#define NS_EV_LEN_MAX 4095u
#define NS_EV_SHIFT_BIAS 131072
NS_HD uint32_t ns_ev_pack(uint32_t len, uint32_t type, int32_t shift) {
    return (len & 0xfffu) | (type & 3u) << 12 | (uint32_t)(shift + NS_EV_SHIFT_BIAS) << 14;
}
NS_HD uint32_t ns_ev_len(uint32_t info) { return info & 0xfffu; }
NS_HD uint32_t ns_ev_type(uint32_t info) { return (info >> 12) & 3u; }
NS_HD int32_t ns_ev_shift(uint32_t info) { return (int32_t)(info >> 14) - NS_EV_SHIFT_BIAS; }
