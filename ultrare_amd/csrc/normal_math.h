// normal_math.h -- `tensor.normal_()`'s arithmetic for a contiguous float32 CPU tensor (utils.py:31-40: MF.init_weight), one element
// pair at a time, for the HOST and the DEVICE alike.
//
// ATen (aten/src/ATen/native/cpu/DistributionTemplates.h, normal_fill_16_AVX2 -- the kernel every x86 host with AVX2 runs) turns 16
// uniforms into 16 normals by Box-Muller:  u1 = 1 - data[j], u2 = data[j + 8];  r = sqrt(-2 log u1), t = 2 pi u2;
// data[j] = r cos t, data[j + 8] = r sin t,  with the polynomial log / sincos of avx_mathfun.h (Cephes' logf and sinf / cosf in
// Pommier's SSE form).  Those are sequences of IEEE float32 add / mul / fma and integer operations, eight lanes at a time without any
// cross-lane step, so one lane can be restated in scalar code -- provided every rounding happens where the vector code's does.
// PyTorch's build contracts the header's mul + add pairs into fused multiply-adds; which pairs, where there is a choice, was
// settled against torch itself (tests/test_cpu_host.py: the four candidate patterns on 2^24 uniforms; exactly one agrees, on all):
//     y * z + e * q1   ->  fma(y, z, e * q1)          (the first multiplication of the statement order is the fused one)
//     y * z - z * 0.5  ->  fma(y, z, -(z * 0.5))
// Everything else has one reading.  This file must be compiled WITHOUT implicit contraction (-ffp-contract=off: the library's flags)
// -- every fma below is written out.  The device's v_fma_f32 / v_mul_f32 / v_add_f32 are IEEE float32 operations with denormals, and
// hipcc's sqrtf is correctly rounded by default (-fhip-fp32-correctly-rounded-divide-sqrt), so the device computes the host's bits;
// ultrare_amd.rng.device_fill_ok() compares a device fill with torch once per process all the same.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define URE_HD __host__ __device__ __forceinline__
#else
#define URE_HD inline
#endif

namespace ure {

URE_HD float nm_bits_to_float(uint32_t b) { return __builtin_bit_cast(float, b); }
URE_HD uint32_t nm_float_to_bits(float f) { return __builtin_bit_cast(uint32_t, f); }
URE_HD float nm_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
#define NM_F(x) ((float)(x))      // the header's constants are double literals narrowed when its tables are initialised

// at::uniform_real_distribution<float>(0, 1) on one 32-bit MT19937 output: 24 bits, exact
URE_HD float nm_uniform(uint32_t mt_output) { return (float)(mt_output & 0xffffffu) * 5.9604644775390625e-8f; }

// log256_ps for 0 < x <= 1 (the caller's x is 1 - u >= 2^-24: never denormal, never <= 0)
template <int kVariant = 0>
URE_HD float nm_log(float x)
{
    const float min_norm = nm_bits_to_float(0x00800000u);
    x = x > min_norm ? x : min_norm;                                      // _mm256_max_ps(x, min_norm_pos)
    const uint32_t bits = nm_float_to_bits(x);
    float e = (float)((int32_t)(bits >> 23) - 0x7f);
    x = nm_bits_to_float((bits & ~0x7f800000u) | 0x3f000000u);            // the mantissa in [0.5, 1)
    e = e + 1.0f;
    const bool lt = x < NM_F(0.707106781186547524);
    const float tmp = lt ? x : 0.0f;
    x = x - 1.0f;
    e = e - (lt ? 1.0f : 0.0f);
    x = x + tmp;
    const float z = x * x;
    float y = NM_F(7.0376836292E-2);
    y = nm_fma(y, x, NM_F(-1.1514610310E-1));
    y = nm_fma(y, x, NM_F(1.1676998740E-1));
    y = nm_fma(y, x, NM_F(-1.2420140846E-1));
    y = nm_fma(y, x, NM_F(1.4249322787E-1));
    y = nm_fma(y, x, NM_F(-1.6668057665E-1));
    y = nm_fma(y, x, NM_F(2.0000714765E-1));
    y = nm_fma(y, x, NM_F(-2.4999993993E-1));
    y = nm_fma(y, x, NM_F(3.3333331174E-1));
    y = y * x;
    if constexpr ((kVariant & 1) == 0)
        y = nm_fma(y, z, e * NM_F(-2.12194440e-4));                       // y * z + e * q1: the first product is the fused one
    else
        y = nm_fma(e, NM_F(-2.12194440e-4), y * z);                       // (the other reading; tests only)
    y = nm_fma(-z, 0.5f, y);                                              // y - z * 0.5 (the product is exact either way)
    x = x + y;
    x = nm_fma(e, NM_F(0.693359375), x);                                       // + e * q2 (exact product)
    return x;
}

// sincos256_ps for x >= 0
template <int kVariant = 0>
URE_HD void nm_sincos(float x, float *s, float *c)
{
    const uint32_t sign_in = nm_float_to_bits(x) & 0x80000000u;
    x = nm_bits_to_float(nm_float_to_bits(x) & 0x7fffffffu);
    float y = x * NM_F(1.27323954473516);                                      // 4 / pi
    int32_t j = (int32_t)y;                                               // _mm256_cvttps_epi32
    j = (j + 1) & ~1;
    y = (float)j;
    const uint32_t swap_sign_sin = ((uint32_t)j & 4u) << 29;
    const bool poly_mask = (j & 2) == 0;
    x = nm_fma(y, NM_F(-0.78515625), x);                                       // extended-precision modular arithmetic: x - y DP1 - y DP2 - y DP3
    x = nm_fma(y, NM_F(-2.4187564849853515625e-4), x);
    x = nm_fma(y, NM_F(-3.77489497744594108e-8), x);
    const uint32_t sign_cos = (~(uint32_t)(j - 2) & 4u) << 29;
    const uint32_t sign_sin = sign_in ^ swap_sign_sin;
    const float z = x * x;
    float yc = NM_F(2.443315711809948E-005);
    yc = nm_fma(yc, z, NM_F(-1.388731625493765E-003));
    yc = nm_fma(yc, z, NM_F(4.166664568298827E-002));
    yc = yc * z;
    if constexpr ((kVariant & 2) == 0)
        yc = nm_fma(yc, z, -(z * 0.5f));                                  // y * z - z * 0.5: the first product is the fused one
    else
        yc = nm_fma(-z, 0.5f, yc * z);                                    // (the other reading; tests only)
    yc = yc + 1.0f;
    float ys = NM_F(-1.9515295891E-4);
    ys = nm_fma(ys, z, NM_F(8.3321608736E-3));
    ys = nm_fma(ys, z, NM_F(-1.6666654611E-1));
    ys = ys * z;
    ys = nm_fma(ys, x, x);
    // the selection as the vector code makes it (and / andnot / sub / add: signed zeros come out as there)
    const float ysin2 = poly_mask ? ys : 0.0f;
    const float ysin1 = poly_mask ? 0.0f : yc;
    ys = ys - ysin2;
    yc = yc - ysin1;
    const float xs = ysin1 + ysin2;
    const float xc = yc + ys;
    *s = nm_bits_to_float(nm_float_to_bits(xs) ^ sign_sin);
    *c = nm_bits_to_float(nm_float_to_bits(xc) ^ sign_cos);
}

// One lane of normal_fill_16_AVX2 with mean 0, std 1: uniforms (ua, ub) = (data[j], data[j + 8]) -> (data[j], data[j + 8])
template <int kVariant = 0>
URE_HD void nm_box_muller(float ua, float ub, float *out_a, float *out_b)
{
    const float u1 = 1.0f - ua;
    const float radius = __builtin_sqrtf(-2.0f * nm_log<kVariant>(u1));
    const float theta = (float)(2.0f * 3.14159265358979323846) * ub;      // _mm256_set1_ps(2.0f * 3.14159265358979323846)
    float sn, cs;
    nm_sincos<kVariant>(theta, &sn, &cs);
    *out_a = nm_fma(radius * cs, 1.0f, 0.0f);
    *out_b = nm_fma(radius * sn, 1.0f, 0.0f);
}

}  // namespace ure
