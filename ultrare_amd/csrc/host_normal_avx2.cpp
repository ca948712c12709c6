// host_normal_avx2.cpp -- the Box-Muller half of ATen's float32 `tensor.normal_()` on x86: normal_fill_16_AVX2
// (aten/src/ATen/native/cpu/DistributionTemplates.h), the kernel that runs on every AVX2-capable host (normal_stub has no AVX-512
// variant: the dispatcher falls back to the AVX2 one).  It turns 16 uniforms into 16 normals with the polynomial log / sincos of
// avx_mathfun.h, a header the installed PyTorch ships (torch/include/ATen/native/cpu/avx_mathfun.h, zlib licence): this file
// INCLUDES that header from the installation it is built against and repeats the dozen lines that call it, so the arithmetic is
// the installed library's own.  Built apart from the rest of the library (ultrare_amd/build.py) with -mavx2 -mfma
// -ffp-contract=fast: PyTorch's build contracts the header's mul / add pairs into FMAs, and only with the same contraction are the
// results bit-identical (99.0 % of the values without it, 100 % with it, checked against torch by ultrare_amd.rng.native_fill_ok at
// run time -- where a bit differs the package keeps torch's own fill).
#include <cstdint>

#if defined(__x86_64__) && defined(URE_HAVE_AVX_MATHFUN)
#define CPU_CAPABILITY_AVX2 1
#include <immintrin.h>
#include <ATen/native/cpu/avx_mathfun.h>

extern "C" int ure_host_normal_blocks(float *data, int64_t n_blocks, float mean, float std_)
{
    if (!__builtin_cpu_supports("avx2") || !__builtin_cpu_supports("fma")) return -4;
    const __m256 two_pi = _mm256_set1_ps(2.0f * 3.14159265358979323846);
    const __m256 one = _mm256_set1_ps(1.0f);
    const __m256 minus_two = _mm256_set1_ps(-2.0f);
    const __m256 mean_v = _mm256_set1_ps(mean);
    const __m256 std_v = _mm256_set1_ps(std_);
    for (int64_t b = 0; b < n_blocks; ++b) {
        float *d = data + 16 * b;
        const __m256 u1 = _mm256_sub_ps(one, _mm256_loadu_ps(d));          // [0, 1) -> (0, 1] for the logarithm
        const __m256 u2 = _mm256_loadu_ps(d + 8);
        const __m256 radius = _mm256_sqrt_ps(_mm256_mul_ps(minus_two, log256_ps(u1)));
        const __m256 theta = _mm256_mul_ps(two_pi, u2);
        __m256 sintheta, costheta;
        sincos256_ps(theta, &sintheta, &costheta);
        _mm256_storeu_ps(d, _mm256_fmadd_ps(_mm256_mul_ps(radius, costheta), std_v, mean_v));
        _mm256_storeu_ps(d + 8, _mm256_fmadd_ps(_mm256_mul_ps(radius, sintheta), std_v, mean_v));
    }
    return 0;
}
#else
extern "C" int ure_host_normal_blocks(float *, int64_t, float, float) { return -4; }       // (not built against a PyTorch that ships the header)
#endif
