// ot.hip -- device kernels of the OT balanced grouping (method/utils.py:628-656).
//
//   ure_ot_cost       utils.py:637  dist = ((X - centroid[:,None])**2).sum(axis=2)
//   ure_ot_centroids  utils.py:648  new_centroid[c] = X[label == c].mean(axis=0)
//   ure_kmeans_cost / ure_kmeans_centroids  utils.py:373-375, 402-403  the comparison clusterer's
//                     distances and centroid update (scipy csr arithmetic, see below)
//
// Group labels must match the reference bit for bit, and the LP that follows is
// sensitive to the last bit of a cost only at near-ties -- which is exactly where a
// different rounding would flip a label.  Both kernels therefore reproduce numpy's
// fp32 evaluation ORDER rather than the fastest one: the squared differences of a row
// are summed with numpy's 8-accumulator pairwise rule, and a centroid is the
// sequential fp32 sum of its member rows in ascending row id divided once by the
// count.  This is byte-for-byte work bounded by HBM reads of X (n*d*4 bytes per
// centroid pass); it is deliberately NOT reshaped into an MFMA GEMM
// (|x|^2 - 2 x.c + |c|^2 rounds differently and would break label parity).
#include "ure_internal.h"

namespace ure {

// numpy pairwise_sum for n <= 128 contiguous fp32 terms t_j = (x_j - c_j)^2.
__device__ float np_block_sum(const float *__restrict__ x, const float *__restrict__ c, int n)
{
    if (n < 8) {
        float res = 0.f;
        for (int j = 0; j < n; ++j) {
            const float t = __fsub_rn(x[j], c[j]);
            res = __fadd_rn(res, __fmul_rn(t, t));
        }
        return res;
    }
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float t = __fsub_rn(x[j], c[j]);
        r[j] = __fmul_rn(t, t);
    }
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float t = __fsub_rn(x[i + j], c[i + j]);
            r[j] = __fadd_rn(r[j], __fmul_rn(t, t));
        }
    }
    float res = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])),
                          __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
    for (; i < n; ++i) {
        const float t = __fsub_rn(x[i], c[i]);
        res = __fadd_rn(res, __fmul_rn(t, t));
    }
    return res;
}

// numpy splits runs longer than 128 in two (first half rounded down to a multiple of 8).
__device__ float np_pairwise(const float *__restrict__ x, const float *__restrict__ c, int n)
{
    if (n <= 128) return np_block_sum(x, c, n);
    int n2 = n / 2;
    n2 -= n2 % 8;
    return __fadd_rn(np_pairwise(x, c, n2), np_pairwise(x + n2, c + n2, n - n2));
}

__global__ __launch_bounds__(kBlock) void ot_cost_kernel(const float *__restrict__ X, const float *__restrict__ C, int64_t n,
                                                         int k, int d, float *__restrict__ dist)
{
    const int64_t total = n * k;
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
        const int64_t i = t % n;
        const int c = (int)(t / n);
        dist[t] = np_pairwise(X + i * d, C + (size_t)c * d, d);
    }
}

// One thread per (cluster, column): ascending walk over the points.
__global__ __launch_bounds__(kBlock) void ot_centroid_kernel(const float *__restrict__ X, const int32_t *__restrict__ label,
                                                             int64_t n, int k, int d, float *__restrict__ C,
                                                             int32_t *__restrict__ counts)
{
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= k * d) return;
    const int c = t / d, j = t % d;
    float sum = 0.f;
    int cnt = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (label[i] == c) {
            const float x = X[i * d + j];
            sum = cnt == 0 ? x : __fadd_rn(sum, x);
            ++cnt;
        }
    }
    C[t] = __fdiv_rn(sum, (float)cnt);
    if (j == 0 && counts) counts[c] = cnt;
}

// ---- comparison clusterers (utils.py:354-418): k-means on a csr embedding ----------------------
// utils.py:373-375  dist = (-2 * sp_mat * centroid.T).A; dist += e_square; dist += c_square, float32:
// every inner sum in scipy's csr order (columns ascending, one multiply and one add per term), then
// ((-2 dot) + |x|^2) + |c|^2.  dist is [n][k] row-major as in the reference.
__global__ __launch_bounds__(kBlock) void kmeans_cost_kernel(const float *__restrict__ X, const float *__restrict__ C, int64_t n,
                                                             int k, int d, float *__restrict__ dist)
{
    const int64_t total = n * k;
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
        const int64_t i = t / k;
        const int c = (int)(t % k);
        const float *__restrict__ x = X + i * d;
        const float *__restrict__ cc = C + (size_t)c * d;
        float dot = 0.f, esq = 0.f, csq = 0.f;
        for (int j = 0; j < d; ++j) {
            dot = __fadd_rn(dot, __fmul_rn(x[j], cc[j]));
            esq = __fadd_rn(esq, __fmul_rn(x[j], x[j]));
            csq = __fadd_rn(csq, __fmul_rn(cc[j], cc[j]));
        }
        dist[t] = __fadd_rn(__fadd_rn(__fmul_rn(-2.0f, dot), esq), csq);
    }
}

// utils.py:402-403  centroid[j] = csr_matrix(sp_mat[label == j].mean(axis=0)): scipy's sparse mean
// multiplies every member row by float32(1 / count) and sums in ascending row order in float32.
__global__ __launch_bounds__(kBlock) void kmeans_centroid_kernel(const float *__restrict__ X, const int32_t *__restrict__ label,
                                                                 int64_t n, int k, int d, float *__restrict__ C,
                                                                 int32_t *__restrict__ counts)
{
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= k * d) return;
    const int c = t / d, j = t % d;
    int cnt = 0;
    for (int64_t i = 0; i < n; ++i) cnt += label[i] == c ? 1 : 0;
    const float inv = (float)(1.0 / (double)cnt);
    float sum = 0.f;
    for (int64_t i = 0; i < n; ++i)
        if (label[i] == c) sum = __fadd_rn(sum, __fmul_rn(X[i * d + j], inv));
    C[t] = cnt ? sum : 0.f;
    if (j == 0 && counts) counts[c] = cnt;
}

}  // namespace ure

using namespace ure;

extern "C" {

int ure_ot_cost(const float *X, const float *C, int64_t n, int k, int d, float *dist, void *stream)
{
    URE_ARG(X && C && dist && n > 0 && k > 0 && d > 0 && d <= 256);
    const int64_t total = n * k;
    const unsigned blocks = (unsigned)std::min<int64_t>((total + kBlock - 1) / kBlock, 256 * 16);
    hipLaunchKernelGGL(ot_cost_kernel, dim3(blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), X, C, n, k, d, dist);
    URE_HIP(hipGetLastError());
    return 0;
}

int ure_ot_centroids(const float *X, const int32_t *label, int64_t n, int k, int d, float *C, int32_t *counts, void *stream)
{
    URE_ARG(X && label && C && n > 0 && k > 0 && d > 0);
    const unsigned blocks = (unsigned)((k * d + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(ot_centroid_kernel, dim3(blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), X, label, n, k, d,
                       C, counts);
    URE_HIP(hipGetLastError());
    return 0;
}

int ure_kmeans_cost(const float *X, const float *C, int64_t n, int k, int d, float *dist_nk, void *stream)
{
    URE_ARG(X && C && dist_nk && n > 0 && k > 0 && d > 0);
    const int64_t total = n * k;
    const unsigned blocks = (unsigned)std::min<int64_t>((total + kBlock - 1) / kBlock, 256 * 16);
    hipLaunchKernelGGL(kmeans_cost_kernel, dim3(blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), X, C, n, k, d, dist_nk);
    URE_HIP(hipGetLastError());
    return 0;
}

int ure_kmeans_centroids(const float *X, const int32_t *label, int64_t n, int k, int d, float *C, int32_t *counts, void *stream)
{
    URE_ARG(X && label && C && n > 0 && k > 0 && d > 0);
    const unsigned blocks = (unsigned)((k * d + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(kmeans_centroid_kernel, dim3(blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), X, label, n, k, d,
                       C, counts);
    URE_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
