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

#include <cmath>
#include <cstring>

namespace ure {

// numpy pairwise_sum for n <= 128 contiguous fp32 terms t_j = (x_j - c_j)^2.
__device__ __forceinline__ float np_block_sum(const float *__restrict__ x, const float *__restrict__ c, int n)
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

// numpy splits runs longer than 128 in two (first half rounded down to a multiple of 8) -- for n <= 256 (ure_ot_cost's limit) that is at most one split, so
// no recursion.  Inlined into its caller, the loads keep their address spaces: as a called, recursive function (rounds 1-4) it read ot_cost_tiled_kernel's
// LDS tile and the centroid through flat pointers, and the kernel spent its time there.
__device__ __forceinline__ float np_pairwise_le256(const float *__restrict__ x, const float *__restrict__ c, int n)
{
    if (n <= 128) return np_block_sum(x, c, n);
    int n2 = n / 2;
    n2 -= n2 % 8;
    return __fadd_rn(np_block_sum(x, c, n2), np_block_sum(x + n2, c + n2, n - n2));
}

__global__ __launch_bounds__(kBlock) void ot_cost_kernel(const float *__restrict__ X, const float *__restrict__ C, int64_t n,
                                                         int k, int d, float *__restrict__ dist)
{
    const int64_t total = n * k;
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
        const int64_t i = t % n;
        const int c = (int)(t / n);
        dist[t] = np_pairwise_le256(X + i * d, C + (size_t)c * d, d);
    }
}

// The same values with coalesced HBM traffic: a workgroup stages 64 rows of X in LDS (rows padded by one float:
// lanes that read one column of 64 different rows hit 64 different banks), then lane = row, wave = centroid
// (c = wave, wave + 4, ...): the centroid's values are wave-uniform, the row is read from LDS in numpy's order,
// and the 64 results of a wave go out as one 256-byte store.  X is read from HBM once instead of once per
// centroid with a stride of d floats per lane (MI355X_MICROARCH.md: 64 lanes in 64 rows, ~17x below peak).
constexpr int kCostRows = 64;
__global__ __launch_bounds__(kBlock) void ot_cost_tiled_kernel(const float *__restrict__ X, const float *__restrict__ C, int64_t n, int k, int d,
                                                               float *__restrict__ dist)
{
    extern __shared__ float tile[];                       // [kCostRows][d + 1]
    const int ld = d + 1;
    const int64_t i0 = (int64_t)blockIdx.x * kCostRows;
    const int rows = (int)min<int64_t>(kCostRows, n - i0);
    for (int t = threadIdx.x; t < rows * d; t += kBlock) tile[(t / d) * ld + (t % d)] = X[i0 * d + t];     // contiguous read
    __syncthreads();
    // (the wave's number as a SCALAR: the centroid's values are then read by scalar loads -- as a vector value the compiler cannot know to be uniform it
    // made every one of them a 64-lane load of one address, 1,024 per wavefront at k = 32, d = 128: the kernel's time)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (lane >= rows) return;
    const float *x = tile + lane * ld;
    for (int c = wave; c < k; c += kWavesPerBlock) dist[(size_t)c * n + i0 + lane] = np_pairwise_le256(x, C + (size_t)c * d, d);
}

// ---- MFMA form of the cost matrix: |x|^2 - 2 x.c + |c|^2 with the n x k x d contraction on the matrix cores ------
// BASELINE.json's north_star asks for the OT cost matrix as a GEMM on MFMA.  The LABELS must equal the reference's bit
// for bit, and this form rounds differently from utils.py:637's (x - c)^2 sum, so it can never be the arithmetic of
// record: it is an optional fast path whose labels the host cross-checks against the exact kernel's every round
// (method/utils.py::ot_cluster, URE_OT_MFMA=1).  fp32 in, fp32 accumulate (v_mfma_f32_32x32x2_f32: exact products,
// no bf16 rounding -- the costs decide near-ties).  One wave = 32 points x 32 centroids: lane l feeds A[l & 31][l >> 5]
// = x_{i0 + (l & 31)}[kk + (l >> 5)] from an LDS tile of its 32 rows and B[l >> 5][l & 31] = c_{c0 + (l & 31)}[kk + (l >> 5)];
// accumulator register r of lane l is row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31.
typedef float ure_f16v __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(kBlock) void ot_cost_mfma_kernel(const float *__restrict__ X, const float *__restrict__ C, int64_t n, int k, int d,
                                                              float *__restrict__ dist)
{
    extern __shared__ float tile[];                       // [4 waves][32 rows][d + 1]
    const int ld = d + 1;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t i0 = ((int64_t)blockIdx.x * kWavesPerBlock + wave) * 32;
    float *xs = tile + (size_t)wave * 32 * ld;
    const int rows = (int)max<int64_t>(0, min<int64_t>(32, n - i0));
    for (int t = lane; t < 32 * d; t += kWave) {
        const int r = t / d, j = t % d;
        xs[r * ld + j] = r < rows ? X[(i0 + r) * d + j] : 0.f;
    }
    __builtin_amdgcn_wave_barrier();
    if (rows == 0) return;
    const int m = lane & 31, h = lane >> 5;
    float xx = 0.f;                                        // |x_m|^2, both half-waves compute it
    for (int j = 0; j < d; ++j) xx = fmaf(xs[m * ld + j], xs[m * ld + j], xx);
    float xrow[16];                                        // |x|^2 of the 16 rows this lane's accumulators belong to (all lanes active here:
#pragma unroll                                             //  a cross-lane read inside the `cj < k` branch below would see inactive lanes)
    for (int r = 0; r < 16; ++r) xrow[r] = __shfl(xx, (r & 3) + 8 * (r >> 2) + 4 * h, kWave);
    for (int c0 = 0; c0 < k; c0 += 32) {
        const int cj = c0 + m;                             // this lane's centroid (as B's column)
        const float *crow = C + (size_t)min(cj, k - 1) * d;
        float cc = 0.f;
        for (int j = 0; j < d; ++j) cc = fmaf(crow[j], crow[j], cc);
        ure_f16v acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int kk = 0; kk < d; kk += 2) {
            const int col = kk + h;
            const float a = col < d ? xs[m * ld + col] : 0.f;
            const float b = (col < d && cj < k) ? crow[col] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        if (cj < k) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < rows) dist[(size_t)cj * n + i0 + row] = fmaxf(fmaf(-2.0f, acc[r], xrow[r]) + cc, 0.f);
            }
        }
    }
}

// ---- warm start of the exact solver: potentials of the clusters by dual subgradient ascent ---------------
// The transportation LP's dual is max_pi sum_i min_c (cost[i][c] - pi[c]) + (n / k) sum_c pi[c]; a subgradient in
// pi[c] is (n / k) - load_c(pi), load_c = the number of points whose cheapest reduced cost is c.  A hundred
// sign-based steps against the imbalance load_c - n / k bring the loads within a few points of balance (measured
// at n = 162,000, k = 32: 53,342 misplaced points -> 10..26), so the exact successive-shortest-path solver that follows
// (ot_solver.cpp) has tens of augmentations to make instead of ~94,000.  This is only a starting point: the
// solver's result is the exact optimum whatever the potentials are.
constexpr int kPotMaxK = 256;
struct pot_state {
    double gap_sum;              // sum over points of (second cheapest - cheapest) cost
    unsigned int loads[kPotMaxK];
    float pi[kPotMaxK];
    float step[kPotMaxK];        // per-cluster step (sign-based: robust to the scale of the costs)
    float prev[kPotMaxK];        // sign of the cluster's previous imbalance
    float best_pi[kPotMaxK];     // the potentials with the smallest imbalance seen so far ...
    float best_imb;              // ... and that imbalance, sum_c |load_c - n / k|
    int it;
};

__global__ __launch_bounds__(kBlock) void ot_pot_step_kernel(const float *__restrict__ dist, int64_t n, int k, pot_state *__restrict__ st, int measure_gap)
{
    __shared__ float pi[kPotMaxK];
    __shared__ unsigned int hist[kPotMaxK];
    __shared__ double gsum[kWavesPerBlock];
    for (int c = threadIdx.x; c < k; c += kBlock) { pi[c] = st->pi[c]; hist[c] = 0; }
    __syncthreads();
    double gap = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        float best = 3.4e38f, second = 3.4e38f;
        int arg = 0;
        // (eight costs requested together, then compared in cluster order: one load, its comparison, the next load was k memory latencies in a row per point)
        for (int c0 = 0; c0 < k; c0 += 8) {
            float cost[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) cost[u] = dist[(size_t)min(c0 + u, k - 1) * n + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = c0 + u;
                if (c < k) {
                    const float v = cost[u] - pi[c];
                    if (v < best) { second = best; best = v; arg = c; }
                    else if (v < second) second = v;
                }
            }
        }
        atomicAdd(&hist[arg], 1u);
        gap += (double)(second - best);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < k; c += kBlock)
        if (hist[c]) atomicAdd(&st->loads[c], hist[c]);
    if (measure_gap) {
        for (int o = 32; o > 0; o >>= 1) gap += __shfl_xor(gap, o, kWave);
        if ((threadIdx.x & 63) == 0) gsum[threadIdx.x >> 6] = gap;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int w = 0; w < kWavesPerBlock; ++w) t += gsum[w];
            atomicAdd(&st->gap_sum, t);
        }
    }
}

__global__ void ot_pot_update_kernel(int64_t n, int k, pot_state *__restrict__ st, int iters)
{
    // Sign-based steps (Rprop): a cluster's step grows by 1.3 while its imbalance keeps its sign and halves when the
    // sign flips; near balance the move is scaled down with the relative imbalance, and over the last third of the
    // iterations every step shrinks by 3 % per iteration.  A plain subgradient step eta_t (load - n / k) oscillated for
    // centroids that sit close together (their cost gaps are far below the mean gap that sets eta).  The iterate with
    // the smallest imbalance is the one handed to the solver.
    __shared__ float red[kPotMaxK];
    const int c = threadIdx.x;
    const float target = (float)n / (float)k;
    const int it = st->it;
    const float g = c < k ? (float)st->loads[c] - target : 0.f;
    red[c] = fabsf(g);
    __syncthreads();
    for (int o = kPotMaxK / 2; o > 0; o >>= 1) {
        if (c < o) red[c] += red[c + o];
        __syncthreads();
    }
    const float imb = red[0];
    const bool better = it == 0 || imb < st->best_imb;
    __syncthreads();
    if (c < k) {
        if (better) st->best_pi[c] = st->pi[c];                  // the potentials that produced these loads
        if (it == 0) { st->step[c] = 0.05f * (float)(st->gap_sum / (double)n); st->prev[c] = 0.f; }
        const float sgn = g > 0.f ? 1.f : (g < 0.f ? -1.f : 0.f);
        const float same = sgn * st->prev[c];
        float step = st->step[c];
        step *= same > 0.f ? 1.3f : (same < 0.f ? 0.5f : 1.0f);
        if (3 * it > 2 * iters) step *= 0.97f;
        st->pi[c] -= step * sgn * fminf(1.0f, fabsf(g) / (0.02f * target) + 0.05f);
        st->step[c] = step;
        st->prev[c] = sgn;
        st->loads[c] = 0;
    }
    if (c == 0) {
        if (better) st->best_imb = imb;
        st->it = it + 1;
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

// The same means from member lists: order[off[c] .. off[c+1]) = the points of cluster c in ascending id (a stable
// counting sort of the labels, made on the host where the labels come from).  A thread still adds its column of its
// cluster's rows one after the other in ascending id -- numpy's order -- but walks ~n / k rows instead of testing all n
// labels (n = 162,000, k = 32, d = 128: 11.9 -> 6.6 ms; with the loads of 32 members in flight, round 5: below).
__global__ __launch_bounds__(kBlock) void ot_centroid_members_kernel(const float *__restrict__ X, const int32_t *__restrict__ order,
                                                                     const int64_t *__restrict__ off, int k, int d, float *__restrict__ C,
                                                                     int32_t *__restrict__ counts)
{
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= k * d) return;
    const int c = t / d, j = t % d;
    const int64_t b = off[c], e = off[c + 1];
    float sum = 0.f;
    // the ADDS stay one after the other in ascending id; the loads need not: 32 member ids, then their 32 values, are requested together (one id, then
    // its value, then the add was two dependent memory levels per member: 6.6 ms for the 5,063 members of a cluster at n = 162,000, k = 32, d = 128)
    constexpr int kAhead = 32;
    int64_t q = b;
    for (; q + kAhead <= e; q += kAhead) {
        int id[kAhead];
        float x[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; ++u) id[u] = order[q + u];
#pragma unroll
        for (int u = 0; u < kAhead; ++u) x[u] = X[(size_t)id[u] * d + j];
        float s0 = q == b ? x[0] : __fadd_rn(sum, x[0]);
#pragma unroll
        for (int u = 1; u < kAhead; ++u) s0 = __fadd_rn(s0, x[u]);
        sum = s0;
    }
    for (; q < e; ++q) {
        const float x = X[(size_t)order[q] * d + j];
        sum = q == b ? x : __fadd_rn(sum, x);
    }
    C[t] = __fdiv_rn(sum, (float)(e - b));
    if (j == 0 && counts) counts[c] = (int32_t)(e - b);
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
    const size_t lds = (size_t)kCostRows * (d + 1) * sizeof(float);
    if (lds <= 64 * 1024 - 256 && (n + kCostRows - 1) / kCostRows <= 0x7fffffff) {
        const unsigned blocks = (unsigned)((n + kCostRows - 1) / kCostRows);
        hipLaunchKernelGGL(ot_cost_tiled_kernel, dim3(blocks), dim3(kBlock), lds, static_cast<hipStream_t>(stream), X, C, n, k, d, dist);
    } else {      // rows too wide for the 64-row LDS tile: one thread per (point, centroid)
        const int64_t total = n * k;
        const unsigned blocks = (unsigned)std::min<int64_t>((total + kBlock - 1) / kBlock, 256 * 16);
        hipLaunchKernelGGL(ot_cost_kernel, dim3(blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), X, C, n, k, d, dist);
    }
    URE_HIP(hipGetLastError());
    return 0;
}

int ure_ot_cost_mfma(const float *X, const float *C, int64_t n, int k, int d, float *dist, void *stream)
{
    URE_ARG(X && C && dist && n > 0 && k > 0 && d > 0 && d <= 256);
    const size_t lds = (size_t)kWavesPerBlock * 32 * (d + 1) * sizeof(float);
    if (lds > 160 * 1024) return fail(-1, "ure_ot_cost_mfma: d=%d does not fit the LDS tile", d);
    static bool raised = false;
    if (!raised && lds > 64 * 1024) {
        URE_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(ot_cost_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        raised = true;
    }
    const unsigned blocks = (unsigned)((n + 32 * kWavesPerBlock - 1) / (32 * kWavesPerBlock));
    hipLaunchKernelGGL(ot_cost_mfma_kernel, dim3(blocks), dim3(kBlock), lds, static_cast<hipStream_t>(stream), X, C, n, k, d, dist);
    URE_HIP(hipGetLastError());
    return 0;
}

int ure_ot_potentials(const float *dist, int64_t n, int k, int iters, double *pi_host, int64_t *misplaced, void *stream)
{
    URE_ARG(dist && pi_host && n > 0 && k > 0 && iters >= 0);
    if (misplaced) *misplaced = -1;
    if (k > kPotMaxK || k < 2 || iters == 0) {                   // no warm start: the solver starts cold
        for (int c = 0; c < k; ++c) pi_host[c] = 0.0;
        return 0;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    pot_state *dev = nullptr;
    pot_state host;
    std::memset(&host, 0, sizeof(host));
    for (int c = 0; c < k; ++c) host.pi[c] = std::isfinite(pi_host[c]) ? (float)pi_host[c] : 0.f;      // start: the caller's (the previous round's)
    URE_HIP(hipMalloc(&dev, sizeof(pot_state)));
    hipError_t e = hipMemcpyAsync(dev, &host, sizeof(pot_state), hipMemcpyHostToDevice, st);
    const unsigned blocks = (unsigned)std::min<int64_t>((n + kBlock - 1) / kBlock, 1024);
    for (int it = 0; it <= iters && e == hipSuccess; ++it) {      // one more evaluation so that the last iterate is judged too
        hipLaunchKernelGGL(ot_pot_step_kernel, dim3(blocks), dim3(kBlock), 0, st, dist, n, k, dev, it == 0 ? 1 : 0);
        hipLaunchKernelGGL(ot_pot_update_kernel, dim3(1), dim3(kPotMaxK), 0, st, n, k, dev, iters);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&host, dev, sizeof(pot_state), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(dev);
    if (e != hipSuccess) return fail((int)e, "ure_ot_potentials: %s", hipGetErrorString(e));
    for (int c = 0; c < k; ++c) pi_host[c] = (double)host.best_pi[c];
    if (misplaced) *misplaced = (int64_t)(host.best_imb / 2.0f + 0.5f);
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

int ure_ot_centroids_members(const float *X, const int32_t *order, const int64_t *off, int64_t n, int k, int d, float *C, int32_t *counts,
                             void *stream)
{
    URE_ARG(X && order && off && C && n > 0 && k > 0 && d > 0);
    const unsigned blocks = (unsigned)((k * d + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(ot_centroid_members_kernel, dim3(blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), X, order, off, k, d, C,
                       counts);
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
