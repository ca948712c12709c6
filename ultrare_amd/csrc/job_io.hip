// job_io.hip -- table traffic of a request that is not training: the start tables of all shards of a job go from where the host's
// draws were uploaded ([rows][k], dense) into the job's padded tables ([rows][d]) in ONE launch (a request of 5 shards made 20
// strided torch copies here, each a launch and a piece of Python under the GIL beside busy worker threads).
#include "ure_internal.h"

namespace {

constexpr int kCopyBatch = 48;

struct copy_entry {
    const float *src;
    float *dst, *dst2;
    int64_t rows;
};

struct copy_batch {
    copy_entry e[kCopyBatch];
    int32_t k, d;
};

__global__ __launch_bounds__(256) void copy_rows_batch_kernel(copy_batch B)
{
    const copy_entry E = B.e[blockIdx.y];
    const int64_t total = E.rows * B.k;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / B.k;
        const int64_t at = row * B.d + (i - row * B.k);
        const float v = E.src[i];
        E.dst[at] = v;
        if (E.dst2) E.dst2[at] = v;
    }
}

struct sse_entry {
    const float *sse;
    int64_t n_user;
};

struct sse_batch {
    sse_entry e[kCopyBatch];
    int32_t epochs;
};

// out[shard][epoch] = sum over the users of sse[epoch][user], in double, in ONE fixed order (thread t adds the elements t, t + 256, ...;
// the 256 partial sums are folded pairwise), so that whoever asks -- one shard or all of a request -- reads the same bits.
__global__ __launch_bounds__(256) void epoch_sse_kernel(sse_batch B, double *out)
{
    __shared__ double part[256];
    const sse_entry E = B.e[blockIdx.y];
    const float *row = E.sse + (int64_t)blockIdx.x * E.n_user;
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < E.n_user; i += 256) acc += (double)row[i];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[(int64_t)blockIdx.y * B.epochs + blockIdx.x] = part[0];
}

}  // namespace

extern "C" int ure_epoch_sse_batch(int32_t n, const float *const *sse, const int64_t *n_user, int32_t epochs, double *out, void *stream)
{
    if (n < 0 || epochs < 0 || (n > 0 && epochs > 0 && (!sse || !n_user || !out))) return ure::fail(-1, "ure_epoch_sse_batch: bad arguments");
    if (epochs == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    for (int32_t at = 0; at < n; at += kCopyBatch) {
        sse_batch B;
        const int32_t m = std::min(kCopyBatch, n - at);
        for (int32_t i = 0; i < m; ++i) {
            if (!sse[at + i] || n_user[at + i] < 0) return ure::fail(-1, "ure_epoch_sse_batch: shard %d: bad arguments", at + i);
            B.e[i] = sse_entry{sse[at + i], n_user[at + i]};
        }
        B.epochs = epochs;
        hipLaunchKernelGGL(epoch_sse_kernel, dim3((unsigned)epochs, (unsigned)m), dim3(256), 0, st, B, out + (int64_t)at * epochs);
        URE_HIP(hipGetLastError());
    }
    return 0;
}

extern "C" int ure_copy_rows_batch(int32_t n, const float *const *src, float *const *dst, float *const *dst2, const int64_t *rows, int32_t k,
                                   int32_t d, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (n < 0 || (n > 0 && (!src || !dst || !rows)) || k <= 0 || d < k) return ure::fail(-1, "ure_copy_rows_batch: bad arguments");
    for (int32_t at = 0; at < n; at += kCopyBatch) {
        copy_batch B;
        const int32_t m = std::min(kCopyBatch, n - at);
        int64_t most = 0;
        for (int32_t i = 0; i < m; ++i) {
            if (!src[at + i] || !dst[at + i] || rows[at + i] < 0) return ure::fail(-1, "ure_copy_rows_batch: table %d: bad arguments", at + i);
            B.e[i] = copy_entry{src[at + i], dst[at + i], dst2 ? dst2[at + i] : nullptr, rows[at + i]};
            most = std::max(most, rows[at + i] * k);
        }
        B.k = k;
        B.d = d;
        if (most == 0) continue;
        const unsigned bx = (unsigned)std::min<int64_t>((most + 1023) / 1024, 1024);
        hipLaunchKernelGGL(copy_rows_batch_kernel, dim3(bx, (unsigned)m), dim3(256), 0, st, B);
        URE_HIP(hipGetLastError());
    }
    return 0;
}
