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

}  // namespace

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
