// tag_prep.hip -- standalone launches of the per-epoch batch-tag phases (tag_prep.h) for the
// cases the step kernel cannot carry them as riders: epoch 0 (nothing to ride on), shards with
// fewer than 3 steps per epoch, and shards too large for the LDS partition (plain scatter).
#include "tag_prep.h"

namespace ure {

__device__ __forceinline__ int standalone_epoch(const ure_shard_t &S, int64_t tick)
{
    const int steps = (S.N + S.batch - 1) / S.batch;
    if (tick >= (int64_t)steps * S.epochs || tick % steps != 0) return -1;      // not an epoch start
    if (tick != 0 && steps >= 3 && tag_partitioned(S.N)) return -1;             // riders did it
    return (int)(tick / steps);
}

__global__ __launch_bounds__(kBlock) void tag_partition_kernel(const ure_shard_t *__restrict__ shards, int64_t tick)
{
    __shared__ __attribute__((aligned(16))) char lds[kTagLds];
    const ure_shard_t &S = shards[blockIdx.y];
    const int epoch = standalone_epoch(S, tick);
    if (epoch < 0 || !tag_partitioned(S.N) || (int)blockIdx.x >= tag_ranges(S.N)) return;
    tag_partition(S, epoch, (int)blockIdx.x, lds);
}

__global__ __launch_bounds__(kBlock) void tag_collect_kernel(const ure_shard_t *__restrict__ shards, int64_t tick)
{
    __shared__ __attribute__((aligned(16))) char lds[kTagLds];
    const ure_shard_t &S = shards[blockIdx.y];
    const int epoch = standalone_epoch(S, tick);
    if (epoch < 0 || !tag_partitioned(S.N) || (int)blockIdx.x >= tag_ranges(S.N)) return;
    tag_collect(S, (int)blockIdx.x, lds);
}

// Shards with more than kMaxRanges ranges: plain scatter into file order (one 64-byte
// memory-side write per entry; such shards have hundreds of steps per epoch to amortise it).
__global__ __launch_bounds__(kBlock) void tag_scatter_kernel(const ure_shard_t *__restrict__ shards, int64_t tick)
{
    const ure_shard_t &S = shards[blockIdx.y];
    const int epoch = standalone_epoch(S, tick);
    if (epoch < 0 || tag_partitioned(S.N)) return;
    const int n = S.N;
    const int32_t *__restrict__ perm = S.perm + (size_t)epoch * n;
    uint16_t *__restrict__ file_tag = S.file_tag;
    const BatchOf batch_of(S.batch);
    for (int b = blockIdx.x * kBlock + threadIdx.x; b < n; b += gridDim.x * kBlock) {
        const int j = ldg(perm + b);
        if ((unsigned)j < (unsigned)n) stg(file_tag + j, (uint16_t)batch_of(b));
    }
}

__global__ __launch_bounds__(kBlock) void tag_derive_kernel(const ure_shard_t *__restrict__ shards, int64_t tick)
{
    const ure_shard_t &S = shards[blockIdx.y];
    const int epoch = standalone_epoch(S, tick);
    if (epoch < 0) return;
    tag_derive(S, epoch, (int)blockIdx.x, (int)gridDim.x);
}

// Host: does any shard of the job need a standalone preparation at `tick`?
bool tag_prep_needed(const ure_job *job, int64_t tick)
{
    for (const ure_shard_t &S : job->host) {
        const int64_t steps = ((int64_t)S.N + S.batch - 1) / S.batch;
        if (tick >= steps * S.epochs || tick % steps != 0) continue;
        if (tick == 0 || steps < 3 || !tag_partitioned(S.N)) return true;
    }
    return false;
}

void launch_tag_prep(const ure_job *job, int64_t tick, hipStream_t st)
{
    const unsigned n_shards = (unsigned)job->host.size();
    if (job->small_shards) {
        const unsigned ranges = (unsigned)tag_ranges(job->max_small_n);
        hipLaunchKernelGGL(tag_partition_kernel, dim3(ranges, n_shards), dim3(kBlock), 0, st, job->dev, tick);
        hipLaunchKernelGGL(tag_collect_kernel, dim3(ranges, n_shards), dim3(kBlock), 0, st, job->dev, tick);
    }
    if (job->large_shards) {
        const unsigned blocks = (unsigned)std::min((job->max_n + kBlock - 1) / kBlock, 4096);
        hipLaunchKernelGGL(tag_scatter_kernel, dim3(blocks, n_shards), dim3(kBlock), 0, st, job->dev, tick);
    }
    const unsigned der_blocks = (unsigned)std::min(tag_derive_blocks(job->max_slots), 2048);
    hipLaunchKernelGGL(tag_derive_kernel, dim3(der_blocks, n_shards), dim3(kBlock), 0, st, job->dev, tick);
}

}  // namespace ure
