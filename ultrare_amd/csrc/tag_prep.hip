// tag_prep.hip -- standalone launches of the per-epoch batch-tag phases (tag_prep.h) for the
// cases the step kernel cannot carry them as riders: epoch 0 (nothing to ride on), shards with
// fewer than 3 steps per epoch, and shards too large for the LDS partition (plain scatter).
#include "tag_prep.h"

namespace ure {

// The epoch whose tags a standalone launch at `tick` prepares for this shard (-1: none).  pass 0: the epoch that starts at the tick
// (everything but touch_mode 2: when no riders could do it; touch_mode 2: epoch 0 only).  pass 1 (touch_mode 2 only): the epoch
// AFTER the one that starts -- at tick 0 and for shards without riders.
__device__ __forceinline__ int standalone_epoch(const ure_shard_t &S, int64_t tick, int pass)
{
    const int steps = (S.N + S.batch - 1) / S.batch;
    if (tick >= (int64_t)steps * S.epochs || tick % steps != 0) return -1;      // not an epoch start
    const bool riders = tag_riders(S);
    const int epoch = (int)(tick / steps);
    if (S.touch_mode == 2) {
        if (pass == 0) return tick == 0 ? 0 : -1;
        return (tick == 0 || !riders) && epoch + 1 < S.epochs ? epoch + 1 : -1;
    }
    if (pass != 0) return -1;
    if (tick != 0 && riders) return -1;                                         // riders did it
    return epoch;
}

__global__ __launch_bounds__(kBlock) void tag_partition_kernel(const ure_shard_t *__restrict__ shards, int64_t tick, int pass)
{
    __shared__ __attribute__((aligned(16))) char lds[kTagLds];
    const ure_shard_t &S = shards[blockIdx.y];
    const int epoch = standalone_epoch(S, tick, pass);
    if (epoch < 0 || !tag_partitioned(S.N) || (int)blockIdx.x >= tag_ranges(S.N)) return;
    tag_partition(S, epoch, (int)blockIdx.x, lds);
}

__global__ __launch_bounds__(kBlock) void tag_collect_kernel(const ure_shard_t *__restrict__ shards, int64_t tick, int pass)
{
    __shared__ __attribute__((aligned(16))) char lds[kTagLds];
    const ure_shard_t &S = shards[blockIdx.y];
    const int epoch = standalone_epoch(S, tick, pass);
    if (epoch < 0 || !tag_partitioned(S.N) || (int)blockIdx.x >= tag_ranges(S.N)) return;
    tag_collect(S, (int)blockIdx.x, lds);
}

// Shards with more than kMaxRanges ranges: plain scatter into file order (one 64-byte
// memory-side write per entry; such shards have hundreds of steps per epoch to amortise it).
__global__ __launch_bounds__(kBlock) void tag_scatter_kernel(const ure_shard_t *__restrict__ shards, int64_t tick, int pass)
{
    const ure_shard_t &S = shards[blockIdx.y];
    const int epoch = standalone_epoch(S, tick, pass);
    if (epoch < 0 || tag_partitioned(S.N) || S.file_tags) return;
    const int n = S.N;
    const int32_t *__restrict__ perm = S.perm + (size_t)epoch * n;
    uint16_t *__restrict__ file_tag = S.file_tag;
    const BatchOf batch_of(S.batch);
    for (int b = blockIdx.x * kBlock + threadIdx.x; b < n; b += gridDim.x * kBlock) {
        const int j = ldg(perm + b);
        if ((unsigned)j < (unsigned)n) stg(file_tag + j, (uint16_t)batch_of(b));
    }
}

__global__ __launch_bounds__(kBlock) void tag_derive_kernel(const ure_shard_t *__restrict__ shards, int64_t tick, int pass)
{
    const ure_shard_t &S = shards[blockIdx.y];
    const int epoch = standalone_epoch(S, tick, pass);
    if (epoch < 0) return;
    tag_derive(S, epoch, (int)blockIdx.x, (int)gridDim.x);
}

// Host: which standalone passes (bit 0, bit 1: see standalone_epoch) does any shard of the job need at `tick`?
int tag_prep_needed(const ure_job *job, int64_t tick)
{
    int need = 0;
    for (const ure_shard_t &S : job->host) {
        const int64_t steps = ((int64_t)S.N + S.batch - 1) / S.batch;
        if (tick >= steps * S.epochs || tick % steps != 0) continue;
        const bool riders = tag_riders(S);
        if (S.touch_mode == 2) {
            if (tick == 0) need |= 1;
            if ((tick == 0 || !riders) && tick / steps + 1 < S.epochs) need |= 2;
        } else if (tick == 0 || !riders) need |= 1;
    }
    return need;
}

void launch_tag_prep(const ure_job *job, int64_t tick, hipStream_t st, int pass)
{
    const unsigned n_shards = (unsigned)job->host.size();
    // (with host-made batch tags -- struct ure_shard: file_tags, the product's default -- only the slot-order gather is left)
    if (job->small_shards && !job->all_file_tags) {
        const unsigned ranges = (unsigned)tag_ranges(job->max_small_n);
        hipLaunchKernelGGL(tag_partition_kernel, dim3(ranges, n_shards), dim3(kBlock), 0, st, job->dev, tick, pass);
        hipLaunchKernelGGL(tag_collect_kernel, dim3(ranges, n_shards), dim3(kBlock), 0, st, job->dev, tick, pass);
    }
    if (job->large_shards && !job->all_file_tags) {
        const unsigned blocks = (unsigned)std::min((job->max_n + kBlock - 1) / kBlock, 4096);
        hipLaunchKernelGGL(tag_scatter_kernel, dim3(blocks, n_shards), dim3(kBlock), 0, st, job->dev, tick, pass);
    }
    const unsigned der_blocks = (unsigned)std::min(tag_derive_blocks(job->max_slots), 2048);
    hipLaunchKernelGGL(tag_derive_kernel, dim3(der_blocks, n_shards), dim3(kBlock), 0, st, job->dev, tick, pass);
}

}  // namespace ure
