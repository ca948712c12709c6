// tag_prep.h -- per-epoch batch tags for gfx950: which optimizer step of the epoch trains
// each interaction.  Device code shared by the step kernel (mf_train.hip), which carries
// these phases as extra workgroups, and by the standalone launches (tag_prep.hip).
//
// Replaces the batching of the reference's DataLoader (read.py:127-133): with the epoch's
// permutation perm (torch RandomSampler), batch s is perm[s*B : (s+1)*B], so interaction j
// trains in step inv[j] / B where inv is the inverse permutation.  The step kernel wants that
// number per slot of its row segments (ent_tag), in slot order.
//
// A direct scatter (tag[pos[perm[b]]] = b / B) costs one 64-byte memory-side write per 2-byte
// store on gfx950 -- stores leave the L2 as they are issued, partial lines are not merged
// (profiles/r01/NOTES.md, tools/microbench/scatter.hip: ~50 G stores/s whatever the element
// size) -- and that fabric traffic does not hide beside the step kernel (measured).  So the
// inverse is built as a radix partition whose global traffic is all full-line:
//   phase A  partition   workgroup c takes perm[c*2K : (c+1)*2K], bins the entries by the
//                        2K-range their target j falls in (LDS counting sort) and writes the
//                        reordered chunk, packed (j mod 2K | tag << 11), plus the bin offsets;
//   phase B  collect     workgroup r gathers bin r of every chunk (~2K entries in all), drops
//                        them into an LDS image of file_tag[r*2K : (r+1)*2K] and writes the
//                        image with full-line stores;
//   phase C  derive      ent_tag[p] = file_tag[ent_src[p]]: coalesced reads, L2-resident 2-byte
//                        gathers, full-line stores.
// While epoch e trains, its steps carry phases A, B and C for epoch e+1, each phase spread over a
// third of the epoch's steps (ent_tag is double-buffered by epoch parity); the kernel boundary
// between steps is the barrier between phases.  Epoch 0, shards with fewer than 3 steps per epoch, and shards beyond kMaxRanges
// ranges (2 M interactions; plain scatter) use standalone launches instead.
#pragma once
#include "ure_internal.h"

namespace ure {

constexpr int kRange = 2048;           // file indices per range = permutation entries per chunk
constexpr int kMaxRanges = 1024;
constexpr int kRangeBits = 11;
static_assert((1 << kRangeBits) == kRange, "range size");

__host__ __device__ inline int tag_ranges(int n) { return (n + kRange - 1) / kRange; }
__host__ __device__ inline bool tag_partitioned(int n) { return tag_ranges(n) <= kMaxRanges; }

// LDS needs of the phases, in bytes (the step kernel overlays them on its own arrays)
constexpr int kTagLdsA = (kMaxRanges + 1) * 4 + kRange * 4 + kBlock * 4;
constexpr int kTagLdsB = kRange * 2 + 2 * kMaxRanges * 4;
constexpr int kTagLds = kTagLdsA > kTagLdsB ? kTagLdsA : kTagLdsB;

// b / batch without an integer division per entry: M = ceil(2^62 / batch); (b * M) >> 62 is
// exact while b * batch < 2^62.
struct BatchOf {
    unsigned long long M;
    __device__ explicit BatchOf(int batch) : M((0x4000000000000000ull + (unsigned long long)batch - 1) / (unsigned long long)batch) {}
    __device__ __forceinline__ int operator()(int b) const { return (int)__umul64hi((unsigned long long)b << 2, M); }
};

// Phase A, workgroup `c` of tag_ranges(N) (256 threads).
__device__ inline void tag_partition(const ure_shard_t &S, int epoch, int c, char *lds)
{
    if (S.file_tags) return;                                            // the host supplies the file tags (struct ure_shard)
    int *cnt = reinterpret_cast<int *>(lds);                            // [kMaxRanges + 1]
    uint32_t *buf = reinterpret_cast<uint32_t *>(cnt + kMaxRanges + 1);  // [kRange]
    int *scan = reinterpret_cast<int *>(buf + kRange);                  // [kBlock]
    const int n = S.N;
    const int n_ranges = tag_ranges(n);
    const int32_t *__restrict__ perm = S.perm + (size_t)epoch * n;
    uint32_t *__restrict__ stage = S.inv_stage;
    int32_t *__restrict__ off = S.inv_off + (size_t)c * (n_ranges + 1);
    const BatchOf batch_of(S.batch);
    const int tid = threadIdx.x;
    const int b_lo = c * kRange, b_hi = min(b_lo + kRange, n);

    for (int t = tid; t <= n_ranges; t += kBlock) cnt[t] = 0;
    __syncthreads();
    constexpr int PER = kRange / kBlock;               // 8 entries per thread
    int jv[PER], rank[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int b = b_lo + k * kBlock + tid;
        jv[k] = b < b_hi ? ldg(perm + b) : -1;
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        rank[k] = 0;
        if ((unsigned)jv[k] < (unsigned)n) rank[k] = atomicAdd(&cnt[jv[k] >> kRangeBits], 1);   // malformed entries are dropped
        else jv[k] = -1;
    }
    __syncthreads();
    // exclusive prefix of cnt[0..n_ranges): four consecutive counters per thread + a block scan
    int loc[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int t = tid * 4 + k;
        loc[k] = t < n_ranges ? cnt[t] : 0;
        sum += loc[k];
    }
    scan[tid] = sum;
    __syncthreads();
    for (int o = 1; o < kBlock; o <<= 1) {
        const int v = tid >= o ? scan[tid - o] : 0;
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
    }
    int run = scan[tid] - sum;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int t = tid * 4 + k;
        if (t < n_ranges) cnt[t] = run;
        run += loc[k];
    }
    if (tid == kBlock - 1) cnt[n_ranges] = scan[tid];
    __syncthreads();
    for (int t = tid; t <= n_ranges; t += kBlock) stg(off + t, cnt[t]);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        if (jv[k] >= 0) {
            const int b = b_lo + k * kBlock + tid;
            buf[cnt[jv[k] >> kRangeBits] + rank[k]] = (uint32_t)(jv[k] & (kRange - 1)) | ((uint32_t)batch_of(b) << kRangeBits);
        }
    }
    __syncthreads();
    const int total = cnt[n_ranges];
    for (int t = tid; t < total; t += kBlock) stg(stage + (size_t)b_lo + t, buf[t]);
}

// Phase B, workgroup `r` of tag_ranges(N) (256 threads).
__device__ inline void tag_collect(const ure_shard_t &S, int r, char *lds)
{
    if (S.file_tags) return;
    uint16_t *img = reinterpret_cast<uint16_t *>(lds);                  // [kRange]
    int *lo = reinterpret_cast<int *>(img + kRange);                    // [kMaxRanges]
    int *hi = lo + kMaxRanges;                                          // [kMaxRanges]
    const int n = S.N;
    const int n_ranges = tag_ranges(n);
    const uint32_t *__restrict__ stage = S.inv_stage;
    const int32_t *__restrict__ off = S.inv_off;
    uint16_t *__restrict__ file_tag = S.file_tag;
    const int tid = threadIdx.x;
    for (int t = tid; t < kRange; t += kBlock) img[t] = 0xFFFFu;        // an index the permutation misses never trains
    for (int c = tid; c < n_ranges; c += kBlock) {
        lo[c] = ldg(off + (size_t)c * (n_ranges + 1) + r);
        hi[c] = ldg(off + (size_t)c * (n_ranges + 1) + r + 1);
    }
    __syncthreads();
    // bins are ~2K / n_ranges entries: a quarter wave per chunk keeps most lanes busy
    const int sub = tid & 15, part = tid >> 4;
    for (int c = part; c < n_ranges; c += kBlock / 16) {
        const uint32_t *__restrict__ src = stage + (size_t)c * kRange;
        for (int t = lo[c] + sub; t < hi[c]; t += 16) {
            const uint32_t e = ldg(src + t);
            img[e & (kRange - 1)] = (uint16_t)(e >> kRangeBits);
        }
    }
    __syncthreads();
    const int j0 = r * kRange;
    for (int t = tid; t < kRange && j0 + t < n; t += kBlock) stg(file_tag + j0 + t, img[t]);
}

// Which half (third) of ent_tag holds an epoch's tags, and how far ahead of training they are prepared.  touch_mode 2 (mf_touch.h,
// "masks one epoch ahead") needs the tags of epoch e + 1 complete when epoch e starts: they are prepared TWO epochs ahead, in three
// buffers.  Everything else: one epoch ahead, two buffers.
__host__ __device__ inline int tag_ahead(const ure_shard_t &S) { return S.touch_mode == 2 ? 2 : 1; }
__host__ __device__ inline size_t tag_buffer(const ure_shard_t &S, int epoch) { return (size_t)(S.touch_mode == 2 ? epoch % 3 : (epoch & 1)) * (size_t)S.n_slots; }

// Phase C, workgroup `blk` of `n_blk` (256 threads, eight slots per thread).
// Workgroup -> (shard, workgroup of the shard) on a 1-D grid of 8 n_sh `per` workgroups (per = ceil(workgroups per shard / 8)).  The dispatcher deals
// consecutive workgroup ids out to the 8 XCDs round robin and every XCD has its own L2: XCD x takes the slices [n_sh x, n_sh (x + 1)) of the 8 n_sh
// slices (slice r of shard k = its workgroups = r mod 8) ONE AFTER THE OTHER, so that with 8 m shards an XCD works through m whole shards in turn and what
// a shard's workgroups gather again and again -- its popular rows -- stays in one L2.  One shard: the identity.  For launches in which EVERY shard has
// work (the step kernel: 99.8 -> 96.8 us at configs[3]'s shape, k = 16); the epoch-start passes of touch_mode 3 lose by it -- the shards of such a job
// start their epochs at different ticks, and a shard alone on one XCD has an eighth of the chip (its scatter 57 -> 158 us; all 32 at once 793 -> 728).
struct WgMap { int shard, wg; };
__device__ __forceinline__ WgMap xcd_shard_map(unsigned id, unsigned n_sh, unsigned per)
{
    const unsigned x = id & 7u, j = id >> 3;
    const unsigned jq = j / per;
    const unsigned slice = n_sh * x + jq;
    return WgMap{(int)(slice >> 3), (int)((j - jq * per) * 8u + (slice & 7u))};
}

__device__ inline void tag_derive(const ure_shard_t &S, int epoch, int blk, int n_blk)
{
    const int32_t *__restrict__ ent_src = S.ent_src;
    const uint16_t *__restrict__ file_tag = S.file_tags ? S.file_tags + (size_t)epoch * S.N : S.file_tag;
    uint16_t *__restrict__ ent_tag = S.ent_tag + tag_buffer(S, epoch);
    const int64_t n8 = S.n_slots / 8;
    for (int64_t q = (int64_t)blk * kBlock + threadIdx.x; q < n8; q += (int64_t)n_blk * kBlock) {
        const int4 s0 = ldg_i4(ent_src + q * 8);
        const int4 s1 = ldg_i4(ent_src + q * 8 + 4);
        const int sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
        unsigned tg[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) tg[k] = sv[k] >= 0 ? (unsigned)ldg(file_tag + sv[k]) : 0xFFFFu;
        stg_u4(ent_tag + q * 8, make_uint4(tg[0] | (tg[1] << 16), tg[2] | (tg[3] << 16), tg[4] | (tg[5] << 16), tg[6] | (tg[7] << 16)));
    }
}

__host__ __device__ inline int tag_derive_blocks(int64_t n_slots) { return (int)((n_slots / 8 + kBlock - 1) / kBlock); }

// Per-shard constants derived once by ure_job_create (device array next to the descriptors), so
// that the step kernel's prologue does no integer division: a 64-bit division costs a wave ~150
// scalar instructions, and every workgroup of every launch starts with three of them otherwise.
struct shard_aux {
    uint64_t inv_steps;     // ceil(2^64 / steps), 0 for steps == 1: tick / steps = umul64hi(tick, inv_steps)
    int32_t steps;          // optimizer steps per epoch = ceil(N / B)
    int32_t ride_m;         // steps / 3, or 0 when the shard's tags are prepared by standalone launches
    int32_t ride_ab;        // workgroups of phase A (and of B) carried by one step = ceil(ranges / m)
    int32_t ride_c;         // the same for phase C
    int32_t ride_only_c;    // 1: epochs of fewer than three steps with host-made tags -- phase C alone, spread over all steps of the epoch
    int32_t ranges;         // tag_ranges(N)
    int32_t derive_blocks;  // tag_derive_blocks(n_slots)
    // touch mode (mf_touch.h); library-owned device memory, NULL otherwise.  An epoch is cut into WINDOWS of 64 steps
    // (the last one shorter); window w of epoch e has the global index e * windows + w.
    int32_t windows;               // windows per epoch = ceil(steps / 64)
    unsigned long long *mask[2];   // [n_user + n_item] steps of the window in which the row is trained (bit = step - 64 w), by global window parity
    const float4 *ptab;            // [epochs][65] A_e^j = {p11, p12, p21, p22}: j optimizer steps without a gradient at epoch e's lr
    unsigned long long *unit_mask; // [n_units] the current window's mask of each work unit's row
    unsigned long long *unit_own;  // [n_units] the steps of the window in which the UNIT's own slots have interactions (a subset of its row's)
    unsigned long long *pass_mask; // [n_slots / 8] (epochs of several windows only, else NULL) the same for every scan pass of a work unit of several
                                   // passes, at index (first slot of the pass) / 8: the unit's lane group skips the passes without a slot of the step
    // touch_mode 2 ("masks one epoch ahead"): the three work-order arrays once per epoch parity (an epoch's are written at the start of
    // the epoch before it), and for the owners of the current epoch the row's first own step of the NEXT epoch (255: none)
    unsigned long long *ahead_masks[2];   // [2 n_um + n_sm] each: unit_mask | unit_own | sched_mask
    uint8_t *unit_nf, *sched_nf;          // [n_um], [n_sm]
    int32_t n_um, n_sm;
    unsigned long long *sched_mask;// [n_active - n_multi] the same for the single-pass rows, in schedule order
    // touch_mode 3 ("indexed", mf_index.h): the per-step slot index of the current epoch, rebuilt at every epoch start
    int32_t ptab_stride;           // entries of ptab per epoch (65; touch_mode 3: steps + 1)
    int32_t idx_words;             // mask words per row and epoch = ceil(steps / 63)
    int32_t idx_chunks;            // ceil(n_slots / idx_chunk): the chunks of the sort by step
    int32_t idx_chunk;             // slots per chunk: 4,096, or 1,024 when no shard of the job has more than 63 steps per epoch
    int32_t idx_hw;                // workgroups of a step launch reserved for heavy rows (upper bound)
    int32_t idx_light;             // upper bound of a step's light items = min(2 B, n_active)
    int32_t *grp_row;              // [n_slots / 8][2] {schedule index, row id} of the row that owns the group of 8 slots (static)
    unsigned long long *W;         // [idx_words][n_user + n_item] bit b of word w = the row is trained in step 63 w + b; bit 63 = the buffer
                                   // its weights are in at step 63 w
    uint8_t *end_par[2];           // [n_user + n_item] buffer of the row at the end of an epoch, by epoch parity
    uint16_t *first_step;          // [n_user + n_item] the row's first step of the epoch (steps: none)
    uint16_t *next_first;          // [idx_words][n_user + n_item] the row's first step in the words after w (steps: none)
    uint32_t *hist;                // [idx_chunks][steps + 1] slots per (chunk, step) -> after the scan: where the chunk's run of the step starts
    uint32_t *seg;                 // [32][steps + 1] scan scratch
    uint32_t *step_begin;          // [steps + 2] first sorted slot of each step; [steps] = sorted slots in all
    uint4 *sslot;                  // [n_slots] slots sorted by (step, row, file order): {opposite id | buffer << 31, rating, row id, step | class << 16 |
                                   // the row's own buffer at the step << 18 | steps until its next own step << 19 (mf_index.h: idx_own_bits)}
    unsigned long long *runflag;   // [n_slots / 64 + 1] bit = the sorted slot starts a (step, row) run
    uint32_t *blk_cnt;             // [n_slots / 2048 + 2] runs that start in each block of 2048 sorted slots -> exclusive prefix
    int4 *items;                   // [2 N + 1] one per (step, row) run, steps ascending: {row id | buffer << 31, first sorted slot, end, gap | class << 16}
    uint4 *items2;                 // [2 N + 1] the first two sorted slots of every item's run: {opposite id | buffer << 31, rating, the same of the second}
    uint32_t *step_item;           // [steps + 2] first item of each step
    uint32_t *heavy_cnt;           // [steps] items of the step that whole workgroups take (a prefix of its items)
    uint32_t *heavy_cum;           // [steps][257] workgroups of the step's heavy items, cumulative
    uint32_t *heavy_map;           // [steps][idx_hw] workgroup of the heavy range -> item of the prefix | part << 9 | parts << 20
    uint32_t *heavy_wg;            // [steps] workgroups the step's heavy items take in all
    uint4 *step_desc;              // [steps] {first item, end, heavy items, heavy workgroups}: what a workgroup of the step kernel reads first
    float *partial;                // [idx_hw][d + 4] partial gradient sums of the rows split over several workgroups
};

constexpr int kTouchWindow = 64;            // steps a 64-bit row mask describes
constexpr int kTouchWindowBits = 6;
// global index of the last window of `epoch` (the window whose masks say in which buffer a row's weights are at the epoch's end)
__host__ __device__ inline int64_t touch_last_window(const shard_aux &a, int64_t epoch) { return (epoch + 1) * a.windows - 1; }

// Does the shard's tag preparation ride on its steps although an epoch has fewer than three of them?  (Host-made tags only: no partition phases.)
__host__ __device__ inline bool tag_short_riders(const ure_shard_t &S)
{
    const int64_t steps = ((int64_t)S.N + S.batch - 1) / S.batch;
    return steps < 3 && S.file_tags != nullptr && S.touch_mode != 3;
}
// ... and does it ride at all (otherwise: standalone launches at every epoch start)
__host__ __device__ inline bool tag_riders(const ure_shard_t &S)
{
    const int64_t steps = ((int64_t)S.N + S.batch - 1) / S.batch;
    return ((steps >= 3 && tag_partitioned(S.N)) || tag_short_riders(S)) && S.touch_mode != 3;
}

inline shard_aux make_shard_aux(const ure_shard_t &S)
{
    shard_aux a{};
    a.steps = (int32_t)(((int64_t)S.N + S.batch - 1) / S.batch);
    a.inv_steps = a.steps == 1 ? 0 : ~0ull / (uint64_t)a.steps + 1;
    a.ranges = tag_ranges(S.N);
    a.derive_blocks = tag_derive_blocks(S.n_slots);
    a.windows = (a.steps + kTouchWindow - 1) / kTouchWindow;
    a.ride_m = a.steps >= 3 && tag_partitioned(S.N) && S.touch_mode != 3 ? a.steps / 3 : 0;      // (touch_mode 3 prepares every epoch's tags at its start)
    if (a.ride_m) {
        a.ride_ab = (a.ranges + a.ride_m - 1) / a.ride_m;
        a.ride_c = (a.derive_blocks + a.ride_m - 1) / a.ride_m;
    } else if (tag_short_riders(S)) {
        // epochs of one or two steps with host-made batch tags (BASELINE.json configs[4]: 16 shards of ~56 k rows, 2 steps per epoch):
        // only the slot-order gather is left of the preparation, and every step carries its share of it -- as a launch of its
        // own per epoch it was 19 % of that configuration's device time (profiles/r04/cfg4_d16_kernel_stats.csv before / after)
        a.ride_only_c = 1;
        a.ride_c = (a.derive_blocks + a.steps - 1) / a.steps;
    }
    return a;
}

__host__ __device__ inline int64_t epoch_of(const shard_aux &a, int64_t tick)
{
#ifdef __HIP_DEVICE_COMPILE__
    return a.inv_steps ? (int64_t)__umul64hi((uint64_t)tick, a.inv_steps) : tick;
#else
    return tick / a.steps;
#endif
}

// The share of the NEXT epoch's tag preparation that step `s` of an epoch carries as extra
// workgroups: with m = steps / 3, phase p runs in steps [p m, (p+1) m), ceil(workgroups / m) of its
// workgroups in each, so that no single launch carries a whole phase.
struct TagRide {
    int phase;      // 0 partition, 1 collect, 2 derive; -1 nothing
    int first;      // first workgroup of the phase this step runs
    int count;
};
__host__ __device__ inline TagRide tag_ride(const shard_aux &a, int s, bool has_next)
{
    TagRide r{-1, 0, 0};
    if (a.ride_only_c) {
        if (!has_next) return r;
        r.phase = 2;
        r.first = s * a.ride_c;
        r.count = max(0, min(a.ride_c, a.derive_blocks - r.first));
        return r;
    }
    const int m = a.ride_m;
    if (!has_next || m == 0 || s >= 3 * m) return r;
    const int p = s >= 2 * m ? 2 : s >= m ? 1 : 0;
    const int q = s - p * m;
    const int c = p < 2 ? a.ride_ab : a.ride_c, nb = p < 2 ? a.ranges : a.derive_blocks;
    r.phase = p;
    r.first = q * c;
    r.count = max(0, min(c, nb - r.first));
    return r;
}

}  // namespace ure
