// mf_index.h -- touch_mode 3, "indexed": touch mode (mf_touch.h) for epochs of hundreds of optimizer steps (full MF at 25 M rows:
// 750 steps per epoch, config.py:182-188 with batch 30,000), where a step trains 30,000 of a shard's 22.5 M interactions.
//
// In 64-step windows (touch_mode 1) every work unit of every row is still looked at in every step -- its masks are tested, the
// heaviest item row has 1.9 M slots that 16 lane groups scan to find the ~2,500 of the step -- and every window start costs two
// dense passes.  Here the slots of an epoch are SORTED BY STEP once, at the epoch's start:
//   sslot  the slots in (step, row, file order) order -- a stable counting sort of the row-major slot array by batch tag, so a
//          step's slots are contiguous, a row's slots of the step are contiguous inside them, and rows keep the schedule's order
//          (heaviest first);
//   items  one entry per (step, row) run: {row id | the buffer its weights are in, first sorted slot, end, the number of steps
//          until the row is trained next (or the epoch ends) | class};
// and step s launches over exactly the items of step s: a lane group per item loads the entry, then -- in one memory level -- the
// row's weights, its momentum and the run's slots, gathers the opposite rows, applies the optimizer and advances the row to its
// next own step in closed form (next-touch form, as in mf_touch.h).  Rows whose runs are long get whole workgroups: "heavy" rows
// (16 or more slots per step on average) one, "split" rows (384 or more) one per 256 slots -- a single compute unit pulls
// ~30 GB/s of gathered rows, the top item's 2,500 rows of 512 bytes would take it 44 us -- whose partial sums a second, tiny launch adds
// in part order and applies.  No atomics on the data path: runs, partial sums and their order are fixed by the sort, results are
// bitwise reproducible.
//
// The epoch start (standalone launches, mf_index launch list in index_epoch_start):
//   masks    W[w][row]: bit b = the row is trained in step 63 w + b.  A workgroup takes 2,048 consecutive slots (at most 256 rows),
//            ORs the tags into an LDS bitmap and stores a row's words whole; rows that straddle chunks are ORed into memory.
//   parity   bit 63 of every word = the buffer the row's weights are in at the word's first step (a row alternates between the two
//            weight buffers with each of its own steps); the row's first step; its buffer at the epoch's end.  By row id.
//   sort     histogram per (chunk of 4,096 slots -- 1,024 in jobs of short epochs --, step) -> exclusive scan over (step, chunk) -> the
//            chunk is sorted in LDS and leaves a step's stretch at a time (equal tags inside a batch of 64 are ranked by lane with
//            ballots: stable).  Every slot travels with the buffer its OPPOSITE row is in at that step (a gather from W in the scatter)
//            and with its OWN row's buffer and the steps until the row's next own step (idx_own_bits).
//   mark     the bitmap of run starts of the sorted slots.
//   items    scan of the run starts -> one entry per run (its buffer and gap copied from the run's first slot); per step the heavy
//            prefix and its workgroup counts.
//   advance  every active row from "valid at the end of the last epoch" to "valid at its first step", into buffer 0 (dense, once
//            per EPOCH; the windows of touch_mode 1 pay it every 64 steps).
//
// Limits: at most 1,008 steps per epoch (16 mask words of 63 steps); the tables are readable at epoch boundaries (as touch_mode 1).
// Included by mf_train.hip after mf_touch.h.
#pragma once

namespace ure {

constexpr int kIdxWin = 63;                     // steps per mask word
constexpr int kIdxMaxWords = 16;
constexpr int kIdxMaxSteps = kIdxWin * kIdxMaxWords;
constexpr int kIdxChunk = 4096;                 // slots of a sort chunk: 64 batches of 64 (shard_aux: idx_chunk -- 1,024 in jobs of short epochs, idx_scatter_short_kernel)
constexpr int kIdxSeg = 32;                     // segments of the scan over chunks
constexpr int kIdxScanBatch = 16;               // counts a thread of the scan has in flight
constexpr int kIdxFlagBlock = 2048;             // sorted slots per workgroup of the mark / emit passes
constexpr int kIdxHeavyMax = 256;               // rows that whole workgroups take (a prefix of the schedule)
constexpr int kIdxPart = 256;                   // slots per workgroup of a split row
constexpr unsigned long long kIdxBits = ~(1ull << 63);
constexpr int kIdxLight = 0, kIdxHeavy = 1, kIdxSplit = 2;

#ifndef URE_INDEX_KGB
#define URE_INDEX_KGB 4                         // rows a lane group gathers together
#endif
#ifndef URE_INDEX_WAVES
#define URE_INDEX_WAVES 4
#endif
// (Rows of 64 bytes, d = 16: configs[3]'s shape, epoch time with (rows gathered together, waves per SIMD) = (4, 4) 4.50-4.53 ms, (8, 4) 5.56,
// (4, 8) 4.50, (1, 8) 4.51, (2, 4) 4.47, (2, 6) 4.48, (2, 7) 4.48, (3, 8) 4.49, (2, 8) 4.34-4.40 -- the one setting that wins does so at 64 VGPRs
// with 4 of them spilled; not taken.  tools/r4_short_ab.sh, three boxes.)

// the epoch that starts at `tick` for this shard, or -1
__device__ __forceinline__ int idx_epoch_start(const ure_shard_t &S, const shard_aux &A, int64_t tick)
{
    if (S.touch_mode != 3 || tick >= (int64_t)A.steps * S.epochs) return -1;
    const int epoch = (int)epoch_of(A, tick);
    return tick == (int64_t)epoch * A.steps ? epoch : -1;
}

__device__ __forceinline__ int idx_buffer_at(unsigned long long word, int b) { return (int)(word >> 63) ^ (__popcll(word & kIdxBits & mask_below(b)) & 1); }

// What a sorted slot carries of its OWN row's step, in the bits above the class of the record's last word (every slot of a (step, row) run the same;
// idx_emit_kernel copies them into the run's item): the buffer the row's weights are in at the step, and the steps that pass until the row is trained
// next (none left in the epoch: until it ends).  The scatter reads them where they are cheap -- its slots come row by row, a wavefront's loads fall into
// a handful of lines -- where the items, in (step, row) order, gathered a mask word and a next_first entry each (28 M of each at the 25 M shape).
constexpr int kIdxOwnBufBit = 18, kIdxGapShift = 19;       // (step: bits 0-9 of 16, class: 16-17, gap: at most 1,007)
__device__ __forceinline__ unsigned idx_own_pack(unsigned long long word, int nxt, unsigned tag)
{
    const int b = (int)(tag % kIdxWin);
    const unsigned long long rest = (word & kIdxBits) >> (b + 1);
    const int gap = rest ? __ffsll((long long)rest) - 1 : nxt - (int)tag - 1;
    return (unsigned)idx_buffer_at(word, b) << kIdxOwnBufBit | (unsigned)gap << kIdxGapShift;
}
__device__ __forceinline__ unsigned idx_own_bits(const shard_aux &A, int n_all, int row, unsigned tag)
{
    const size_t at = (size_t)(tag / kIdxWin) * n_all + row;
    // (epochs of one mask word: no later word, the row waits until the epoch ends)
    return idx_own_pack(ldg(A.W + at), A.idx_words > 1 ? (int)ldg(A.next_first + at) : A.steps, tag);
}

// ---- once per job: which row owns every group of 8 slots (segments lie in schedule order, 8-aligned)
__global__ __launch_bounds__(kBlock) void idx_grp_row_kernel(const ure_shard_t *__restrict__ shards, const shard_aux *__restrict__ aux)
{
    const ure_shard_t &S = shards[blockIdx.y];
    const shard_aux &A = aux[blockIdx.y];
    if (S.touch_mode != 3) return;
    const int64_t n_grp = S.n_slots / 8;
    for (int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x; g < n_grp; g += (int64_t)gridDim.x * kBlock) {
        int lo = 0, hi = S.n_active - 1;                 // the last active row whose segment starts at or before slot 8 g
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if ((int64_t)ldg(S.sched + 4 * (size_t)mid + 1) <= 8 * g) lo = mid; else hi = mid - 1;
        }
        const int row_id = ldg(S.sched + 4 * (size_t)lo);
        stg(A.grp_row + 2 * g, lo);                      // {schedule index, row id}: whoever walks the slots needs both, in one load
        stg(A.grp_row + 2 * g + 1, row_id);
    }
}

// ---- epoch start 1: the step masks of every row.  Workgroup c takes the slots [2048 c, 2048 (c + 1)) (at most 256 rows: 32 KB of LDS, five
// workgroups per CU -- with the sort's chunks of 4,096 it was two, and the kernel a chain of memory latencies: 118 us per epoch at the 25 M shape).
constexpr int kIdxMaskChunk = 2048;
__global__ __launch_bounds__(kBlock) void idx_masks_kernel(const ure_shard_t *__restrict__ shards, const shard_aux *__restrict__ aux, int64_t tick)
{
    __shared__ unsigned long long bits[(kIdxMaskChunk / 8) * kIdxMaxWords];   // [rows of the chunk][words]
    const ure_shard_t &S = shards[blockIdx.y];
    const shard_aux &A = aux[blockIdx.y];
    const int epoch = idx_epoch_start(S, A, tick);
    const int64_t n_grp = S.n_slots / 8;
    const int64_t g_lo = (int64_t)blockIdx.x * (kIdxMaskChunk / 8), g_hi = min(g_lo + kIdxMaskChunk / 8, n_grp);
    if (epoch < 0 || g_lo >= n_grp) return;
    const int words = A.idx_words, steps = A.steps;
    const int idx0 = ldg(A.grp_row + 2 * g_lo);
    const int n_rows = ldg(A.grp_row + 2 * (g_hi - 1)) - idx0 + 1;
    for (int t = threadIdx.x; t < n_rows * words; t += kBlock) bits[t] = 0ull;
    __syncthreads();
    const uint16_t *__restrict__ ent_tag = S.ent_tag + tag_buffer(S, epoch);
    for (int64_t g = g_lo + threadIdx.x; g < g_hi; g += kBlock) {
        const int local = ldg(A.grp_row + 2 * g) - idx0;
        const uint4 t4 = ldg_u4(ent_tag + g * 8);
        const unsigned tw[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned st = (tw[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu;
            if (st < (unsigned)steps) atomicOr(&bits[local * words + (int)(st / kIdxWin)], 1ull << (st % kIdxWin));
        }
    }
    __syncthreads();
    const int n_all = S.n_user + S.n_item;
    for (int t = threadIdx.x; t < n_rows * words; t += kBlock) {
        const int local = t / words, w = t - local * words;
        const int4 sc = ldg_i4(S.sched + 4 * (size_t)(idx0 + local));
        const bool inside = (int64_t)sc.y >= g_lo * 8 && (int64_t)sc.z <= g_hi * 8;
        unsigned long long *dst = A.W + (size_t)w * n_all + sc.x;
        if (inside) stg(dst, bits[t]);                                   // (every word of the row, zeros too: nothing to clear)
        else if (bits[t]) atomicOr(dst, bits[t]);                        // a row that straddles chunks (its words were cleared: idx_clear_kernel)
    }
}

// the words of the rows that straddle the masks' chunks are ORed into memory: they start from zero.  One thread per (word, chunk edge).
__global__ __launch_bounds__(kBlock) void idx_clear_kernel(const ure_shard_t *__restrict__ shards, const shard_aux *__restrict__ aux, int64_t tick)
{
    const ure_shard_t &S = shards[blockIdx.y];
    const shard_aux &A = aux[blockIdx.y];
    if (idx_epoch_start(S, A, tick) < 0) return;
    const int n_all = S.n_user + S.n_item;
    const int64_t n_grp = S.n_slots / 8;
    const int64_t chunks = (n_grp + kIdxMaskChunk / 8 - 1) / (kIdxMaskChunk / 8);
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < chunks * 2; t += (int64_t)gridDim.x * kBlock) {
        const int64_t c = t >> 1;
        const int64_t g = (t & 1) ? min((c + 1) * (kIdxMaskChunk / 8), n_grp) - 1 : c * (kIdxMaskChunk / 8);      // the chunk's first / last group
        const int row_id = ldg(A.grp_row + 2 * g + 1);
        for (int w = 0; w < A.idx_words; ++w) stg(A.W + (size_t)w * n_all + row_id, 0ull);
    }
}

// ---- epoch start 2: bit 63 of every word, the first step, the buffer at the epoch's end.  One thread per ROW ID (rows without interactions in
// the shard have words of zeros -- the masks' memory is cleared when the job is made and only active rows are ever written -- and get "no step"):
// in schedule order every load and store below was a gather (126 us per epoch at the 25 M shape).
__global__ __launch_bounds__(kBlock) void idx_parity_kernel(const ure_shard_t *__restrict__ shards, const shard_aux *__restrict__ aux, int64_t tick)
{
    const ure_shard_t &S = shards[blockIdx.y];
    const shard_aux &A = aux[blockIdx.y];
    const int epoch = idx_epoch_start(S, A, tick);
    if (epoch < 0) return;
    const int n_all = S.n_user + S.n_item;
    for (int row_id = blockIdx.x * kBlock + threadIdx.x; row_id < n_all; row_id += gridDim.x * kBlock) {
        int par = 0, first = A.steps;
        unsigned long long v[kIdxMaxWords];
#pragma unroll
        for (int w = 0; w < kIdxMaxWords; ++w) v[w] = w < A.idx_words ? ldg(A.W + (size_t)w * n_all + row_id) & kIdxBits : 0ull;
#pragma unroll
        for (int w = 0; w < kIdxMaxWords; ++w) {
            if (w >= A.idx_words) break;
            if (v[w] && first == A.steps) first = w * kIdxWin + __ffsll((long long)v[w]) - 1;
            stg(A.W + (size_t)w * n_all + row_id, v[w] | ((unsigned long long)par << 63));
            par ^= __popcll(v[w]) & 1;
        }
        stg(A.first_step + row_id, (uint16_t)first);
        stg(A.end_par[epoch & 1] + row_id, (uint8_t)par);
        // next_first[w][row] = the row's first own step in the words after w (steps: none): what a slot of a row's LAST step of
        // word w needs to know how long the row then waits (idx_own_bits)
        int nxt = A.steps;
#pragma unroll
        for (int w = kIdxMaxWords - 1; w >= 0; --w) {
            if (w >= A.idx_words) continue;
            stg(A.next_first + (size_t)w * n_all + row_id, (uint16_t)nxt);
            if (v[w]) nxt = w * kIdxWin + __ffsll((long long)v[w]) - 1;
        }
    }
}

// ---- epoch start 3: slots per (chunk, step).  One wavefront per chunk of 4,096 slots.
__global__ __launch_bounds__(kBlock) void idx_hist_kernel(const ure_shard_t *__restrict__ shards, const shard_aux *__restrict__ aux, int64_t tick)
{
    __shared__ unsigned cnt[kWavesPerBlock][kIdxMaxSteps];
    const ure_shard_t &S = shards[blockIdx.y];
    const shard_aux &A = aux[blockIdx.y];
    const int epoch = idx_epoch_start(S, A, tick);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = (int)blockIdx.x * kWavesPerBlock + wave;
    if (epoch < 0 || c >= A.idx_chunks) return;                     // (wave-uniform; no workgroup barrier below)
    const int steps = A.steps;
    unsigned *mine = cnt[wave];
    for (int s = lane; s < steps; s += kWave) mine[s] = 0u;
    __builtin_amdgcn_wave_barrier();
    const uint16_t *__restrict__ ent_tag = S.ent_tag + tag_buffer(S, epoch);
    const int64_t lo = (int64_t)c * A.idx_chunk, hi = min(lo + A.idx_chunk, S.n_slots);
    for (int64_t p = lo + lane * 8; p < hi; p += kWave * 8) {
        const uint4 t4 = ldg_u4(ent_tag + p);
        const unsigned tw[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned st = (tw[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu;
            if (st < (unsigned)steps) atomicAdd(&mine[st], 1u);
        }
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t *__restrict__ out = A.hist + (size_t)c * (steps + 1);
    for (int s = lane; s < steps; s += kWave) stg(out + s, mine[s]);
}

// ---- epoch start 4: exclusive scan of hist over (step, chunk), in three launches.  Chunks are cut into 32 segments.
__device__ __forceinline__ void idx_seg_range(const shard_aux &A, int seg, int *c0, int *c1)
{
    const int per = (A.idx_chunks + kIdxSeg - 1) / kIdxSeg;
    *c0 = min(seg * per, A.idx_chunks);
    *c1 = min(*c0 + per, A.idx_chunks);
}
__global__ __launch_bounds__(kBlock) void idx_scan1_kernel(const ure_shard_t *__restrict__ shards, const shard_aux *__restrict__ aux, int64_t tick)
{
    const ure_shard_t &S = shards[blockIdx.z];
    const shard_aux &A = aux[blockIdx.z];
    const int s = blockIdx.x * kBlock + threadIdx.x;
    if (idx_epoch_start(S, A, tick) < 0 || s >= A.steps) return;
    int c0, c1;
    idx_seg_range(A, (int)blockIdx.y, &c0, &c1);
    unsigned sum = 0;
    for (int c = c0; c < c1; c += kIdxScanBatch) {               // (a batch of loads in flight per thread: the walk is a chain of memory latencies otherwise)
        unsigned t[kIdxScanBatch];
#pragma unroll
        for (int k = 0; k < kIdxScanBatch; ++k) t[k] = c + k < c1 ? ldg(A.hist + (size_t)(c + k) * (A.steps + 1) + s) : 0u;
#pragma unroll
        for (int k = 0; k < kIdxScanBatch; ++k) sum += t[k];
    }
    stg(A.seg + (size_t)blockIdx.y * (A.steps + 1) + s, sum);
}
// one workgroup per shard: step_begin = exclusive prefix of the steps' totals; seg[y][s] <- where segment y's slots of step s start
__global__ __launch_bounds__(1024) void idx_scan2_kernel(const ure_shard_t *__restrict__ shards, const shard_aux *__restrict__ aux, int64_t tick)
{
    __shared__ unsigned tot[1024];
    const ure_shard_t &S = shards[blockIdx.x];
    const shard_aux &A = aux[blockIdx.x];
    if (idx_epoch_start(S, A, tick) < 0) return;
    const int s = threadIdx.x, steps = A.steps;
    unsigned mine = 0;
    unsigned t[kIdxSeg];
#pragma unroll
    for (int y = 0; y < kIdxSeg; ++y) {
        t[y] = s < steps ? ldg(A.seg + (size_t)y * (steps + 1) + s) : 0u;
        mine += t[y];
    }
    tot[s] = mine;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const unsigned v = s >= o ? tot[s - o] : 0u;
        __syncthreads();
        tot[s] += v;
        __syncthreads();
    }
    const unsigned begin = tot[s] - mine;
    if (s < steps) {
        stg(A.step_begin + s, begin);
        unsigned run = begin;
#pragma unroll
        for (int y = 0; y < kIdxSeg; ++y) {
            stg(A.seg + (size_t)y * (steps + 1) + s, run);
            run += t[y];
        }
    }
    if (s == steps) { stg(A.step_begin + steps, begin); stg(A.step_begin + steps + 1, begin); }      // sorted slots in all (s == steps <= 1008 < 1024)
}
__global__ __launch_bounds__(kBlock) void idx_scan3_kernel(const ure_shard_t *__restrict__ shards, const shard_aux *__restrict__ aux, int64_t tick)
{
    const ure_shard_t &S = shards[blockIdx.z];
    const shard_aux &A = aux[blockIdx.z];
    const int s = blockIdx.x * kBlock + threadIdx.x;
    if (idx_epoch_start(S, A, tick) < 0 || s >= A.steps) return;
    int c0, c1;
    idx_seg_range(A, (int)blockIdx.y, &c0, &c1);
    unsigned run = ldg(A.seg + (size_t)blockIdx.y * (A.steps + 1) + s);
    // (the loads of a batch before its stores: one load, one store at a time was 343 memory latencies in a row -- 85 us per epoch at the 25 M shape)
    for (int c = c0; c < c1; c += kIdxScanBatch) {
        unsigned t[kIdxScanBatch];
#pragma unroll
        for (int k = 0; k < kIdxScanBatch; ++k) t[k] = c + k < c1 ? ldg(A.hist + (size_t)(c + k) * (A.steps + 1) + s) : 0u;
#pragma unroll
        for (int k = 0; k < kIdxScanBatch; ++k) {
            if (c + k < c1) stg(A.hist + (size_t)(c + k) * (A.steps + 1) + s, run);
            run += t[k];
        }
    }
}

// ---- epoch start 5: the stable scatter.  One wavefront per chunk walks its slots in order, 64 at a time; slots of a batch with
// the same tag are ranked by lane (ten ballots find a lane's peers), the chunk's running offsets live in LDS, private to the wave.
__global__ __launch_bounds__(kBlock) void idx_scatter_kernel(const ure_shard_t *__restrict__ shards, const shard_aux *__restrict__ aux, int64_t tick)
{
    __shared__ unsigned cnt[kWavesPerBlock][kIdxMaxSteps];
    const ure_shard_t &S = shards[blockIdx.y];
    const shard_aux &A = aux[blockIdx.y];
    const int epoch = idx_epoch_start(S, A, tick);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = (int)blockIdx.x * kWavesPerBlock + wave;
    if (epoch < 0 || c >= A.idx_chunks) return;
    const int steps = A.steps;
    unsigned *mine = cnt[wave];
    const uint32_t *__restrict__ off = A.hist + (size_t)c * (steps + 1);
    for (int s = lane; s < steps; s += kWave) mine[s] = ldg(off + s);
    __builtin_amdgcn_wave_barrier();
    const uint16_t *__restrict__ ent_tag = S.ent_tag + tag_buffer(S, epoch);
    const int64_t lo = (int64_t)c * A.idx_chunk, hi = min(lo + A.idx_chunk, S.n_slots);
    const unsigned long long below = (1ull << lane) - 1ull;
    const int n_all = S.n_user + S.n_item;
    // Two stages ahead of the running offsets (nothing of either depends on them): what a batch reads of the row-major arrays is
    // requested TWO batches ahead, the mask word of its slots' opposite rows -- a gather that needs the tag, the opposite id and the row --
    // one batch ahead.  (Unpipelined, the two dependent levels sat in every one of a wave's 64 sequential iterations: 1.36 ms per epoch.)
    struct Batch { unsigned tag; int oid, idx, row; float r; };
    auto fetch = [&](int64_t p) {
        Batch b{0xFFFFu, 0, 0, 0, 0.f};
        if (p < hi) {
            b.tag = ldg(ent_tag + p);
            b.oid = ldg(S.ent_oid + p);
            b.r = ldg(S.ent_r + p);
            const ure_i2 g = *(const ure_i2 URE_AS1 *)(A.grp_row + 2 * (p >> 3));
            const int gi = g.x, gr = g.y;
            b.idx = gi; b.row = gr;
        }
        return b;
    };
    auto word_of = [&](const Batch &b) {
        // the mask word of the slot's OPPOSITE row (this slot's row is an item row: the opposite row is a user)
        if (b.tag >= (unsigned)steps) return 0ull;
        const int other = b.row >= S.n_user ? b.oid : S.n_user + b.oid;
        return ldg(A.W + (size_t)(b.tag / kIdxWin) * n_all + other);
    };
    auto own_of = [&](const Batch &b) { return b.tag < (unsigned)steps ? idx_own_bits(A, n_all, b.row, b.tag) : 0u; };
    Batch cur = fetch(lo + lane), nxt = fetch(lo + kWave + lane);
    unsigned long long cur_word = word_of(cur);
    unsigned cur_own = own_of(cur);
    for (int64_t p0 = lo; p0 < hi; p0 += kWave) {
        const Batch far = fetch(p0 + 2 * kWave + lane);
        const unsigned long long nxt_word = word_of(nxt);
        const unsigned nxt_own = own_of(nxt), own = cur_own;
        const unsigned tag = cur.tag;
        const int idx = cur.idx, row = cur.row;
        const float r = cur.r;
        const bool valid = tag < (unsigned)steps;
        const int oid = cur.oid | (valid ? idx_buffer_at(cur_word, (int)(tag % kIdxWin)) << 31 : 0);
        cur = nxt; cur_word = nxt_word; cur_own = nxt_own; nxt = far;
        const int cls = idx < S.n_split ? kIdxSplit : idx < S.n_multi ? kIdxHeavy : kIdxLight;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 10; ++b) {
            const bool bit = (tag >> b) & 1u;
            const unsigned long long m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        if (valid) {
            const unsigned base = mine[tag];                            // (all peers read before their first lane writes: LDS runs a wave's accesses in order)
            const int rank = __popcll(peers & below);
            if (rank == 0) mine[tag] = base + (unsigned)__popcll(peers);
            stg_u4(A.sslot + (size_t)(base + rank), make_uint4((unsigned)oid, __float_as_uint(r), (unsigned)row, tag | ((unsigned)cls << 16) | own));
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- epoch start 5, epochs of at most 63 steps (a job of short epochs of narrow rows: engine.INDEX_SHORT_EPOCH_MAX_D).  With few steps
// a chunk's slots of one step are MANY -- 1,024 slots / 27 steps = 38 -- and they are neighbours in the sorted array: the wavefront
// sorts 1,024 slots at a time in LDS (count per step, prefix, place) and writes them out in sorted order, so a step's share leaves as
// one run of ~600 bytes instead of 38 scattered 16-byte stores (partial-line stores run at ~34 G/s: 1.3 ms per epoch for the 45 M
// sorted slots of configs[3]'s shape, whatever the row width).  Every slot lands exactly where idx_scatter_kernel puts it.
constexpr int kIdxStage = 1024;                 // slots a wavefront sorts in LDS at a time
#ifndef URE_INDEX_STAGED_WAVES
#define URE_INDEX_STAGED_WAVES 8
#endif
constexpr int kIdxStagedWaves = URE_INDEX_STAGED_WAVES;   // wavefronts that sort a chunk of an epoch of 64+ steps together (idx_scatter_staged_kernel)

// The stage holds a word per slot -- where the slot is among the 1,024, its step, the buffer bit of its opposite row, the bits of its own row's
// step -- and the record is put together on the way out (loads of lines the wavefront has just read): 4 KB of LDS per wavefront, so that the
// wavefronts a CU holds are bounded by registers (four per SIMD), not by LDS (two with full records in the stage; 5.70 against 5.64 ms per epoch
// of configs[3]'s shape at k = 16, profiles/r05/NOTES.md 4b).
__global__ __launch_bounds__(kBlock) void idx_scatter_short_kernel(const ure_shard_t *__restrict__ shards, const shard_aux *__restrict__ aux, int64_t tick)
{
    __shared__ unsigned stage_c[kWavesPerBlock][kIdxStage];
    __shared__ unsigned goff[kWavesPerBlock][64], lcnt[kWavesPerBlock][64], lstart[kWavesPerBlock][64], lfill[kWavesPerBlock][64];
    const ure_shard_t &S = shards[blockIdx.y];
    const shard_aux &A = aux[blockIdx.y];
    const int epoch = idx_epoch_start(S, A, tick);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = (int)blockIdx.x * kWavesPerBlock + wave;
    if (epoch < 0 || c >= A.idx_chunks) return;
    const int steps = A.steps;                                      // <= 63: a step is its own lane below
    unsigned *stc = stage_c[wave];
    unsigned *g_off = goff[wave], *l_cnt = lcnt[wave], *l_start = lstart[wave], *l_fill = lfill[wave];
    g_off[lane] = lane < steps ? ldg(A.hist + (size_t)c * (steps + 1) + lane) : 0u;
    l_cnt[lane] = 0u;
    __builtin_amdgcn_wave_barrier();
    const uint16_t *__restrict__ ent_tag = S.ent_tag + tag_buffer(S, epoch);
    const int64_t lo = (int64_t)c * A.idx_chunk, hi = min(lo + A.idx_chunk, S.n_slots);
    const unsigned long long below = (1ull << lane) - 1ull;
    const int n_all = S.n_user + S.n_item;
    struct Batch { unsigned tag; int oid, row; };
    auto fetch = [&](int64_t p) {
        Batch b{0xFFFFu, 0, 0};
        if (p < hi) {
            b.tag = ldg(ent_tag + p);
            b.oid = ldg(S.ent_oid + p);
            b.row = ldg(A.grp_row + 2 * (p >> 3) + 1);
        }
        return b;
    };
    auto word_of = [&](const Batch &b) {
        if (b.tag >= (unsigned)steps) return 0ull;
        const int other = b.row >= S.n_user ? b.oid : S.n_user + b.oid;
        return ldg(A.W + (size_t)(b.tag / kIdxWin) * n_all + other);
    };
    // Eight batches at a time: their loads of the row-major arrays go out together, then the eight gathers of the opposite rows' mask
    // words, then the batches are placed one after the other -- the two memory levels are paid once per 512 slots, not per 64.
    constexpr int kRound = 8;
    for (int64_t s0 = lo; s0 < hi; s0 += kIdxStage) {
        const int64_t s1 = min(s0 + kIdxStage, hi);
        // the counts of these 1,024 slots per step (their tags once more: 2 KB, in the caches), then the exclusive prefix over the steps
        for (int64_t p = s0 + lane * 8; p < s1; p += kWave * 8) {
            const uint4 t4 = ldg_u4(ent_tag + p);
            const unsigned tw[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned t = (tw[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu;
                if (t < (unsigned)steps) atomicAdd(&l_cnt[t], 1u);
            }
        }
        __builtin_amdgcn_wave_barrier();
        const unsigned mine_n = l_cnt[lane];
        unsigned incl = mine_n;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            const unsigned v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        l_start[lane] = incl - mine_n;
        l_fill[lane] = incl - mine_n;
        __builtin_amdgcn_wave_barrier();
        for (int64_t r0 = s0; r0 < s1; r0 += kRound * kWave) {
            Batch bt[kRound];
            unsigned long long wd[kRound];
            unsigned own[kRound];
#pragma unroll
            for (int k = 0; k < kRound; ++k) bt[k] = fetch(r0 + k * kWave + lane);
#pragma unroll
            for (int k = 0; k < kRound; ++k) {
                wd[k] = word_of(bt[k]);
                own[k] = bt[k].tag < (unsigned)steps ? idx_own_bits(A, n_all, bt[k].row, bt[k].tag) : 0u;
            }
#pragma unroll
            for (int k = 0; k < kRound; ++k) {
                const unsigned tag = bt[k].tag;
                const bool valid = tag < (unsigned)steps;
                const unsigned opp = valid ? (unsigned)idx_buffer_at(wd[k], (int)(tag % kIdxWin)) : 0u;
                unsigned long long peers = __ballot(valid);
#pragma unroll
                for (int b = 0; b < 6; ++b) {                               // (tags below 64)
                    const bool bit = (tag >> b) & 1u;
                    const unsigned long long m = __ballot(bit);
                    peers &= bit ? m : ~m;
                }
                if (valid) {
                    const unsigned base = l_fill[tag];                      // (all peers read before their first lane writes: LDS runs a wave's accesses in order)
                    const int rank = __popcll(peers & below);
                    if (rank == 0) l_fill[tag] = base + (unsigned)__popcll(peers);
                    // slot among the 1,024 | step << 10 | opposite row's buffer << 16 | own row's buffer << 17 | gap << 18 (steps, gaps below 64)
                    stc[base + rank] = (unsigned)(r0 + k * kWave + lane - s0) | tag << 10 | opp << 16 | (own[k] >> kIdxOwnBufBit) << 17;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        // the staged slots leave in sorted order: slot i of the stage belongs to the step whose range [l_start, l_start + l_cnt) holds i
        const unsigned total = l_start[63] + l_cnt[63];
        for (unsigned i = lane; i < total; i += kWave) {
            const unsigned v = stc[i];
            const int64_t p = s0 + (int64_t)(v & 1023u);
            const ure_i2 g = *(const ure_i2 URE_AS1 *)(A.grp_row + 2 * (p >> 3));
            const int gi = g.x, gr = g.y;
            const int cls = gi < S.n_split ? kIdxSplit : gi < S.n_multi ? kIdxHeavy : kIdxLight;
            const uint4 rec = make_uint4((unsigned)ldg(S.ent_oid + p) | ((v >> 16) & 1u) << 31, __float_as_uint(ldg(S.ent_r + p)), (unsigned)gr,
                                         ((v >> 10) & 63u) | ((unsigned)cls << 16) | (v >> 17) << kIdxOwnBufBit);
            const unsigned tag = rec.w & 0xFFFFu;
            stg_u4(A.sslot + (size_t)(g_off[tag] + (i - l_start[tag])), rec);
        }
        __builtin_amdgcn_wave_barrier();
        g_off[lane] += l_cnt[lane];
        l_cnt[lane] = 0u;
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- epoch start 5, epochs of 64 .. 1,008 steps: the same idea with a WORKGROUP per chunk.  A chunk's slots of one step are few -- 4,096
// slots / 750 steps = 5.5 -- but they are neighbours in the sorted array, and idx_scatter_kernel stores them one 16-byte record at a time,
// minutes of the wave's life apart: every record leaves its L2 as a partial-line write of its own (45 M of them per epoch at the 25 M
// shape: ~34 G/s, 1.31 ms whatever else the kernel does).  Here the four wavefronts of a workgroup sort ONE chunk in LDS -- count per (wave,
// step), prefix over (step, wave), every wave places its quarter in slot order (the ranks of idx_scatter_kernel: ballots inside a batch of 64,
// running counters) -- and the chunk leaves in sorted order: a step's 5.5 records are one stretch of 88 bytes, stored by neighbouring lanes.
// Every slot lands exactly where idx_scatter_kernel puts it.
template <int WAVES>
__global__ __launch_bounds__(WAVES * kWave) void idx_scatter_staged_kernel(const ure_shard_t *__restrict__ shards, const shard_aux *__restrict__ aux, int64_t tick)
{
    // the stage holds a word per slot -- where the slot is in the chunk, its step, the buffer bit of its opposite row -- and the record is put
    // together when the chunk leaves (three loads of lines this workgroup has just read): 16 KB instead of 64 for full records, two workgroups per CU
    __shared__ unsigned stage_c[kIdxChunk];
    __shared__ uint16_t stage_own[kIdxChunk];                   // ... and the bits of its own row's step (idx_own_bits >> 18), read beside the opposite row's word
    __shared__ unsigned fill[WAVES][kIdxMaxSteps];              // per (wave, step): count, then where the wave's next record of the step goes
    __shared__ unsigned lstart[kIdxMaxSteps + 1], goff[kIdxMaxSteps];    // a step's first record in the stage / in the sorted array
    __shared__ unsigned wave_tot[WAVES];
    constexpr int kThreads = WAVES * kWave, kPer = (kIdxMaxSteps + kThreads - 1) / kThreads;      // steps a thread takes of the prefix
    const ure_shard_t &S = shards[blockIdx.y];
    const shard_aux &A = aux[blockIdx.y];
    const int epoch = idx_epoch_start(S, A, tick);
    const int c = (int)blockIdx.x;
    if (epoch < 0 || c >= A.idx_chunks) return;                         // (workgroup-uniform)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int steps = A.steps;
    for (int s = tid; s < steps; s += kThreads) {
#pragma unroll
        for (int w = 0; w < WAVES; ++w) fill[w][s] = 0u;
        goff[s] = ldg(A.hist + (size_t)c * (steps + 1) + s);
    }
    const uint16_t *__restrict__ ent_tag = S.ent_tag + tag_buffer(S, epoch);
    const int64_t lo = (int64_t)c * kIdxChunk, hi = min(lo + kIdxChunk, S.n_slots);
    const int64_t w_lo = lo + (int64_t)wave * (kIdxChunk / WAVES), w_hi = min(w_lo + kIdxChunk / WAVES, hi);
    const int n_all = S.n_user + S.n_item;
    __syncthreads();
    // ---- the wave's slots per step
    unsigned *mine = fill[wave];
    for (int64_t p = w_lo + lane * 8; p < w_hi; p += kWave * 8) {
        const uint4 t4 = ldg_u4(ent_tag + p);
        const unsigned tw[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned t = (tw[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu;
            if (t < (unsigned)steps) atomicAdd(&mine[t], 1u);
        }
    }
    __syncthreads();
    // ---- exclusive prefix over (step, wave): thread t takes the steps [kPer t, kPer t + kPer)
    {
        unsigned own = 0u;
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int s = kPer * tid + k;
            if (s < steps)
                for (int w = 0; w < WAVES; ++w) own += fill[w][s];
        }
        unsigned incl = own;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            const unsigned v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        if (lane == kWave - 1) wave_tot[wave] = incl;
        __syncthreads();
        unsigned run = incl - own;
        for (int w = 0; w < wave; ++w) run += wave_tot[w];
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int s = kPer * tid + k;
            if (s < steps) {
                lstart[s] = run;
                for (int w = 0; w < WAVES; ++w) {
                    const unsigned n_ws = fill[w][s];
                    fill[w][s] = run;
                    run += n_ws;
                }
            }
        }
        if (tid == kThreads - 1) lstart[kIdxMaxSteps] = run;           // the chunk's valid slots
    }
    __syncthreads();
    // ---- every wave places its quarter, in slot order
    const unsigned long long below = (1ull << lane) - 1ull;
    struct Batch { unsigned tag; int oid, row; };
    auto fetch = [&](int64_t p) {
        Batch b{0xFFFFu, 0, 0};
        if (p < w_hi) {
            b.tag = ldg(ent_tag + p);
            b.oid = ldg(S.ent_oid + p);
            b.row = ldg(A.grp_row + 2 * (p >> 3) + 1);
        }
        return b;
    };
    auto word_of = [&](const Batch &b) {
        if (b.tag >= (unsigned)steps) return 0ull;
        const int other = b.row >= S.n_user ? b.oid : S.n_user + b.oid;
        return ldg(A.W + (size_t)(b.tag / kIdxWin) * n_all + other);
    };
    constexpr int kRound = 8;
    for (int64_t r0 = w_lo; r0 < w_hi; r0 += kRound * kWave) {
        Batch bt[kRound];
        unsigned long long wd[kRound];
        unsigned own[kRound];
#pragma unroll
        for (int k = 0; k < kRound; ++k) bt[k] = fetch(r0 + k * kWave + lane);
#pragma unroll
        for (int k = 0; k < kRound; ++k) {
            wd[k] = word_of(bt[k]);
            // (the slot's own row's word and next_first entry: cached lines -- a chunk holds ~20 rows -- beside the gather above.  The kernel takes
            // 1.10 ms per epoch at the 25 M shape with them, 0.96 without, from LDS copies of the rows' words as well: NOTES 4)
            own[k] = bt[k].tag < (unsigned)steps ? idx_own_bits(A, n_all, bt[k].row, bt[k].tag) : 0u;
        }
#pragma unroll
        for (int k = 0; k < kRound; ++k) {
            const unsigned tag = bt[k].tag;
            const bool valid = tag < (unsigned)steps;
            const unsigned opp = valid ? (unsigned)idx_buffer_at(wd[k], (int)(tag % kIdxWin)) : 0u;
            unsigned long long peers = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 10; ++b) {
                const bool bit = (tag >> b) & 1u;
                const unsigned long long m = __ballot(bit);
                peers &= bit ? m : ~m;
            }
            if (valid) {
                const unsigned base = mine[tag];                        // (all peers read before their first lane writes: LDS runs a wave's accesses in order)
                const int rank = __popcll(peers & below);
                if (rank == 0) mine[tag] = base + (unsigned)__popcll(peers);
                stage_c[base + rank] = (unsigned)(r0 + k * kWave + lane - lo) | opp << 12 | tag << 13;
                stage_own[base + rank] = (uint16_t)(own[k] >> kIdxOwnBufBit);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    // ---- the chunk leaves in sorted order
    const unsigned total = lstart[kIdxMaxSteps];
    for (unsigned i = tid; i < total; i += kThreads) {
        const unsigned v = stage_c[i];
        const int64_t p = lo + (int64_t)(v & 4095u);
        const unsigned tag = v >> 13;
        const ure_i2 g = *(const ure_i2 URE_AS1 *)(A.grp_row + 2 * (p >> 3));
        const int gi = g.x, gr = g.y;
        const int cls = gi < S.n_split ? kIdxSplit : gi < S.n_multi ? kIdxHeavy : kIdxLight;
        const uint4 rec = make_uint4((unsigned)ldg(S.ent_oid + p) | ((v >> 12) & 1u) << 31, __float_as_uint(ldg(S.ent_r + p)), (unsigned)gr,
                                     tag | ((unsigned)cls << 16) | (unsigned)stage_own[i] << kIdxOwnBufBit);
        stg_u4(A.sslot + (size_t)(goff[tag] + (i - lstart[tag])), rec);
    }
}

// ---- epoch start 6: the bitmap of run starts (a sorted slot whose row or step differs from its predecessor's) and their number per
// block of 2,048 sorted slots.  (The buffer of a slot's opposite row, a gather from W, is marked by the scatter itself: round 4
// had a pass of its own here that rewrote all 720 MB of sorted slots.)
__global__ __launch_bounds__(kBlock) void idx_mark_kernel(const ure_shard_t *__restrict__ shards, const shard_aux *__restrict__ aux, int64_t tick)
{
    __shared__ unsigned wave_cnt[kWavesPerBlock];
    const ure_shard_t &S = shards[blockIdx.y];
    const shard_aux &A = aux[blockIdx.y];
    if (idx_epoch_start(S, A, tick) < 0) return;
    const int64_t total = ldg(A.step_begin + A.steps);
    const int64_t q_lo = (int64_t)blockIdx.x * kIdxFlagBlock;
    if (q_lo >= total) return;                                      // (workgroup-uniform)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned starts = 0;
    for (int it = 0; it < kIdxFlagBlock / kBlock; ++it) {
        const int64_t q = q_lo + it * kBlock + threadIdx.x;
        bool start = false;
        if (q < total) {
            const ure_u2 key = *(const ure_u2 URE_AS1 *)(reinterpret_cast<const unsigned *>(A.sslot + q) + 2);            // {row id, step | class << 16}
            if (q == 0) start = true;
            else {
                const ure_u2 prev = *(const ure_u2 URE_AS1 *)(reinterpret_cast<const unsigned *>(A.sslot + q - 1) + 2);
                const unsigned kx = key.x, ky = key.y, px = prev.x, py = prev.y;
                start = kx != px || ky != py;
            }
        }
        const unsigned long long vote = __ballot(start);
        if (lane == 0 && q < total) stg(A.runflag + (q >> 6), vote);
        starts += (unsigned)__popcll(vote);
    }
    if (lane == 0) wave_cnt[wave] = starts;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned sum = 0;
        for (int k = 0; k < kWavesPerBlock; ++k) sum += wave_cnt[k];
        stg(A.blk_cnt + blockIdx.x, sum);
    }
}

// one workgroup per shard: blk_cnt <- its exclusive prefix; blk_cnt[n_blk] = the number of items
__global__ __launch_bounds__(1024) void idx_blkscan_kernel(const ure_shard_t *__restrict__ shards, const shard_aux *__restrict__ aux, int64_t tick)
{
    __shared__ unsigned part[1024];
    const ure_shard_t &S = shards[blockIdx.x];
    const shard_aux &A = aux[blockIdx.x];
    if (idx_epoch_start(S, A, tick) < 0) return;
    const int64_t total = ldg(A.step_begin + A.steps);
    const int n_blk = (int)((total + kIdxFlagBlock - 1) / kIdxFlagBlock);
    const int per = (n_blk + 1023) / 1024;
    const int b0 = min((int)threadIdx.x * per, n_blk), b1 = min(b0 + per, n_blk);
    unsigned mine = 0;
    for (int b = b0; b < b1; ++b) mine += ldg(A.blk_cnt + b);
    part[threadIdx.x] = mine;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const unsigned v = (int)threadIdx.x >= o ? part[threadIdx.x - o] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    unsigned run = part[threadIdx.x] - mine;
    for (int b = b0; b < b1; ++b) {
        const unsigned t = ldg(A.blk_cnt + b);
        stg(A.blk_cnt + b, run);
        run += t;
    }
    if (threadIdx.x == 1023) stg(A.blk_cnt + n_blk, part[1023]);
}

// index of the item that sorted slot q starts (or would start), from the run-start bitmap and the block prefix
__device__ __forceinline__ unsigned idx_item_of(const shard_aux &A, int64_t q)
{
    const int64_t blk = q / kIdxFlagBlock;
    unsigned at = ldg(A.blk_cnt + blk);
    for (int64_t w = blk * (kIdxFlagBlock / 64); w < (q >> 6); ++w) at += (unsigned)__popcll(ldg(A.runflag + w));
    return at + (unsigned)__popcll(ldg(A.runflag + (q >> 6)) & mask_below((int)(q & 63)));
}

// ---- epoch start 7: one item per run.  Workgroup b takes the sorted slots [2048 b, 2048 (b + 1)).
__global__ __launch_bounds__(kBlock) void idx_emit_kernel(const ure_shard_t *__restrict__ shards, const shard_aux *__restrict__ aux, int64_t tick)
{
    __shared__ unsigned before[kIdxFlagBlock / 64];             // runs that start in the block before each of its 64-slot words
    const ure_shard_t &S = shards[blockIdx.y];
    const shard_aux &A = aux[blockIdx.y];
    if (idx_epoch_start(S, A, tick) < 0) return;
    const int64_t total = ldg(A.step_begin + A.steps);
    const int64_t q_lo = (int64_t)blockIdx.x * kIdxFlagBlock;
    if (q_lo >= total) return;
    const int lane = threadIdx.x & 63;
    const int64_t n_words = (total + 63) >> 6;
    if (threadIdx.x == 0) {
        unsigned run = ldg(A.blk_cnt + blockIdx.x);
        for (int w = 0; w < kIdxFlagBlock / 64; ++w) {
            before[w] = run;
            const int64_t wi = (q_lo >> 6) + w;
            if (wi < n_words) run += (unsigned)__popcll(ldg(A.runflag + wi));
        }
    }
    __syncthreads();
    for (int it = 0; it < kIdxFlagBlock / kBlock; ++it) {
        const int64_t q = q_lo + it * kBlock + threadIdx.x;
        if (q >= total) continue;
        const int64_t wi = q >> 6;
        const unsigned long long flags = ldg(A.runflag + wi);
        if (!((flags >> lane) & 1ull)) continue;
        const unsigned i = before[(int)(wi - (q_lo >> 6))] + (unsigned)__popcll(flags & mask_below(lane));
        const uint4 rec = ldg_u4(A.sslot + q);
        const int row_id = (int)rec.z, cls = (int)(rec.w >> 16) & 3;
        // where the run ends: the next run start (a few words on for the runs of the heaviest rows)
        unsigned long long later = lane < 63 ? flags >> (lane + 1) : 0ull;
        int64_t end = later ? q + 1 + (__ffsll((long long)later) - 1) : -1;
        for (int64_t w2 = wi + 1; end < 0; ++w2) {
            if (w2 >= n_words) { end = total; break; }
            const unsigned long long f2 = ldg(A.runflag + w2);
            if (f2) end = (w2 << 6) + (__ffsll((long long)f2) - 1);
        }
        // the buffer of the row at this step and the steps that pass until its next own step came with the slot (idx_own_bits)
        const int buf = (int)(rec.w >> kIdxOwnBufBit) & 1, gap = (int)(rec.w >> kIdxGapShift) & 0x3FF;
        stg_i4(A.items + i, make_int4(row_id | (buf << 31), (int)q, (int)min(end, total), gap | (cls << 16)));
        // the run's first two slots travel with the item (items2): most runs are one or two slots long (1.6 on average at the 25 M shape),
        // and their gathers then start with the row's own loads instead of one memory level later
        uint4 two = make_uint4(rec.x, rec.y, 0u, 0u);
        if (min(end, total) - q >= 2) {
            const ure_u2 nx = *(const ure_u2 URE_AS1 *)(A.sslot + q + 1);
            const unsigned nxx = nx.x, nxy = nx.y;
            two.z = nxx; two.w = nxy;
        }
        stg_u4(A.items2 + i, two);
    }
}

// ---- epoch start 8: per step its first item, and the heavy prefix of its items with their workgroup counts.  One workgroup per step.
__global__ __launch_bounds__(kIdxHeavyMax) void idx_heavy_kernel(const ure_shard_t *__restrict__ shards, const shard_aux *__restrict__ aux, int64_t tick)
{
    __shared__ unsigned scan[kIdxHeavyMax];
    __shared__ unsigned item0, item1;
    const ure_shard_t &S = shards[blockIdx.y];
    const shard_aux &A = aux[blockIdx.y];
    if (idx_epoch_start(S, A, tick) < 0) return;
    const int s = blockIdx.x;
    if (s >= A.steps) return;
    const int64_t total = ldg(A.step_begin + A.steps);
    if (threadIdx.x < 2) {
        const int64_t q = ldg(A.step_begin + s + (int)threadIdx.x);
        const unsigned n_items = ldg(A.blk_cnt + (total + kIdxFlagBlock - 1) / kIdxFlagBlock);
        const unsigned i = q >= total ? n_items : idx_item_of(A, q);
        (threadIdx.x ? item1 : item0) = i;
        if (threadIdx.x == 0) stg(A.step_item + s, i);
        else if (s == A.steps - 1) { stg(A.step_item + A.steps, i); stg(A.step_item + A.steps + 1, i); }
    }
    __syncthreads();
    const int k = threadIdx.x;
    unsigned parts = 0;
    if (item0 + k < item1) {
        const int4 e = ldg_i4(reinterpret_cast<const int32_t *>(A.items + item0 + k));
        const int cls = (e.w >> 16) & 3;
        if (cls == kIdxSplit) parts = (unsigned)((e.z - e.y + kIdxPart - 1) / kIdxPart);
        else if (cls == kIdxHeavy) parts = 1u;
    }
    // the heavy items are a prefix of the step's items (rows keep the schedule's order): the first light one ends it
    scan[k] = parts;
    __syncthreads();
    __shared__ unsigned n_heavy;
    if (k == 0) {
        unsigned h = 0;
        while (h < (unsigned)kIdxHeavyMax && scan[h] != 0u) ++h;
        n_heavy = h;
        stg(A.heavy_cnt + s, h);
    }
    __syncthreads();
    if ((unsigned)k >= n_heavy) parts = 0;
    scan[k] = parts;
    __syncthreads();
    for (int o = 1; o < kIdxHeavyMax; o <<= 1) {
        const unsigned v = k >= o ? scan[k - o] : 0u;
        __syncthreads();
        scan[k] += v;
        __syncthreads();
    }
    uint32_t *cum = A.heavy_cum + (size_t)s * (kIdxHeavyMax + 1);
    if (k == 0) stg(cum, 0u);
    stg(cum + k + 1, scan[k]);
    // ... and for every workgroup of the step's heavy range the item and the part it takes (a search through `cum` in the step
    // kernel was a chain of up to 256 dependent loads: the step's critical path)
    uint32_t *map = A.heavy_map + (size_t)s * A.idx_hw;
    const unsigned first = scan[k] - parts;
    for (unsigned p = 0; p < parts && first + p < (unsigned)A.idx_hw; ++p) stg(map + first + p, (unsigned)k | (p << 9) | (parts << 20));
    if (k == kIdxHeavyMax - 1) {
        stg(A.heavy_wg + s, min(scan[k], (unsigned)A.idx_hw));
        stg_u4(A.step_desc + s, make_uint4(item0, item1, n_heavy, min(scan[k], (unsigned)A.idx_hw)));
    }
}

// ---- epoch start 9: every active row from "valid at the end of the last epoch" (buffer end_par) to "valid at its first step of this
// epoch", into buffer 0.  One lane per float4 of a row.
__global__ __launch_bounds__(kBlock) void idx_advance_kernel(const ure_shard_t *__restrict__ shards, const shard_aux *__restrict__ aux, int64_t tick)
{
    const ure_shard_t &S = shards[blockIdx.y];
    const shard_aux &A = aux[blockIdx.y];
    const int epoch = idx_epoch_start(S, A, tick);
    if (epoch < 0) return;
    const int d4 = S.d / 4;
    const int64_t total = (int64_t)S.n_active * d4;
    const float4 *__restrict__ tab = A.ptab + (size_t)epoch * A.ptab_stride;
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
        const int idx = (int)(t / d4), c4 = (int)(t % d4);
        const int row_id = ldg(S.sched + 4 * (size_t)idx);
        const int from = epoch > 0 ? (int)ldg(A.end_par[(epoch - 1) & 1] + row_id) : 0;
        const int j = (int)ldg(A.first_step + row_id);
        if (from == 0 && j == 0) continue;
        const bool is_user = row_id < S.n_user;
        const size_t o = (size_t)(is_user ? row_id : row_id - S.n_user) * S.d + (size_t)c4 * 4;
        float *wsrc = (is_user ? S.U[from] : S.V[from]) + o, *wdst = (is_user ? S.U[0] : S.V[0]) + o;
        float *mom = (is_user ? S.mU : S.mV) + o;
        RowVec<1> w, m;
        w.q[0] = ldg_f4(wsrc);
        m.q[0] = ldg_f4(mom);
        if (j != 0) row_advance<1>(w, m, tab[j]);
        stg_f4(wdst, w.q[0]);
        if (j != 0) stg_f4(mom, m.q[0]);
    }
}

// ---- one optimizer step.  Workgroups of a shard: [0, idx_hw) the heavy items (one each, split rows one per 256 slots) |
// the light items, a lane group each.
template <int LPR, int V4>
__device__ __forceinline__ void idx_sgd(const RowVec<V4> &w, const RowVec<V4> &m4, const RowVec<V4> &acc, float lam, float mu, float lr, RowVec<V4> &nw, RowVec<V4> &nm)
{
#pragma unroll
    for (int i = 0; i < V4; ++i) {                   // torch.optim.SGD: g += lam w ; buf = mu buf + g ; w -= lr buf
        const float4 ww = w.q[i], mm = m4.q[i], aa = acc.q[i];
        float4 g, wn;
        g.x = fmaf(lam, ww.x, aa.x); g.y = fmaf(lam, ww.y, aa.y); g.z = fmaf(lam, ww.z, aa.z); g.w = fmaf(lam, ww.w, aa.w);
        g.x = __fadd_rn(__fmul_rn(mu, mm.x), g.x); g.y = __fadd_rn(__fmul_rn(mu, mm.y), g.y);
        g.z = __fadd_rn(__fmul_rn(mu, mm.z), g.z); g.w = __fadd_rn(__fmul_rn(mu, mm.w), g.w);
        wn.x = fmaf(-lr, g.x, ww.x); wn.y = fmaf(-lr, g.y, ww.y); wn.z = fmaf(-lr, g.z, ww.z); wn.w = fmaf(-lr, g.w, ww.w);
        nm.q[i] = g;
        nw.q[i] = wn;
    }
}

// the update of one row, by the LPR lanes that hold it: optimizer, closed-form advance over `gap` steps, stores, train loss
template <int LPR, int V4>
__device__ __forceinline__ void idx_finish_row(const ure_shard_t &S, const shard_aux &A, int epoch, int row_id, int buf, int gap, const RowVec<V4> &w,
                                               const RowVec<V4> &m4, const RowVec<V4> &acc, float sse, float lr, int sub)
{
    constexpr int D = LPR * V4 * 4;
    const bool is_user = row_id < S.n_user;
    const int row = is_user ? row_id : row_id - S.n_user;
    const size_t row_off = (size_t)row * D;
    RowVec<V4> nw, nm;
    idx_sgd<LPR, V4>(w, m4, acc, S.lam, S.mu, lr, nw, nm);
    if (gap > 0) row_advance<V4>(nw, nm, A.ptab[(size_t)epoch * A.ptab_stride + gap]);
    // (written through the L2 with sc1 -- so that the 38 MB of rows a step rewrites are not left dirty for the kernel's end -- the
    // launch took 27.4 us against 26.1 with plain stores: profiles/r04/NOTES.md)
    row_store<LPR, V4>((is_user ? S.mU : S.mV) + row_off, sub, nm);
    row_store<LPR, V4>((is_user ? S.U[buf ^ 1] : S.V[buf ^ 1]) + row_off, sub, nw);
    if (is_user && sub == 0 && sse != 0.f) {
        float *slot = S.sse + (size_t)epoch * S.n_user + row;
        stg(slot, ldg(slot) + sse);
    }
}

// the gradient of the slots [first, end) taken with stride `stride` by this lane group: acc += 2 e v, sse += e^2
template <int LPR, int V4>
__device__ __forceinline__ void idx_gather(const ure_shard_t &S, const shard_aux &A, bool is_user, const RowVec<V4> &w, int first, int end, int stride, bool have,
                                           int sub, RowVec<V4> &acc, float &sse)
{
    constexpr int D = LPR * V4 * 4;
    constexpr int kGB = URE_INDEX_KGB;
    const float *__restrict__ other0 = is_user ? S.V[0] : S.U[0];
    const float *__restrict__ other1 = is_user ? S.V[1] : S.U[1];
    const uint4 *__restrict__ sslot = A.sslot;
    const int last = end - 1;
    ure_u2 rec[kGB];
#pragma unroll
    for (int k = 0; k < kGB; ++k) {
        const int q = min(first + k * stride, last);
        rec[k] = ure_u2{0u, 0u};
        if (have && first < end) rec[k] = *(const ure_u2 URE_AS1 *)(sslot + q);
    }
    for (int t = first; __any(have && t < end); t += kGB * stride) {
        bool act[kGB];
        float r[kGB];
        RowVec<V4> v[kGB];
#pragma unroll
        for (int k = 0; k < kGB; ++k) {
            act[k] = have && t + k * stride < end;
            const unsigned rbits = rec[k].y;             // (a copy first: __builtin_bit_cast of the vector ELEMENT read element 0, hipcc 7.2)
            r[k] = __uint_as_float(rbits);
            const unsigned o = act[k] ? rec[k].x : 0u;
            v[k] = row_load<LPR, V4>(((o >> 31) ? other1 : other0) + (size_t)(o & 0x7FFFFFFFu) * D, sub);
        }
        // the next batch's slots are requested before this batch's arithmetic
        const int tn = t + kGB * stride;
#pragma unroll
        for (int k = 0; k < kGB; ++k) {
            const int q = min(tn + k * stride, last);
            if (have && tn < end) rec[k] = *(const ure_u2 URE_AS1 *)(sslot + q);
        }
#pragma unroll
        for (int k = 0; k < kGB; ++k) {
            const float p = group_sum<LPR>(row_dot<V4>(w, v[k]));
            const float e = p - r[k];
            const float ge = act[k] ? 2.0f * e : 0.0f;
            if (act[k]) sse = fmaf(e, e, sse);
            row_axpy<V4>(acc, ge, v[k]);
        }
    }
}

template <int LPR, int V4>
__device__ __forceinline__ void mf_index_step(const ure_shard_t *__restrict__ shards, const shard_aux *__restrict__ aux, int64_t tick, unsigned n_sh, unsigned per)
{
    constexpr int G = kWave / LPR;
    constexpr int UPB = kBlock / LPR;
    constexpr int D = LPR * V4 * 4;
    __shared__ float4 part_acc[UPB][V4][LPR];
    __shared__ float part_sse[UPB];
    const WgMap wm = xcd_shard_map(blockIdx.x, n_sh, per);          // (a shard's gathers of its popular rows from one L2)
    const unsigned wg = (unsigned)wm.wg;
    const ure_shard_t &S = shards[wm.shard];
    const shard_aux &A = aux[wm.shard];
    const int steps = A.steps;
    if (tick >= (int64_t)steps * S.epochs) return;
    const int epoch = (int)epoch_of(A, tick);
    const int s = (int)(tick - (int64_t)epoch * steps);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int sub = lane & (LPR - 1), grp = lane / LPR;
    const int local = wave * G + grp;
    const float lr = ldg(S.lr + epoch);
    const uint4 sd = ldg_u4(A.step_desc + s);              // {first item, end, heavy items, workgroups of the heavy items}
    const unsigned i0 = sd.x, i1 = sd.y, n_heavy = sd.z;
    const int4 *__restrict__ items = A.items;
    if ((int)wg < A.idx_hw) {
        // ---- a heavy item (or one part of a split one): the workgroup's lane groups take its slots round robin
        const unsigned b = wg;
        if (b >= sd.w) return;
        const unsigned hm = ldg(A.heavy_map + (size_t)s * A.idx_hw + b);     // item of the heavy prefix | part << 9 | parts << 20
        const unsigned k = hm & 0x1FFu;
        const int part = (int)((hm >> 9) & 0x7FFu), n_parts = (int)(hm >> 20);
        const int4 e = ldg_i4(reinterpret_cast<const int32_t *>(items + i0 + k));
        const int row_id = e.x & 0x7FFFFFFF, buf = (int)((unsigned)e.x >> 31), gap = e.w & 0xFFFF;
        const int first = n_parts > 1 ? e.y + part * kIdxPart : e.y;
        const int end = n_parts > 1 ? min(e.z, first + kIdxPart) : e.z;
        const bool is_user = row_id < S.n_user;
        const size_t row_off = (size_t)(is_user ? row_id : row_id - S.n_user) * D;
        const RowVec<V4> w = row_load<LPR, V4>((is_user ? S.U[buf] : S.V[buf]) + row_off, sub);
        RowVec<V4> m4 = row_zero<V4>(), acc = m4;
        if (local == 0 && n_parts == 1) m4 = row_load<LPR, V4>((is_user ? S.mU : S.mV) + row_off, sub);
        float sse = 0.f;
        idx_gather<LPR, V4>(S, A, is_user, w, first + local, end, UPB, true, sub, acc, sse);
#pragma unroll
        for (int i = 0; i < V4; ++i) part_acc[local][i][sub] = acc.q[i];
        if (sub == 0) part_sse[local] = sse;
        __syncthreads();
        if (local != 0) return;
        // the lane groups' sums in lane-group order (a fixed order: the result does not depend on timing)
#pragma unroll
        for (int i = 0; i < V4; ++i) {
            float4 a4 = part_acc[0][i][sub];
            for (int g = 1; g < UPB; ++g) {
                const float4 t = part_acc[g][i][sub];
                a4.x += t.x; a4.y += t.y; a4.z += t.z; a4.w += t.w;
            }
            acc.q[i] = a4;
        }
        sse = 0.f;
        for (int g = 0; g < UPB; ++g) sse += part_sse[g];
        if (n_parts == 1) {
            idx_finish_row<LPR, V4>(S, A, epoch, row_id, buf, gap, w, m4, acc, sse, lr, sub);
            return;
        }
        // a part of a split row: the partial sums go to memory, idx_combine_kernel adds the parts in order and applies them
        float *out = A.partial + (size_t)b * (D + 4);
        row_store<LPR, V4>(out, sub, acc);
        if (sub == 0) stg(out + D, sse);
        return;
    }
    // ---- light items: a lane group each
    const unsigned i = i0 + n_heavy + (wg - (unsigned)A.idx_hw) * UPB + (unsigned)local;
    const bool have = i < i1;
    if (!__any(have)) return;
    int4 e = make_int4(0, 0, 0, 0);
    if (have) e = ldg_i4(reinterpret_cast<const int32_t *>(items + i));
    const int row_id = e.x & 0x7FFFFFFF, buf = (int)((unsigned)e.x >> 31), gap = e.w & 0xFFFF;
    const bool is_user = row_id < S.n_user;
    const size_t row_off = (size_t)(is_user ? row_id : row_id - S.n_user) * D;
    RowVec<V4> w = row_zero<V4>(), m4 = w, acc = w;
    if (have) {
        w = row_load<LPR, V4>((is_user ? S.U[buf] : S.V[buf]) + row_off, sub);
        m4 = row_load<LPR, V4>((is_user ? S.mU : S.mV) + row_off, sub);
    }
    float sse = 0.f;
#ifndef URE_INDEX_NO_INLINE
    {
        // the run's first two slots came with the item: their gathers are in flight with the row's own loads
        const uint4 two = have ? ldg_u4(A.items2 + i) : make_uint4(0u, 0u, 0u, 0u);
        const int cnt = e.z - e.y;
        const float *__restrict__ o0 = is_user ? S.V[0] : S.U[0];
        const float *__restrict__ o1 = is_user ? S.V[1] : S.U[1];
        const bool a0 = have && cnt >= 1, a1 = have && cnt >= 2;
        const unsigned x0 = a0 ? two.x : 0u, x1 = a1 ? two.z : 0u;
        const RowVec<V4> v0 = row_load<LPR, V4>(((x0 >> 31) ? o1 : o0) + (size_t)(x0 & 0x7FFFFFFFu) * D, sub);
        const RowVec<V4> v1 = row_load<LPR, V4>(((x1 >> 31) ? o1 : o0) + (size_t)(x1 & 0x7FFFFFFFu) * D, sub);
        {
            const float ee = group_sum<LPR>(row_dot<V4>(w, v0)) - __uint_as_float(two.y);
            if (a0) sse = fmaf(ee, ee, sse);
            row_axpy<V4>(acc, a0 ? 2.0f * ee : 0.0f, v0);
        }
        {
            const float ee = group_sum<LPR>(row_dot<V4>(w, v1)) - __uint_as_float(two.w);
            if (a1) sse = fmaf(ee, ee, sse);
            row_axpy<V4>(acc, a1 ? 2.0f * ee : 0.0f, v1);
        }
        e.y = min(e.y + 2, e.z);
    }
#endif
    idx_gather<LPR, V4>(S, A, is_user, w, e.y, e.z, 1, have, sub, acc, sse);
    if (have) idx_finish_row<LPR, V4>(S, A, epoch, row_id, buf, gap, w, m4, acc, sse, lr, sub);
}

template <int LPR, int V4>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(URE_INDEX_WAVES))) void mf_index_step_kernel(const ure_shard_t *__restrict__ shards, const shard_aux *__restrict__ aux, int64_t tick, unsigned n_sh, unsigned per)
{
    mf_index_step<LPR, V4>(shards, aux, tick, n_sh, per);
}

// the rows split over several workgroups: their parts' sums in part order, then the update.  One wavefront per split item.
template <int LPR, int V4>
__global__ __launch_bounds__(kWave) void idx_combine_kernel(const ure_shard_t *__restrict__ shards, const shard_aux *__restrict__ aux, int64_t tick)
{
    constexpr int D = LPR * V4 * 4;
    const ure_shard_t &S = shards[blockIdx.y];
    const shard_aux &A = aux[blockIdx.y];
    const int steps = A.steps;
    if (tick >= (int64_t)steps * S.epochs) return;
    const int epoch = (int)epoch_of(A, tick);
    const int s = (int)(tick - (int64_t)epoch * steps);
    const unsigned k = blockIdx.x;
    if (k >= ldg(A.heavy_cnt + s)) return;
    const uint32_t *__restrict__ cum = A.heavy_cum + (size_t)s * (kIdxHeavyMax + 1);
    const unsigned c0 = ldg(cum + k), c1 = ldg(cum + k + 1);
    if (c1 - c0 < 2u || (int)threadIdx.x >= LPR) return;
    const int sub = threadIdx.x;
    const int4 e = ldg_i4(reinterpret_cast<const int32_t *>(A.items + ldg(A.step_item + s) + k));
    const int row_id = e.x & 0x7FFFFFFF, buf = (int)((unsigned)e.x >> 31), gap = e.w & 0xFFFF;
    const bool is_user = row_id < S.n_user;
    const size_t row_off = (size_t)(is_user ? row_id : row_id - S.n_user) * D;
    const RowVec<V4> w = row_load<LPR, V4>((is_user ? S.U[buf] : S.V[buf]) + row_off, sub);
    const RowVec<V4> m4 = row_load<LPR, V4>((is_user ? S.mU : S.mV) + row_off, sub);
    RowVec<V4> acc = row_load<LPR, V4>(A.partial + (size_t)c0 * (D + 4), sub);
    float sse = ldg(A.partial + (size_t)c0 * (D + 4) + D);
    for (unsigned p = c0 + 1; p < c1; ++p) {
        const RowVec<V4> t = row_load<LPR, V4>(A.partial + (size_t)p * (D + 4), sub);
#pragma unroll
        for (int i = 0; i < V4; ++i) { acc.q[i].x += t.q[i].x; acc.q[i].y += t.q[i].y; acc.q[i].z += t.q[i].z; acc.q[i].w += t.q[i].w; }
        sse += ldg(A.partial + (size_t)p * (D + 4) + D);
    }
    idx_finish_row<LPR, V4>(S, A, epoch, row_id, buf, gap, w, m4, acc, sse, ldg(S.lr + epoch), sub);
}

}  // namespace ure
