// perm_chain.hip -- an epoch's batch tags made on the device by MANY workgroups per permutation, for any number of rows: what
// perm_tags.hip computes with one workgroup per shuffle (rounds of reservations: 0.6-0.76 ms per 180 k rows, at most 2^20 rows), here
// without rounds -- torch.randperm's permutation (read.py:127-133: the DataLoader's RandomSampler), its inverse, tag[f] = inverse[f] / batch,
// bit for bit what ure_host_randperm_tags computes.
//
// THE SHUFFLE WITHOUT ITS LOOP.  The inverse permutation is the product of the swaps (i, t_i), t_i = i + mt() % (n - i), applied to the
// identity from i = n - 2 down to 0 (host_rng.cpp: one_perm_tags).  Position p is touched by swap p itself -- the first to touch it,
// every older swap lies to its right -- and then by the swaps j < p with t_j = p, in falling order of j; a swap (j, q) leaves A[q] = j
// and A[j] = what q held.  With T_q = { j : t_j = q } (the swaps that TARGET q) this gives the final array in closed form:
//     inv[p] = min T_p                   when a swap j < p targets p (the last one to touch p leaves its own index there),
//            = h(p)                      otherwise: what swap p itself fetched from q = t_p, namely
//     h(p)   = p                         when t_p = p,
//            = the next larger member of T_q behind p   when there is one (that swap was the last to write q before swap p reads it),
//            = h(q)                      when p is the largest member of T_q and q is not a member: q still holds what ITS swap fetched
//                                        (h(n - 1) = n - 1: no swap starts there).
// So all that is needed is every T_q in rising order -- a grouping of the n - 1 swaps by target -- and a chase along "largest member"
// links, which is short (half a hop per row on average: a random target list has about one member).  No swap waits for another.
//
// SIX LAUNCHES (the link pass is two), stream ordered, no grid barrier.  Up to 2^20 rows (at most 1,024 buckets) no global atomic at all: the
// targets pass leaves a (tile, bucket) count matrix, the bucket pass sums its columns -- every swap's place is fixed.  Beyond: one global add per
// (tile, bucket) to count and one to reserve space:
//   shuffle_words_kernel    one workgroup per permutation walks MT19937, 623 new words per barrier (the recurrence substituted into itself
//                           twice), the raw words into t[].
//   shuffle_targets_kernel  one workgroup per (permutation, tile of 16,384 swaps): t[j] = j + temper(word) % (n - j) in place, and the
//                           swaps counted per BUCKET = range of 1,024 targets (2,048 / 4,096 / 16,384 beyond 2^24 / 2^25 / 2^26 rows;
//                           ranges of 2,048 / 4,096 targets measured 4-12 % / 35 % slower on a request's shapes).
//   shuffle_bucket_kernel   one workgroup per (permutation, tile of 16,384 swaps): the swaps go to their bucket's stretch of `pairs` as
//                           (j, t_j) -- the stretch from the prefix of the counts, a tile's place in it from one returning global
//                           add per bucket it touches, the rest from LDS cursors.  The order inside a bucket is whatever the atomics
//                           made it; nothing below depends on it.
//   shuffle_link_kernel     one workgroup per (permutation, bucket): counts its swaps per target (LDS), prefix, places them per target --
//                           in LDS when they fit (6,144 swaps; a permutation's last eight buckets, launched apart: 16,384 -- a target has
//                           ln(n / (n - q)) members on average), else in memory, read back 256 targets at a time (consecutive targets, consecutive lists: one
//                           coalesced stretch) -- and lane = target stores every member's link H[j] (the smallest member above it, a value,
//                           or "go on at q") and the target's own value (Minv: the smallest member of all).
//   shuffle_resolve_kernel  one lane per row: Minv or the chase through H, divided by the batch size, stored as uint16.
// A permutation's workgroups sit on ONE XCD (workgroup id mod 8 = permutation mod 8): its 20 bytes per row stay in that L2 while it fits.
//
// MEMORY MODEL.  Between launches: stream order.  Inside shuffle_link_kernel's in-memory path the members a workgroup placed are read back
// by other waves of the SAME workgroup: workgroup-scope release (fence + barrier) / acquire (fence), as the language defines them -- no
// assumption about which compute unit a wave is on.
#include "ure_internal.h"
#include "mt_jump_dev.h"

namespace ure {
namespace {

constexpr int kMtN = 624, kMtLag = 227;
constexpr int kDrawBlock = 640;                  // ten wavefronts: lanes 0..622 own a word of the step
constexpr int kDrawWide = 623;                   // words per dependent step
constexpr int kDrawRing = 8192;                  // the generator's words kept in LDS (a step reads 1,305 back)
constexpr int64_t kSegmentedRows = 1 << 21;      // permutations beyond this: the generator's stream in segments (mt_jump_dev.h)
constexpr int kFewBuckets = 1024;                // ... of a launch whose permutations have at most 2^20 rows
constexpr int kMaxBuckets = 16384;               // per permutation (LDS counters of the draw and bucket passes)
constexpr int kTileBlock = 1024;
constexpr int kTileLoads = 4;                    // 16-byte loads per lane: a tile of the bucket pass is 16,384 swaps
constexpr int kTile = kTileBlock * 4 * kTileLoads;
constexpr int kLinkBlock = 512;
constexpr int kLinkLoads = 8;                    // pairs a lane of the link pass has in flight
constexpr int kStage = 6144;                     // members a link workgroup keeps in LDS: three per target of its range (24 KB at 2,048 targets) ...
constexpr int kStage10 = 3072, kStage12 = 12288; // ... 12 KB at 1,024 targets (eight workgroups per CU), 48 KB at 4,096 and beyond
constexpr int kStageHeavy = 16384;               // ... and one of a permutation's last kHeavyParts buckets (64 KB)
constexpr int kHeavyParts = 8;
constexpr int kResolveBlock = 1024;
constexpr int kResolveRows = 4096;               // rows per workgroup of the resolve pass
constexpr unsigned kNone = 0xffffffffu;
constexpr unsigned kGoOn = 0x80000000u;          // H[j] = kGoOn | q: "what swap q fetched"

// A barrier that orders LDS traffic only: __syncthreads() also waits for the wave's GLOBAL stores (its fence covers every address space),
// and a words-kernel step ends with one -- ~700 cycles of an L2 round trip per step that nothing depends on.
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ unsigned mt_temper(unsigned x)
{
    x ^= x >> 11;
    x ^= (x << 7) & 0x9d2c5680u;
    x ^= (x << 15) & 0xefc60000u;
    x ^= x >> 18;
    return x;
}

// a permutation's scratch: t [n_al] (the link pass places its members there once the bucket pass has read it) | H [n_al] | Minv [n_al] |
// pairs [2 n_al] | totals [b_al] | cursors [b_al] | bases [b_al]
struct perm_view {
    unsigned *t, *H, *minv;
    uint2 *pairs;
    unsigned *totals, *cursors, *bases;
};

__host__ __device__ __forceinline__ int64_t words_per_perm(int64_t n_al, int64_t b_al) { return 5 * n_al + 3 * b_al; }

__device__ __forceinline__ perm_view view_of(unsigned *scratch, int perm, int64_t n_al, int64_t b_al)
{
    unsigned *b = scratch + (size_t)perm * (size_t)words_per_perm(n_al, b_al);
    unsigned *c = b + 5 * n_al;
    return perm_view{b, b + n_al, b + 2 * n_al, reinterpret_cast<uint2 *>(b + 3 * n_al), c, c + b_al, c + 2 * b_al};
}

// MT19937's recurrence x[p] = x[p - 227] ^ F(p), F(p) = f(x[p - 624], x[p - 623]) (f: the twist's linear part), gives 227 new words per
// dependent step.  Substituted into itself twice,
//     x[p] = x[p - 681] ^ F(p - 454) ^ F(p - 227) ^ F(p),
// a step may be as wide as F(p) itself allows: 623 words (x[p - 623] must exist) -- 289 barriers per 180 k outputs instead of 793.  (The
// first 454 words behind the seed block have no x[p - 681]: they take the form with one / two terms.)
__device__ __forceinline__ unsigned mt_mix(unsigned a, unsigned b)
{
    const unsigned y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

// The seed blocks of a launch whose permutations are cut into segments (beyond kSegmentedRows rows): segment 0 of every permutation =
// init_genrand(seed), the other segments zero -- the jump tree (mt_jump_dev.h) adds their start blocks into them.
__global__ __launch_bounds__(kDrawBlock) void shuffle_seed_kernel(const ure_perm_t *__restrict__ perms, int n_perms, unsigned *__restrict__ states, int J)
{
    const int perm = blockIdx.x, tid = threadIdx.x;
    if (perm >= n_perms) return;
    unsigned *mine = states + (size_t)perm * J * kMtN;
    for (int k = kMtN + tid; k < J * kMtN; k += kDrawBlock) mine[k] = 0u;
    if (tid == 0) {
        unsigned v = (unsigned)((unsigned long long)perms[perm].seed & 0xffffffffull);
        mine[0] = v;
        for (int j = 1; j < kMtN; ++j) {
            v = 1812433253u * (v ^ (v >> 30)) + (unsigned)j;
            mine[j] = v;
        }
    }
}

// One workgroup per (permutation, segment): the generator's raw words, in order, into t[] (the targets pass turns them into targets in
// place); the words live in a ring of 8,192 in LDS.  states == nullptr: ONE segment, from the permutation's seed (at::mt19937(seed):
// init_genrand; the first draw regenerates); else segment `seg` starts behind the block states[perm][seg].  Segment 0 also clears the
// permutation's bucket counters.
__global__ __launch_bounds__(kDrawBlock) void shuffle_words_kernel(const ure_perm_t *__restrict__ perms, int n_perms, unsigned *__restrict__ scratch, int64_t n_al,
                                                                   int64_t b_al, int range_log2, const unsigned *__restrict__ states, int J)
{
    constexpr unsigned M = kDrawRing - 1;
    __shared__ unsigned ring[kDrawRing];
    const int tid = threadIdx.x;
    const int perm = blockIdx.x / J, seg = blockIdx.x % J;
    if (perm >= n_perms) return;
    const int n = perms[perm].n;
    if (n <= 0) return;
    const perm_view P = view_of(scratch, perm, n_al, b_al);
    if (seg == 0) {
        const int n_buckets = (int)((((int64_t)n - 1) >> range_log2) + 1);
        for (int b = tid; b < n_buckets; b += kDrawBlock) {
            P.totals[b] = 0u;
            P.cursors[b] = 0u;
        }
    }
    const int64_t w_lo = (int64_t)seg * jmp::kSegWords;
    const int64_t total_all = (int64_t)n - 1;
    if (w_lo >= total_all) return;
    const int total = (int)min<int64_t>(states ? jmp::kSegWords : total_all, total_all - w_lo);       // this segment's words
    if (states) {
        const unsigned *blk = states + ((size_t)perm * J + seg) * kMtN;
        for (int k = tid; k < kMtN; k += kDrawBlock) ring[((unsigned)k - kMtN) & M] = blk[k];
    } else if (tid == 0) {
        unsigned v = (unsigned)((unsigned long long)perms[perm].seed & 0xffffffffull);
        ring[(0u - kMtN) & M] = v;
        for (int j = 1; j < kMtN; ++j) {
            v = 1812433253u * (v ^ (v >> 30)) + (unsigned)j;
            ring[((unsigned)j - kMtN) & M] = v;
        }
    }
    __syncthreads();
    unsigned *out = P.t + w_lo;
#pragma unroll 1
    for (int g0 = 0; g0 < total; g0 += kDrawWide) {
        const unsigned g = (unsigned)(g0 + tid);
        if (tid < kDrawWide && (int)g < total) {
            // (all seven reads at once -- one LDS round trip per step; the first 454 words behind the start block leave the terms they do not have out)
            const bool two = g >= (unsigned)kMtLag, three = g >= 2u * kMtLag;
            const unsigned a0 = ring[(g - kMtN) & M], a1 = ring[(g - kMtN + 1) & M];
            const unsigned b0 = ring[(g - kMtLag - kMtN) & M], b1 = ring[(g - kMtLag - kMtN + 1) & M];
            const unsigned c0 = ring[(g - 2 * kMtLag - kMtN) & M], c1 = ring[(g - 2 * kMtLag - kMtN + 1) & M];
            const unsigned x = ring[(g - (three ? 3u * kMtLag : two ? 2u * kMtLag : (unsigned)kMtLag)) & M];
            const unsigned v = x ^ mt_mix(a0, a1) ^ (two ? mt_mix(b0, b1) : 0u) ^ (three ? mt_mix(c0, c1) : 0u);
            ring[g & M] = v;
            out[g] = v;
        }
        lds_barrier();
    }
}

// One workgroup per (permutation, tile of 16,384 swaps): raw word -> target, t[j] = j + temper(word) % (n - j), in place, and the swaps
// per BUCKET (range of 2^range_log2 targets) added to the permutation's counters -- one global add per bucket the tile touches.
template <int CAP>
__global__ __launch_bounds__(kTileBlock) void shuffle_targets_kernel(const ure_perm_t *__restrict__ perms, int n_perms, int parts_max, unsigned *__restrict__ scratch,
                                                                      int64_t n_al, int64_t b_al, int range_log2, unsigned *__restrict__ mat)
{
    __shared__ unsigned cnt[CAP];                               // (CAP = the launch's most buckets per permutation: 4 KB of LDS up to 2^20 rows, not 64)
    const int tid = threadIdx.x;
    const int x8 = blockIdx.x & 7, l = blockIdx.x >> 3;
    const int perm = (l / parts_max) * 8 + x8, part = l % parts_max;
    if (perm >= n_perms) return;
    const int n = perms[perm].n;
    const int64_t j0l = (int64_t)part * kTile;
    if (j0l >= (int64_t)n - 1) return;
    const perm_view P = view_of(scratch, perm, n_al, b_al);
    const int n_buckets = (int)((((int64_t)n - 1) >> range_log2) + 1);
    const unsigned j0 = (unsigned)j0l, j_end = (unsigned)(n - 1);
    const int b_lo = (int)(j0 >> range_log2);                   // t_j >= j: the tile's swaps target buckets b_lo .. n_buckets - 1
    for (int b = b_lo + tid; b < n_buckets; b += kTileBlock) cnt[b] = 0u;
    __syncthreads();
    uint4 *t4 = reinterpret_cast<uint4 *>(P.t);
    uint4 v[kTileLoads];
#pragma unroll
    for (int u = 0; u < kTileLoads; ++u) {
        const unsigned j = j0 + 4u * (unsigned)(tid + u * kTileBlock);
        v[u] = j < j_end ? t4[j >> 2] : make_uint4(0u, 0u, 0u, 0u);
    }
    auto target = [&](unsigned j, unsigned word) {
        if (j >= j_end) return word;                            // (beyond the last swap: left as it is)
        const unsigned q = j + mt_temper(word) % ((unsigned)n - j);
        atomicAdd(&cnt[q >> range_log2], 1u);
        return q;
    };
#pragma unroll
    for (int u = 0; u < kTileLoads; ++u) {
        const unsigned j = j0 + 4u * (unsigned)(tid + u * kTileBlock);
        if (j < j_end) {
            v[u].x = target(j, v[u].x);
            v[u].y = target(j + 1, v[u].y);
            v[u].z = target(j + 2, v[u].z);
            v[u].w = target(j + 3, v[u].w);
            t4[j >> 2] = v[u];
        }
    }
    __syncthreads();
    if (CAP == kFewBuckets) {
        // few buckets: the tile's counts into its row of the permutation's (tile, bucket) matrix -- the bucket pass sums the columns, no atomic
        unsigned *row = mat + ((size_t)perm * parts_max + part) * kFewBuckets;
        for (int b = tid; b < n_buckets; b += kTileBlock) row[b] = b >= b_lo ? cnt[b] : 0u;
        return;
    }
    for (int b = b_lo + tid; b < n_buckets; b += kTileBlock) {
        const unsigned c = cnt[b];
        if (c) atomicAdd(&P.totals[b], c);
    }
}

// workgroup id -> (permutation, part): a permutation's workgroups have id mod 8 = permutation mod 8 (one XCD), parts in launch order
__device__ __forceinline__ bool place_of(int parts_max, int n_perms, int *perm, int *part)
{
    const int x = blockIdx.x & 7, l = blockIdx.x >> 3;
    *perm = (l / parts_max) * 8 + x;
    *part = l % parts_max;
    return *perm < n_perms;
}

// exclusive prefix of v[0 .. n) in place, `Block` lanes, n a few thousand: lane = element (no bank conflicts), a round per `Block` elements
template <int Block>
__device__ __forceinline__ unsigned prefix_in_place(unsigned *v, int n, unsigned *wave_sum, unsigned *carry_word)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) *carry_word = 0u;
    __syncthreads();
    for (int base = 0; base < n; base += Block) {
        const int x = base + tid;
        const unsigned c = x < n ? v[x] : 0u;
        unsigned s = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned o = __shfl_up(s, d, 64);
            if (lane >= d) s += o;
        }
        if (lane == 63) wave_sum[wave] = s;
        __syncthreads();
        unsigned before = *carry_word;
        for (int k = 0; k < wave; ++k) before += wave_sum[k];
        if (x < n) v[x] = before + s - c;
        __syncthreads();
        if (tid == Block - 1) *carry_word = before + s;
    }
    __syncthreads();
    return *carry_word;
}

template <int CAP>
__global__ __launch_bounds__(kTileBlock) void shuffle_bucket_kernel(const ure_perm_t *__restrict__ perms, int n_perms, int parts_max, unsigned *__restrict__ scratch,
                                                                     int64_t n_al, int64_t b_al, int range_log2, const unsigned *__restrict__ mat)
{
    __shared__ unsigned cur[CAP];                               // the tile's count per bucket, then its cursor in the bucket's stretch
    __shared__ unsigned base[CAP];
    __shared__ unsigned wave_sum[kTileBlock / 64];
    __shared__ unsigned s_carry;
    const int tid = threadIdx.x;
    int perm, part;
    if (!place_of(parts_max, n_perms, &perm, &part)) return;
    const int n = perms[perm].n;
    const int64_t j0l = (int64_t)part * kTile;
    if (j0l >= (int64_t)n - 1) {
        if (!(part == 0 && n == 1)) return;                     // (one row: no swap, but its bucket's base is read by the link pass)
    }
    const perm_view P = view_of(scratch, perm, n_al, b_al);
    const int n_buckets = (int)((((int64_t)n - 1) >> range_log2) + 1);
    const unsigned j0 = (unsigned)j0l, j_end = (unsigned)(n - 1);
    const int b_lo = (int)(j0 >> range_log2);                   // t_j >= j: the tile's swaps target buckets b_lo .. n_buckets - 1
    const uint4 *t4 = reinterpret_cast<const uint4 *>(P.t);
    uint4 v[kTileLoads];
#pragma unroll
    for (int u = 0; u < kTileLoads; ++u) {
        const unsigned j = j0 + 4u * (unsigned)(tid + u * kTileBlock);
        v[u] = j < j_end ? t4[j >> 2] : make_uint4(0u, 0u, 0u, 0u);
    }
    if (CAP == kFewBuckets) {
        // few buckets: a bucket's swaps and the ones of the tiles before this one from the columns of the (tile, bucket) matrix the targets
        // pass left -- every tile's place in every stretch is fixed, no atomic, no cursor
        const int n_tiles = (int)(((int64_t)n - 1 + kTile - 1) / kTile);
        const unsigned *m = mat + (size_t)perm * parts_max * kFewBuckets;
        for (int b = tid; b < n_buckets; b += kTileBlock) {
            unsigned all = 0u, before = 0u;
            for (int t = 0; t < n_tiles; ++t) {
                const unsigned c = m[(size_t)t * kFewBuckets + b];
                before += t < part ? c : 0u;
                all += c;
            }
            base[b] = all;
            cur[b] = before;
            if (part == 0) P.totals[b] = all;
        }
        prefix_in_place<kTileBlock>(base, n_buckets, wave_sum, &s_carry);
        for (int b = tid; b < n_buckets; b += kTileBlock) {
            if (part == 0) P.bases[b] = base[b];
            cur[b] += base[b];
        }
        __syncthreads();
    } else {
        for (int b = tid; b < n_buckets; b += kTileBlock) {
            cur[b] = 0u;
            base[b] = P.totals[b];
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kTileLoads; ++u) {
            const unsigned j = j0 + 4u * (unsigned)(tid + u * kTileBlock);
            if (j < j_end) atomicAdd(&cur[v[u].x >> range_log2], 1u);
            if (j + 1 < j_end) atomicAdd(&cur[v[u].y >> range_log2], 1u);
            if (j + 2 < j_end) atomicAdd(&cur[v[u].z >> range_log2], 1u);
            if (j + 3 < j_end) atomicAdd(&cur[v[u].w >> range_log2], 1u);
        }
        prefix_in_place<kTileBlock>(base, n_buckets, wave_sum, &s_carry);    // (its first barrier ends the counting)
        if (part == 0)
            for (int b = tid; b < n_buckets; b += kTileBlock) P.bases[b] = base[b];
        for (int b = b_lo + tid; b < n_buckets; b += kTileBlock) {
            const unsigned c = cur[b];
            if (c) cur[b] = base[b] + atomicAdd(&P.cursors[b], c);  // the tile's place in the bucket's stretch (where it lies does not matter)
        }
        __syncthreads();
    }
    auto put = [&](unsigned j, unsigned q) { P.pairs[atomicAdd(&cur[q >> range_log2], 1u)] = make_uint2(j, q); };
#pragma unroll
    for (int u = 0; u < kTileLoads; ++u) {
        const unsigned j = j0 + 4u * (unsigned)(tid + u * kTileBlock);
        if (j < j_end) put(j, v[u].x);
        if (j + 1 < j_end) put(j + 1, v[u].y);
        if (j + 2 < j_end) put(j + 2, v[u].z);
        if (j + 3 < j_end) put(j + 3, v[u].w);
    }
}

// exclusive prefix of v[0 .. Block * Per) in place: a lane sums its Per consecutive elements, ONE scan of the lanes' sums, the lane writes
// its elements' prefixes back.  -> the total.
template <int Block, int Per>
__device__ __forceinline__ unsigned prefix_by_lane(unsigned *v, unsigned *wave_sum)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned c[Per];
    unsigned mine = 0u;
#pragma unroll
    for (int k = 0; k < Per; ++k) {
        c[k] = v[tid * Per + k];
        mine += c[k];
    }
    unsigned s = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned o = __shfl_up(s, d, 64);
        if (lane >= d) s += o;
    }
    if (lane == 63) wave_sum[wave] = s;
    __syncthreads();
    unsigned before = 0u, total = 0u;
#pragma unroll
    for (int k = 0; k < Block / 64; ++k) {
        const unsigned ws = wave_sum[k];
        if (k < wave) before += ws;
        total += ws;
    }
    unsigned run = before + s - mine;
#pragma unroll
    for (int k = 0; k < Per; ++k) {
        v[tid * Per + k] = run;
        run += c[k];
    }
    __syncthreads();
    return total;
}

// Stage: the members a workgroup keeps in LDS.  The last kHeavyParts buckets of a permutation hold several times the average (a target
// near the end has ln(n / (n - q)) members): they are launched apart with a big stage, the others with a small one -- five workgroups
// per compute unit instead of two.
template <int RL, int Stage>
__global__ __launch_bounds__(kLinkBlock) void shuffle_link_kernel(const ure_perm_t *__restrict__ perms, int n_perms, int parts_max, unsigned *__restrict__ scratch,
                                                                   int64_t n_al, int64_t b_al, int heavy)
{
    constexpr int kRange = 1 << RL, kPer = kRange / kLinkBlock;
    __shared__ unsigned off[kRange];
    __shared__ unsigned stage[Stage];
    __shared__ unsigned wave_sum[kLinkBlock / 64];
    const int tid = threadIdx.x;
    int perm, part;
    if (!place_of(parts_max, n_perms, &perm, &part)) return;
    const int n = perms[perm].n;
    const int64_t q0l = (int64_t)part << RL;
    if (q0l >= n) return;
    const int n_buckets = (int)((((int64_t)n - 1) >> RL) + 1);
    if ((part >= n_buckets - kHeavyParts) != (heavy != 0)) return;
    const unsigned q0 = (unsigned)q0l, q1 = (unsigned)min<int64_t>(n, q0l + kRange);
    const int R = (int)(q1 - q0);
    const perm_view P = view_of(scratch, perm, n_al, b_al);
    const unsigned count = P.totals[part], at = P.bases[part];
    const uint2 *mine = P.pairs + at;
#pragma unroll
    for (int k = 0; k < kPer; ++k) off[k * kLinkBlock + tid] = 0u;
    __syncthreads();
    // ---- members per target (kLinkLoads pairs per lane in flight; a bucket of up to 2,048 swaps stays in registers for the second pass)
    uint2 first[kLinkLoads];
#pragma unroll
    for (int u = 0; u < kLinkLoads; ++u) {
        const unsigned e = (unsigned)(u * kLinkBlock + tid);
        first[u] = e < count ? mine[e] : make_uint2(0u, 0u);
    }
#pragma unroll
    for (int u = 0; u < kLinkLoads; ++u)
        if ((unsigned)(u * kLinkBlock + tid) < count) atomicAdd(&off[first[u].y - q0], 1u);
    for (unsigned e0 = kLinkBlock * kLinkLoads; e0 < count; e0 += kLinkBlock * kLinkLoads) {
        uint2 m[kLinkLoads];
#pragma unroll
        for (int u = 0; u < kLinkLoads; ++u) {
            const unsigned e = e0 + (unsigned)(u * kLinkBlock + tid);
            m[u] = e < count ? mine[e] : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < kLinkLoads; ++u)
            if (e0 + (unsigned)(u * kLinkBlock + tid) < count) atomicAdd(&off[m[u].y - q0], 1u);
    }
    __syncthreads();
    prefix_by_lane<kLinkBlock, kPer>(off, wave_sum);
    // ---- the members, target by target (off[x] becomes the END of target x's list; its start is off[x - 1])
    const bool in_lds = count <= (unsigned)Stage;               // (uniform)
    unsigned *placed = P.t + at;                                // (the bucket pass is done with t; a bucket's members take the stretch its pairs have)
    auto place = [&](const uint2 m) {
        const unsigned pos = atomicAdd(&off[m.y - q0], 1u);
        if (in_lds)
            stage[pos] = m.x;
        else
            placed[pos] = m.x;
    };
#pragma unroll
    for (int u = 0; u < kLinkLoads; ++u)
        if ((unsigned)(u * kLinkBlock + tid) < count) place(first[u]);
    for (unsigned e0 = kLinkBlock * kLinkLoads; e0 < count; e0 += kLinkBlock * kLinkLoads) {
        uint2 m[kLinkLoads];
#pragma unroll
        for (int u = 0; u < kLinkLoads; ++u) {
            const unsigned e = e0 + (unsigned)(u * kLinkBlock + tid);
            m[u] = e < count ? mine[e] : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < kLinkLoads; ++u)
            if (e0 + (unsigned)(u * kLinkBlock + tid) < count) place(m[u]);
    }
    if (in_lds) {
        __syncthreads();
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    // ---- the lists; lane = target: every member's link is the smallest member above it (the order inside a list is whatever the LDS
    // atomics made it), the smallest member of all is the target's value
    const unsigned last = (unsigned)(n - 1);
    for (int b0 = 0; b0 < R; b0 += kLinkBlock) {
        unsigned seg0 = 0u;
        bool staged = in_lds;
        if (!in_lds) {
            // consecutive targets, consecutive lists: the members of these 256 targets are ONE stretch of `placed`, read coalesced
            seg0 = b0 ? off[b0 - 1] : 0u;
            const unsigned seg1 = off[min(b0 + kLinkBlock, R) - 1];
            staged = seg1 - seg0 <= (unsigned)Stage;            // (uniform; a denser stretch is read from memory)
            if (staged)
                for (unsigned i = tid; i < seg1 - seg0; i += kLinkBlock) stage[i] = placed[seg0 + i];
            __syncthreads();
        }
        const int x = b0 + tid;
        if (x < R) {
            const unsigned e1 = off[x], e0 = x ? off[x - 1] : 0u;
            const unsigned q = q0 + (unsigned)x;
            unsigned least = kNone;
            for (unsigned i = e0; i < e1; ++i) {
                const unsigned a = staged ? stage[i - seg0] : placed[i];
                least = min(least, a);
                unsigned succ = kNone;
                for (unsigned m = e0; m < e1; ++m) {
                    const unsigned o = staged ? stage[m - seg0] : placed[m];
                    if (o > a) succ = min(succ, o);
                }
                P.H[a] = succ != kNone ? succ : (a == q ? a : (q == last ? last : (kGoOn | q)));
            }
            P.minv[q] = least;
        }
        if (!in_lds) __syncthreads();
    }
}

__global__ __launch_bounds__(kResolveBlock) void shuffle_resolve_kernel(const ure_perm_t *__restrict__ perms, int n_perms, int parts_max, unsigned *__restrict__ scratch,
                                                                         int64_t n_al, int64_t b_al, unsigned *__restrict__ broken)
{
    int perm, part;
    if (!place_of(parts_max, n_perms, &perm, &part)) return;
    const int n = perms[perm].n;
    const int64_t lo = (int64_t)part * kResolveRows;
    if (lo >= n) return;
    const perm_view P = view_of(scratch, perm, n_al, b_al);
    const unsigned batch = (unsigned)perms[perm].batch;
    uint16_t *out = perms[perm].tags;
    const unsigned hi = (unsigned)min<int64_t>(n, lo + kResolveRows);
    constexpr int U = 4;                                        // rows a lane has in flight
    for (unsigned p0 = (unsigned)lo + threadIdx.x; p0 < hi; p0 += U * kResolveBlock) {
        unsigned v[U], h[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned p = p0 + (unsigned)(u * kResolveBlock);
            v[u] = p < hi ? P.minv[p] : 0u;
            h[u] = p < hi && p != (unsigned)(n - 1) ? P.H[p] : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned p = p0 + (unsigned)(u * kResolveBlock);
            if (p >= hi) continue;
            unsigned r = v[u];
            if (r == kNone) {
                if (p == (unsigned)(n - 1)) {
                    r = p;
                } else {
                    // rising positions: the chase ends at a value, at the latest below n - 1.  (A link that does not rise cannot come out
                    // of the link pass; should memory ever hold one, the row gets the tag that matches NO batch and the flag the host reads
                    // -- ultrare_amd.engine.TrainJob.check_tags -- instead of a walk without end.)
                    unsigned at = p;
                    r = h[u];
                    while (r & kGoOn) {
                        const unsigned q = r & ~kGoOn;
                        if (q <= at || q >= (unsigned)(n - 1)) {
                            r = kNone;
                            break;
                        }
                        at = q;
                        r = P.H[q];
                    }
                }
            }
            if (r >= (unsigned)n) {
                *broken = 0xdeadu;
                out[p] = (uint16_t)0xFFFFu;
            } else {
                out[p] = (uint16_t)(r / batch);
            }
        }
    }
}

int range_log2_of(int64_t n_max, int32_t wanted)
{
    if (wanted) return wanted;
    return n_max <= ((int64_t)kMaxBuckets << 10) ? 10 : n_max <= ((int64_t)kMaxBuckets << 11) ? 11 : n_max <= ((int64_t)kMaxBuckets << 12) ? 12 : 14;
}

int64_t buckets_al(int64_t n_max, int rl) { return (((n_max - 1) >> rl) + 1 + 63) / 64 * 64; }

}  // namespace
}  // namespace ure

extern "C" int64_t ure_device_shuffle_tags_flag(int64_t n_max, int32_t n_perms)
{
    if (n_max <= 0 || n_perms <= 0) return 0;
    const int64_t n_al = (n_max + 63) / 64 * 64;
    return ure::words_per_perm(n_al, std::min<int64_t>(ure::kMaxBuckets, ure::buckets_al(n_max, 10))) * (int64_t)n_perms;       // (sized for the finest ranges)
}

extern "C" int64_t ure_device_shuffle_tags_scratch(int64_t n_max, int32_t n_perms)
{
    if (n_max <= 0 || n_perms <= 0) return 0;
    const int64_t J = n_max > ure::kSegmentedRows ? (n_max - 1 + ure::jmp::kSegWords - 1) / ure::jmp::kSegWords : 0;
    const int64_t tiles = std::max<int64_t>(1, (n_max - 1 + ure::kTile - 1) / ure::kTile);
    // (+ the segments' start blocks, + the (tile, bucket) count matrices of launches of few buckets)
    return ure_device_shuffle_tags_flag(n_max, n_perms) + 64 + (int64_t)n_perms * J * ure::kMtN + (int64_t)n_perms * tiles * ure::kFewBuckets;
}

extern "C" int ure_device_shuffle_tags(const ure_perm_t *perms, int32_t n_perms, int64_t n_max, uint32_t *scratch, int64_t scratch_words, int32_t range_log2,
                                       void *stream)
{
    using namespace ure;
    URE_ARG(n_perms >= 0 && n_max >= 0 && (range_log2 == 0 || range_log2 == 10 || range_log2 == 11 || range_log2 == 12 || range_log2 == 14));
    if (n_perms == 0 || n_max == 0) return 0;
    URE_ARG(perms && scratch);
    const int rl = range_log2_of(n_max, range_log2);
    if (((n_max - 1) >> rl) + 1 > kMaxBuckets)
        return fail(-1, "ure_device_shuffle_tags: %lld rows: more than %d ranges of %d targets", (long long)n_max, kMaxBuckets, 1 << rl);
    if (scratch_words < ure_device_shuffle_tags_scratch(n_max, n_perms)) return fail(-1, "ure_device_shuffle_tags: scratch too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t n_al = (n_max + 63) / 64 * 64, b_al = buckets_al(n_max, rl);
    uint32_t *broken = scratch + ure_device_shuffle_tags_flag(n_max, n_perms);     // (never cleared here: the caller that reads it clears it when it makes the scratch)
    const int64_t slots = ((int64_t)n_perms + 7) / 8;
    const int64_t tiles = std::max<int64_t>(1, (n_max - 1 + kTile - 1) / kTile), ranges = ((n_max - 1) >> rl) + 1, rows = (n_max + kResolveRows - 1) / kResolveRows;
    if (8 * slots * std::max(ranges, rows) > 0x7fffffffll)
        return fail(-1, "ure_device_shuffle_tags: %d permutations of up to %lld rows are too many for one launch", (int)n_perms, (long long)n_max);
    if (n_max > kSegmentedRows) {
        // millions of rows: the generator's stream in segments of 1,024 blocks, their start blocks by the jump tree -- one workgroup walking
        // 22.5 M words took 12.5 ms of an 18.5 ms shuffle
        const int64_t J = (n_max - 1 + jmp::kSegWords - 1) / jmp::kSegWords;
        uint32_t *states = broken + 64;
        const uint32_t *levels = nullptr;
        if (jmp::device_levels(jmp::levels_of(J), st, &levels)) return fail(-1, "ure_device_shuffle_tags: the jump polynomials could not be made / uploaded");
        hipLaunchKernelGGL(shuffle_seed_kernel, dim3((unsigned)n_perms), dim3(kDrawBlock), 0, st, perms, (int)n_perms, states, (int)J);
        jmp::launch_tree(states, (int)n_perms, J, levels, st);
        hipLaunchKernelGGL(shuffle_words_kernel, dim3((unsigned)(n_perms * J)), dim3(kDrawBlock), 0, st, perms, (int)n_perms, scratch, n_al, b_al, rl, states, (int)J);
    } else {
        hipLaunchKernelGGL(shuffle_words_kernel, dim3((unsigned)n_perms), dim3(kDrawBlock), 0, st, perms, (int)n_perms, scratch, n_al, b_al, rl,
                           static_cast<const unsigned *>(nullptr), 1);
    }
    const bool few = ranges <= kFewBuckets;                     // (buckets per permutation: small LDS tables, a count matrix instead of global atomics)
    const int64_t J_all = n_max > kSegmentedRows ? (n_max - 1 + jmp::kSegWords - 1) / jmp::kSegWords : 0;
    uint32_t *mat = broken + 64 + (int64_t)n_perms * J_all * kMtN;
    if (few)
        hipLaunchKernelGGL(shuffle_targets_kernel<kFewBuckets>, dim3((unsigned)(8 * slots * tiles)), dim3(kTileBlock), 0, st, perms, (int)n_perms, (int)tiles, scratch, n_al, b_al, rl, mat);
    else
        hipLaunchKernelGGL(shuffle_targets_kernel<kMaxBuckets>, dim3((unsigned)(8 * slots * tiles)), dim3(kTileBlock), 0, st, perms, (int)n_perms, (int)tiles, scratch, n_al, b_al, rl, mat);
    if (few)
        hipLaunchKernelGGL(shuffle_bucket_kernel<kFewBuckets>, dim3((unsigned)(8 * slots * tiles)), dim3(kTileBlock), 0, st, perms, (int)n_perms, (int)tiles, scratch, n_al, b_al, rl, mat);
    else
        hipLaunchKernelGGL(shuffle_bucket_kernel<kMaxBuckets>, dim3((unsigned)(8 * slots * tiles)), dim3(kTileBlock), 0, st, perms, (int)n_perms, (int)tiles, scratch, n_al, b_al, rl, mat);
    for (int heavy = 0; heavy < 2; ++heavy) {
        const dim3 grid((unsigned)(8 * slots * ranges)), block(kLinkBlock);
        if (rl == 10 && !heavy) hipLaunchKernelGGL((shuffle_link_kernel<10, kStage10>), grid, block, 0, st, perms, (int)n_perms, (int)ranges, scratch, n_al, b_al, heavy);
        if (rl == 10 && heavy) hipLaunchKernelGGL((shuffle_link_kernel<10, kStageHeavy>), grid, block, 0, st, perms, (int)n_perms, (int)ranges, scratch, n_al, b_al, heavy);
        if (rl == 11 && !heavy) hipLaunchKernelGGL((shuffle_link_kernel<11, kStage>), grid, block, 0, st, perms, (int)n_perms, (int)ranges, scratch, n_al, b_al, heavy);
        if (rl == 11 && heavy) hipLaunchKernelGGL((shuffle_link_kernel<11, kStageHeavy>), grid, block, 0, st, perms, (int)n_perms, (int)ranges, scratch, n_al, b_al, heavy);
        if (rl == 12 && !heavy) hipLaunchKernelGGL((shuffle_link_kernel<12, kStage12>), grid, block, 0, st, perms, (int)n_perms, (int)ranges, scratch, n_al, b_al, heavy);
        if (rl == 12 && heavy) hipLaunchKernelGGL((shuffle_link_kernel<12, kStageHeavy>), grid, block, 0, st, perms, (int)n_perms, (int)ranges, scratch, n_al, b_al, heavy);
        if (rl == 14 && !heavy) hipLaunchKernelGGL((shuffle_link_kernel<14, kStage12>), grid, block, 0, st, perms, (int)n_perms, (int)ranges, scratch, n_al, b_al, heavy);
        if (rl == 14 && heavy) hipLaunchKernelGGL((shuffle_link_kernel<14, kStageHeavy>), grid, block, 0, st, perms, (int)n_perms, (int)ranges, scratch, n_al, b_al, heavy);
    }
    hipLaunchKernelGGL(shuffle_resolve_kernel, dim3((unsigned)(8 * slots * rows)), dim3(kResolveBlock), 0, st, perms, (int)n_perms, (int)rows, scratch, n_al, b_al, broken);
    URE_HIP(hipGetLastError());
    return 0;
}
