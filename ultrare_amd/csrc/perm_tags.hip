// perm_tags.hip -- an epoch's batch tags made ON THE DEVICE from the epoch's seed: torch.randperm's permutation (read.py:127-133: the
// DataLoader's RandomSampler; ATen's randperm for n < 2^32 / 20 is a Fisher-Yates shuffle driven by MT19937), its inverse, and
// tag[f] = inverse[f] / batch (struct ure_shard: file_tags) -- what ure_host_randperm_tags computes on the host, bit for bit.
//
// The shuffle looks sequential -- swap i depends on every swap before it -- but only through the positions two swaps share, and a
// random shuffle's dependence chains are O(log n) long.  It is run with DETERMINISTIC RESERVATIONS (Shun, Gu, Blelloch, Fineman,
// Gibbons: "Sequential random permutation, list contraction and tree contraction are highly parallel", SODA 2015): a few thousand
// pending swaps, the oldest ones, each write their index with atomicMin into a reservation word of their two positions; a swap that
// finds its own index in both has no older pending swap touching either position and is carried out, the others try again in the
// next round together with newly admitted ones.  Every swap therefore sees exactly the array the sequential loop would show it: the
// result is the sequential loop's permutation, whatever the timing.
//
// The INVERSE permutation is what a tag needs (the step of file row f is its position in the epoch / batch), and the inverse of a
// product of transpositions is the same transpositions in the opposite order: the swaps (i, i + z_i) run from i = n - 2 down to 0 on
// the identity (as host_rng.cpp: one_perm_tags).  The partners z_i = mt() % (n - i) do not depend on the swaps, so all of them are
// drawn first: MT19937's block recurrence st[k] = st[k + 397] ^ f(st[k], st[k + 1]) is three 227-wide parallel phases per 624 outputs.
//
// One workgroup of 1,024 lanes makes one permutation at a time (its state -- partners and inverse: 8 bytes per row -- stays in its
// XCD's L2; the reservation words are in LDS) and takes the next of the launch when it is done; `groups` workgroups work side by side.
//
// MEMORY MODEL.  inv[] and z[] are exchanged between the WAVES OF ONE WORKGROUP through global memory with plain loads and stores.  The
// rule this rests on is LLVM's AMDGPU memory model for gfx90a / gfx942 / gfx950 (AMDGPUUsage, "Memory Model", code sequences for
// workgroup scope in NON-tgsplit mode): all waves of a workgroup run on one compute unit and share its vector L1, so
//     release at workgroup scope  =  s_waitcnt vmcnt(0) (the wave's stores have reached the L1 / L2) before the s_barrier,
//     acquire at workgroup scope  =  nothing (no buffer_inv: a load cannot hit a line staler than a store of the same CU),
// which is exactly what stands below: `s_waitcnt vmcnt(0)` in front of every barrier that ends a round, no cache invalidate behind
// it.  (A __threadfence_block() would emit the same release but the compiler may keep it weaker than vmcnt(0) for LDS-only
// traffic; an agent-scope fence writes the L2 back: 27 -> 7.5 us per round, measured.)  In THREADGROUP-SPLIT mode (-mtgsplit) the
// waves of a workgroup may sit on different CUs, workgroup-scope acquire becomes a buffer_inv sc0 and this code would read stale
// lines SILENTLY -- a wrong permutation, which the give-up flag cannot see.  Such a build is refused (below).
#include "ure_internal.h"

// (hipcc 7.2 defines no macro for the tgsplit target feature, so the refusal is made where the build is: ultrare_amd/build.py rejects
// -mtgsplit, and tests/test_cpu_host.py reads the TG_SPLIT bit of every kernel descriptor of the built library.)

namespace ure {
namespace {

constexpr int kPermBlock = 1024;
constexpr int kPermInFlight = 4;                 // pending swaps per lane: 4,096 per permutation
constexpr int kMtN = 624, kMtM = 397;
constexpr unsigned kPrioBits = 20;               // a swap's index: n <= 2^20 rows (larger shards keep the host path); 4,096 rounds between wipes
constexpr unsigned kRounds = 1u << (32 - kPrioBits);
constexpr int kResWords = 16384;                // reservation words in LDS (64 KB)

__device__ __forceinline__ unsigned mt_twist(unsigned a, unsigned b, unsigned far)
{
    const unsigned y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

__device__ __forceinline__ unsigned mt_temper(unsigned x)
{
    x ^= x >> 11;
    x ^= (x << 7) & 0x9d2c5680u;
    x ^= (x << 15) & 0xefc60000u;
    x ^= x >> 18;
    return x;
}

__global__ __launch_bounds__(kPermBlock) void perm_tags_kernel(const ure_perm_t *__restrict__ perms, int n_perms, unsigned *__restrict__ scratch, int64_t words_per_group,
                                                               unsigned *__restrict__ gave_up)
{
    __shared__ unsigned st[2][kMtN];
    __shared__ unsigned next_p;
    // The reservation words live in LDS, one per position modulo 16,384: two positions that share a word are taken for one -- an extra
    // conflict now and then (the younger swap waits a round), never a wrong one -- and a round costs no memory traffic for them
    // (in global memory every pending swap made 4 L2 requests per round: 27 us per round of 4,096 swaps, 1.8 ms per 180 k-row shuffle).
    __shared__ unsigned res[kResWords];
    const int tid = threadIdx.x;
    unsigned *z = scratch + (size_t)blockIdx.x * words_per_group;
    unsigned *inv = z + (words_per_group / 2);
    for (int f = tid; f < kResWords; f += kPermBlock) res[f] = 0xffffffffu;
    unsigned round = kRounds - 1;                               // counts down: a later round's reservations are SMALLER than any left behind
    for (int perm = blockIdx.x; perm < n_perms; perm += gridDim.x) {
        const int n = perms[perm].n, batch = perms[perm].batch;
        // ---- the partners: z[i] = mt() % (n - i), i = 0 .. n - 2, in generator order
        if (tid == 0) {
            unsigned v = (unsigned)((unsigned long long)perms[perm].seed & 0xffffffffull);
            st[0][0] = v;
            for (int j = 1; j < kMtN; ++j) {
                v = 1812433253u * (v ^ (v >> 30)) + (unsigned)j;
                st[0][j] = v;
            }
        }
        __syncthreads();
        int cur = 0;
        for (int base = 0; base < n - 1; base += kMtN) {
            const unsigned *o = st[cur];
            unsigned *w = st[cur ^ 1];
            if (tid < kMtN - kMtM) w[tid] = mt_twist(o[tid], o[tid + 1], o[tid + kMtM]);                                  // k in [0, 227): the far word is an old one
            __syncthreads();
            if (tid < kMtN - kMtM) w[tid + 227] = mt_twist(o[tid + 227], o[tid + 228], w[tid]);                             // [227, 454): new words of the first phase
            __syncthreads();
            if (tid < kMtN - 1 - 454) w[tid + 454] = mt_twist(o[tid + 454], o[tid + 455], w[tid + 227]);                   // [454, 623)
            __syncthreads();
            if (tid == 0) w[kMtN - 1] = mt_twist(o[kMtN - 1], w[0], w[kMtM - 1]);
            __syncthreads();
            if (tid < kMtN) {
                const int i = base + tid;
                if (i < n - 1) z[i] = mt_temper(w[tid]) % (unsigned)(n - i);
            }
            cur ^= 1;
        }
        // ---- the swap counter
        if (tid == 0) inv[n - 1] = (unsigned)(n - 1);           // (the one position that is never a swap's own)
        if (tid == 0) next_p = 0u;
        __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0): the partners and the identity are written before anybody reads them
        __syncthreads();
        // ---- the swaps i = n - 2 - p, p = 0 .. n - 2 (p: the swap's index in running order = its priority)
        unsigned pr[kPermInFlight];                              // 0xffffffff: the slot is empty
        int pi[kPermInFlight], pj[kPermInFlight];
#pragma unroll
        for (int k = 0; k < kPermInFlight; ++k) pr[k] = 0xffffffffu;
        const unsigned total = (unsigned)(n - 1);
        // a slot takes the next swap in running order (the pending set stays closed under "older than a pending one")
        auto admit = [&](int k) {
            const unsigned p = atomicAdd(&next_p, 1u);
            if (p < total) {
                pr[k] = p;
                pi[k] = n - 2 - (int)p;
                pj[k] = pi[k] + (int)z[pi[k]];
            }
        };
#pragma unroll
        for (int k = 0; k < kPermInFlight; ++k) admit(k);
        unsigned spins = 0;
        bool failed = false;
        for (;;) {
            if (++spins > 4u * total + 1024u) {                   // (cannot happen: the oldest pending swap is carried out in every round)
                if (tid == 0) gave_up[blockIdx.x] = 0xdeadu;
                failed = true;                                    // (uniform: every lane counts the same rounds)
                break;
            }
            unsigned mine = 0;
#pragma unroll
            for (int k = 0; k < kPermInFlight; ++k) mine += pr[k] != 0xffffffffu;
            // (this barrier also ends the round before: its swaps are done -- the vmcnt wait below -- before anybody reserves again)
            if (!__syncthreads_or((int)mine)) break;              // nobody holds a swap and none is left to admit
            if (round == 0u) {                                   // the round counter wrapped: wipe the reservations (a shuffle of 2^20 rows takes ~400 rounds; the counter runs on from shuffle to shuffle)
                for (int f = tid; f < kResWords; f += kPermBlock) res[f] = 0xffffffffu;
                round = kRounds - 1;
                __syncthreads();
            }
            const unsigned tag = round << kPrioBits;
            --round;
#pragma unroll
            for (int k = 0; k < kPermInFlight; ++k)
                if (pr[k] != 0xffffffffu) {
                    atomicMin(&res[pi[k] & (kResWords - 1)], tag | pr[k]);
                    atomicMin(&res[pj[k] & (kResWords - 1)], tag | pr[k]);
                }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < kPermInFlight; ++k)
                if (pr[k] != 0xffffffffu) {
                    const unsigned v = tag | pr[k];
                    const unsigned got_i = res[pi[k] & (kResWords - 1)], got_j = res[pj[k] & (kResWords - 1)];
                    if (got_i == v && got_j == v) {
                        // Position i is touched for the first time by swap i itself (every older swap lies to the right of it): it still holds i, and
                        // nobody has to have written that.  So: one load, two stores, and no pass that writes the identity first.
                        // (plain accesses: workgroup-scope acquire is a no-op in non-tgsplit mode -- see MEMORY MODEL above)
                        if (pj[k] != pi[k]) {
                            const unsigned b = inv[pj[k]];
                            inv[pi[k]] = b;
                            inv[pj[k]] = (unsigned)pi[k];
                        } else {
                            inv[pi[k]] = (unsigned)pi[k];
                        }
                        pr[k] = 0xffffffffu;
                        admit(k);                                // the slot's next swap: its partner is loaded beside this round's memory traffic
                    }
                }
            __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0) = the workgroup-scope RELEASE of this round's stores (the next barrier publishes them)
        }
        // ---- the tags: file row f trains at position inv[f] of the epoch
        // (a shuffle that was given up leaves tags that match NO batch: nothing trains on a wrong permutation, and the host finds the
        // flag when it reads the job's results -- ultrare_amd.engine.TrainJob.check_tags)
        uint16_t *out = perms[perm].tags;
        for (int f = tid; f < n; f += kPermBlock) out[f] = failed ? (uint16_t)0xFFFFu : (uint16_t)(inv[f] / (unsigned)batch);
        __syncthreads();
    }
}

}  // namespace
}  // namespace ure

extern "C" int64_t ure_device_randperm_tags_scratch(int64_t n_max, int32_t groups)
{
    if (n_max <= 0 || groups <= 0) return 0;
    return 2 * ((n_max + 63) / 64 * 64) * (int64_t)groups + ((int64_t)groups + 63) / 64 * 64 + 64;        // (+ a word per group: its "gave up" flag)
}

extern "C" int ure_device_randperm_tags(const ure_perm_t *perms, int32_t n_perms, int64_t n_max, uint32_t *scratch, int64_t scratch_words, int32_t groups,
                                        void *stream)
{
    URE_ARG(n_perms >= 0 && n_max >= 0 && groups > 0);
    if (n_perms == 0 || n_max == 0) return 0;
    URE_ARG(perms && scratch);
    if (n_max > (int64_t)(1 << ure::kPrioBits)) return ure::fail(-1, "ure_device_randperm_tags: %lld rows: more than 2^20 (such shards keep the host path)", (long long)n_max);
    const int g = std::min<int>(groups, n_perms);
    if (scratch_words < ure_device_randperm_tags_scratch(n_max, groups)) return ure::fail(-1, "ure_device_randperm_tags: scratch too small");
    const int64_t per_group = 2 * ((n_max + 63) / 64 * 64);
    // (the flags sit behind the scratch of `groups` workgroups, whatever this launch uses of them; they are never cleared here: a caller
    // that reads them clears them when it makes the scratch)
    hipLaunchKernelGGL(ure::perm_tags_kernel, dim3((unsigned)g), dim3(ure::kPermBlock), 0, static_cast<hipStream_t>(stream), perms, (int)n_perms, scratch, per_group,
                       scratch + (size_t)groups * per_group);
    URE_HIP(hipGetLastError());
    return 0;
}
