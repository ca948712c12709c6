// mf_init.hip -- MF.init_weight's two kept fills (utils.py:31-40) made ON THE DEVICE, bit for bit torch's CPU `tensor.normal_()`.
//
// The reference draws every model's start tables from torch's global CPU generator: n uniforms off MT19937 in stream order, then
// Box-Muller 16 at a time (ATen normal_fill_16_AVX2; restated per lane in normal_math.h).  Rounds 1-4 made them on the host and
// uploaded them -- 909 M normals and 3.6 GB over PCIe for BASELINE.json configs[3] (32 shards x (162,000 + 60,000) x 128), which is
// why that request was host-bound ten times over.  Here the host only positions generators (mt_jump.cpp: microseconds
// per shard) and the device does the rest:
//
//   * A shard's draws are cut into SEGMENTS of 1,024 generator blocks (639 k outputs).  Segment 0 starts at the shard's own state;
//     segment j's start block comes from a doubling tree of jumps on the device -- level t computes segments [2^t, 2^(t+1)) from
//     segments [0, 2^t) with the polynomial of 2^t x 1,024 blocks (one squaring of the level before, on the host):
//     mt_jump_kernel regenerates the 20,561 raw words a jump needs into a circular LDS buffer and XORs the ~10,000 windows of
//     the polynomial's support, a quarter of the support per workgroup, lane = word of the new block.
//   * mf_init_fill_kernel: one workgroup per (shard, segment) walks its blocks eight at a time -- MT19937's recurrence
//     x[p] = x[p - 227] ^ f(x[p - 624], x[p - 623]) gives 227 new words per barrier -- and every lane turns pairs of tempered outputs
//     into normals (nm_box_muller) straight into the tables.  A 16-block that straddles two segments belongs to the one its first
//     draw falls in (a segment generates one block more than it owns).
//   * The re-drawn tail of a fill whose length 16 does not divide (ATen draws 16 fresh uniforms for the LAST 16 elements) is a
//     piece of its own; the main piece leaves those 16 elements alone, so no element is written twice.
//
// No atomics on the data path of the fill; the jump's four partial sums per block meet by atomicXor (exact in any order).
#include <atomic>
#include <mutex>
#include <string>
#include <thread>

#include "normal_math.h"
#include "ure_internal.h"

extern "C" int ure_host_mt_advance(uint8_t *state, int64_t n_bytes, int64_t n_draws);
extern "C" int ure_host_mt_jump_support(int64_t blocks, uint16_t *support, int32_t capacity, int32_t *n_support);

namespace ure {
namespace {

constexpr int kMtN = 624, kMtM = 397, kMtLag = kMtN - kMtM;      // 227 new words depend on older words only
constexpr int kDeg = 19937;
constexpr int kSegBlocks = 1024;                // generator blocks per segment
constexpr int kChunkBlocks = 8;                 // ... walked eight at a time
constexpr int kFillBlock = 640;                 // ten wavefronts: lanes 0..622 own a word of a generation step, all of them turn outputs into normals
constexpr int kFillWide = 623;                  // words per dependent step
constexpr int kJumpLanes = 640;                 // ten wavefronts: lanes 0..623 own a word of the new block
constexpr int kJumpParts = 4;                   // workgroups per jump, each a quarter of the support's degree range
constexpr int kJumpPartSpan = (kDeg + kJumpParts - 1) / kJumpParts;
constexpr int kJumpRing = 8192;                 // circular LDS buffer of raw words (32 KB) >= kJumpPartSpan + 624 + 1
static_assert(kJumpPartSpan + kMtN + 1 <= kJumpRing, "a part's windows fit the ring");
constexpr int kMaxLevels = 24;

struct init_piece {
    float *dest;          // element e of the piece goes to dest[e] when e < limit
    int64_t draw_off;     // the piece's first draw, counted from the shard's first
    int64_t n16;          // 16-blocks
    int64_t limit;
};

struct init_shard {
    init_piece piece[4];  // U main, U tail, V main, V tail (n16 = 0: absent)
    int64_t n_out;        // outputs of the block sequence the shard needs: q0 + its draws
    int32_t q0;           // index of the shard's first draw in block 0
    int32_t pad;
};

__device__ __forceinline__ unsigned mt_twist(unsigned a, unsigned b, unsigned far)
{
    const unsigned y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

__device__ __forceinline__ unsigned mt_mix(unsigned a, unsigned b)     // (the twist's linear part)
{
    const unsigned y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

__device__ __forceinline__ unsigned mt_temper(unsigned x)
{
    x ^= x >> 11;
    x ^= (x << 7) & 0x9d2c5680u;
    x ^= (x << 15) & 0xefc60000u;
    x ^= x >> 18;
    return x;
}

// states [n_shards][J][624]; level with n_src source segments: workgroup (shard, i, part) adds part `part` of the jump of segment i
// into segment n_src + i (zeroed before the tree starts).  sup: the polynomial's support, ascending; part_at [kJumpParts + 1]: where the
// list crosses the multiples of kJumpPartSpan.
__global__ __launch_bounds__(kJumpLanes) void mt_jump_kernel(unsigned *__restrict__ states, int J, int n_src, int n_dst, const unsigned *__restrict__ sup,
                                                              const int *__restrict__ part_at)
{
    __shared__ unsigned ring[kJumpRing];
    constexpr unsigned M = kJumpRing - 1;
    const int tid = threadIdx.x;
    const int part = blockIdx.x % kJumpParts;
    const int i = (blockIdx.x / kJumpParts) % n_dst;
    const int shard = blockIdx.x / (kJumpParts * n_dst);
    const int s0 = part_at[part], s1 = part_at[part + 1];
    if (s0 == s1) return;
    const unsigned *src = states + ((size_t)shard * J + i) * kMtN;
    unsigned *dst = states + ((size_t)shard * J + n_src + i) * kMtN;
    if (tid < kMtN) ring[tid] = src[tid];
    __syncthreads();
    // raw words up to the last one a window of this part reads: x[last degree + 1 + 623]
    const int p_end = (int)sup[s1 - 1] + 1 + kMtN;
    for (int pos = kMtN; pos < p_end; pos += kMtLag) {
        const int p = pos + tid;
        if (tid < kMtLag && p < p_end) ring[p & M] = mt_twist(ring[(p - kMtN) & M], ring[(p - kMtN + 1) & M], ring[(p - kMtLag) & M]);
        __syncthreads();
    }
    if (tid < kMtN) {
        unsigned a0 = 0u, a1 = 0u, a2 = 0u, a3 = 0u;
        const unsigned base = 1u + (unsigned)tid;
        int s = s0;
        for (; s + 4 <= s1; s += 4) {
            a0 ^= ring[(sup[s] + base) & M];
            a1 ^= ring[(sup[s + 1] + base) & M];
            a2 ^= ring[(sup[s + 2] + base) & M];
            a3 ^= ring[(sup[s + 3] + base) & M];
        }
        for (; s < s1; ++s) a0 ^= ring[(sup[s] + base) & M];
        atomicXor(dst + tid, a0 ^ a1 ^ a2 ^ a3);
    }
}

// A barrier that orders LDS traffic only (as csrc/perm_chain.hip's): __syncthreads() also waits for the wave's global stores -- the tables'
// elements of the chunk before --, an L2 round trip per chunk that nothing here depends on.
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__global__ __launch_bounds__(kFillBlock) void mf_init_fill_kernel(const init_shard *__restrict__ shards, const unsigned *__restrict__ states, int J)
{
    __shared__ unsigned w[(kChunkBlocks + 1) * kMtN];
    const int tid = threadIdx.x;
    const int shard = blockIdx.x / J, seg = blockIdx.x % J;
    const init_shard *S = shards + shard;
    const int64_t n_out = S->n_out;
    const int64_t seg_lo = (int64_t)seg * kSegBlocks * kMtN;
    if (seg_lo >= n_out) return;
    const unsigned *st = states + ((size_t)shard * J + seg) * kMtN;
    for (int k = tid; k < kMtN; k += kFillBlock) w[k] = st[k];
    lds_barrier();
    const int64_t q0 = S->q0;
    for (int chunk = 0; chunk < kSegBlocks / kChunkBlocks; ++chunk) {
        const int64_t lo = seg_lo + (int64_t)chunk * kChunkBlocks * kMtN;
        if (lo >= n_out) break;
        const int64_t hi = lo + kChunkBlocks * kMtN;
        // the chunk's blocks 1 .. 8 behind block 0 (block 8 is the next chunk's block 0 and lends its first words to a straddling 16-block)
        // (the recurrence x[p] = x[p - 227] ^ F(p), F(p) = f(x[p - 624], x[p - 623]), substituted into itself twice: x[p] = x[p - 681] ^ F(p - 454) ^
        // F(p - 227) ^ F(p) -- 623 new words per barrier, the width x[p - 623] allows, instead of 227; the first 454 words behind block 0 have
        // no x[p - 681] and take the form with one / two terms.  As csrc/perm_chain.hip's words pass.)
        for (int g0 = 0; g0 < kChunkBlocks * kMtN; g0 += kFillWide) {
            const int g = g0 + tid;
            if (tid < kFillWide && g < kChunkBlocks * kMtN) {
                const int p = kMtN + g;
                const bool two = g >= kMtLag, three = g >= 2 * kMtLag;
                const unsigned a0 = w[p - kMtN], a1 = w[p - kMtN + 1];
                const unsigned b0 = two ? w[p - kMtLag - kMtN] : 0u, b1 = two ? w[p - kMtLag - kMtN + 1] : 0u;
                const unsigned c0 = three ? w[p - 2 * kMtLag - kMtN] : 0u, c1 = three ? w[p - 2 * kMtLag - kMtN + 1] : 0u;
                const unsigned x = w[p - (three ? 3 * kMtLag : two ? 2 * kMtLag : kMtLag)];
                w[p] = x ^ mt_mix(a0, a1) ^ (two ? mt_mix(b0, b1) : 0u) ^ (three ? mt_mix(c0, c1) : 0u);
            }
            lds_barrier();
        }
#pragma unroll 1
        for (int pi = 0; pi < 4; ++pi) {
            const int64_t n16 = S->piece[pi].n16;
            if (n16 == 0) continue;
            const int64_t g0 = q0 + S->piece[pi].draw_off;                 // the piece's first draw in the block sequence
            const int64_t t_lo = lo <= g0 ? 0 : (lo - g0 + 15) >> 4;        // 16-blocks whose first draw falls into [lo, hi)
            int64_t t_hi = hi <= g0 ? 0 : (hi - g0 + 15) >> 4;
            t_hi = t_hi < n16 ? t_hi : n16;
            if (t_lo >= t_hi) continue;
            float *dest = S->piece[pi].dest;
            const int64_t limit = S->piece[pi].limit;
            const int64_t base = g0 - lo;                                   // (negative when the piece starts before the chunk)
            for (int64_t i = t_lo * 8 + tid; i < t_hi * 8; i += kFillBlock) {
                const int64_t e = ((i >> 3) << 4) + (i & 7);
                const int loc = (int)(base + e);
                float a, b;
                nm_box_muller(nm_uniform(mt_temper(w[loc])), nm_uniform(mt_temper(w[loc + 8])), &a, &b);
                if (e < limit) dest[e] = a;
                if (e + 8 < limit) dest[e + 8] = b;
            }
        }
        lds_barrier();
        for (int k = tid; k < kMtN; k += kFillBlock) w[k] = w[kChunkBlocks * kMtN + k];
        lds_barrier();
    }
}

int64_t fill_draws_of(int64_t n) { return n ? n + ((n % 16) ? 16 : 0) : 0; }

int levels_of(int64_t J)
{
    int t = 0;
    while (((int64_t)1 << t) < J) ++t;
    return t;
}

int64_t max_segments(int64_t nu, int64_t nv)
{
    const int64_t blocks = (kMtN - 1 + fill_draws_of(nu) + fill_draws_of(nv) + kMtN - 1) / kMtN;
    return std::max<int64_t>(1, (blocks + kSegBlocks - 1) / kSegBlocks);
}

constexpr int64_t kDescWords = sizeof(init_shard) / 4;
constexpr int64_t kLevelWords = 8 + kDeg + 3;                          // part_at [5] (padded to 8) + the support

int64_t plan_words(int32_t n_shards, int64_t J) { return n_shards * kDescWords + levels_of(J) * kLevelWords + (int64_t)n_shards * kMtN; }

// Library-owned pinned staging for the plan's upload: one buffer, reused once the copy that read it is done.
struct Staging {
    std::mutex lock;
    void *ptr = nullptr;
    size_t cap = 0;
    hipEvent_t ev = nullptr;
    bool pending = false;
} g_stage;

void regenerate(uint32_t *st)
{
    uint32_t x[2 * kMtN];
    for (int k = 0; k < kMtN; ++k) x[k] = st[k];
    for (int n = 0; n < kMtN; ++n) {
        const uint32_t y = (x[n] & 0x80000000u) | (x[n + 1] & 0x7fffffffu);
        x[n + kMtN] = x[n + kMtM] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    for (int k = 0; k < kMtN; ++k) st[k] = x[kMtN + k];
}

}  // namespace
}  // namespace ure

extern "C" int64_t ure_device_mf_init_scratch(int32_t n_shards, int64_t nu, int64_t nv)
{
    if (n_shards <= 0 || nu < 0 || nv < 0) return 0;
    const int64_t J = ure::max_segments(nu, nv);
    return ure::plan_words(n_shards, J) + (int64_t)n_shards * J * ure::kMtN + 64;
}

extern "C" int ure_device_mf_init(int32_t n_shards, uint8_t *const *states, int64_t n_bytes, const int64_t *skip_draws, float *const *U0, int64_t nu,
                                  float *const *V0, int64_t nv, uint32_t *scratch, int64_t scratch_words, int n_threads, void *stream)
{
    using namespace ure;
    URE_ARG(n_shards >= 0 && nu >= 0 && nv >= 0);
    if (n_shards == 0 || nu + nv == 0) return 0;
    URE_ARG(states && skip_draws && U0 && V0 && scratch && n_bytes >= (int64_t)(24 + 8 * kMtN));
    if ((nu && nu < 16) || (nv && nv < 16)) return fail(-1, "ure_device_mf_init: a fill of fewer than 16 elements takes ATen's scalar path, not restated");
    if (scratch_words < ure_device_mf_init_scratch(n_shards, nu, nv)) return fail(-1, "ure_device_mf_init: scratch too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t du = fill_draws_of(nu), dv = fill_draws_of(nv);

    // ---- host: every shard's generator to its kept fills (block 0 + the first draw's index in it), and past them
    std::vector<init_shard> desc((size_t)n_shards);
    std::vector<uint32_t> seg0((size_t)n_shards * kMtN);
    std::vector<int> rc((size_t)n_shards, 0);
    std::vector<std::string> why((size_t)n_shards);
    auto prepare = [&](int s) {
        uint8_t *state = states[s];
        if (skip_draws[s])
            if ((rc[s] = ure_host_mt_advance(state, n_bytes, skip_draws[s]))) { why[s] = ure_last_error(); return; }
        int32_t left;
        uint64_t next;
        __builtin_memcpy(&left, state + 8, 4);
        __builtin_memcpy(&next, state + 16, 8);
        if (left < 1 || left > kMtN || next > (uint64_t)kMtN) { rc[s] = -1; why[s] = "not a torch CPU generator state"; return; }
        const uint64_t *wide = reinterpret_cast<const uint64_t *>(state + 24);
        uint32_t *b0 = seg0.data() + (size_t)s * kMtN;
        for (int k = 0; k < kMtN; ++k) b0[k] = (uint32_t)wide[k];
        int q0 = (int)next;
        if (left == 1) {                        // the next draw regenerates: block 0 is the regenerated one
            regenerate(b0);
            q0 = 0;
        }
        init_shard &D = desc[(size_t)s];
        __builtin_memset(&D, 0, sizeof(D));
        D.q0 = q0;
        D.n_out = q0 + du + dv;
        int64_t off = 0;
        int at = 0;
        for (int f = 0; f < 2; ++f) {
            float *dest = f ? V0[s] : U0[s];
            const int64_t n = f ? nv : nu;
            if (n) {
                D.piece[at++] = init_piece{dest, off, n / 16, (n % 16) ? n - 16 : n};
                if (n % 16) D.piece[at++] = init_piece{dest + (n - 16), off + n, 1, 16};
            }
            off += f ? dv : du;
        }
        if ((rc[s] = ure_host_mt_advance(state, n_bytes, du + dv))) why[s] = ure_last_error();
    };
    const int nt = std::max(1, std::min<int>(n_threads > 0 ? n_threads : host_threads(), n_shards));
    if (nt == 1 || du + dv < (4 << 20)) {
        for (int s = 0; s < n_shards; ++s) prepare(s);
    } else {
        prepare(0);                              // (the first computes the distances' polynomials; the others find them memoised)
        std::atomic<int> next_s{1};
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; ++t)
            pool.emplace_back([&]() {
                for (int s = next_s.fetch_add(1); s < n_shards; s = next_s.fetch_add(1)) prepare(s);
            });
        for (auto &th : pool) th.join();
    }
    for (int s = 0; s < n_shards; ++s)
        if (rc[s]) return fail(rc[s], "ure_device_mf_init: shard %d: %s", s, why[s].c_str());

    int64_t J = 1;
    for (int s = 0; s < n_shards; ++s) J = std::max<int64_t>(J, (desc[s].n_out + (int64_t)kSegBlocks * kMtN - 1) / ((int64_t)kSegBlocks * kMtN));
    URE_ARG(J <= max_segments(nu, nv));
    const int levels = levels_of(J);
    URE_ARG(levels <= kMaxLevels);
    const int64_t pw = plan_words(n_shards, J);

    // ---- the plan: descriptors | per level: part_at [8], support [19,940] | block 0 of every shard -- through pinned staging, one copy
    std::lock_guard<std::mutex> hold(g_stage.lock);
    if (g_stage.pending) {
        URE_HIP(hipEventSynchronize(g_stage.ev));
        g_stage.pending = false;
    }
    if (g_stage.cap < (size_t)pw * 4) {
        if (g_stage.ptr) URE_HIP(hipHostFree(g_stage.ptr));
        g_stage.ptr = nullptr;
        g_stage.cap = 0;
        URE_HIP(hipHostMalloc(&g_stage.ptr, (size_t)pw * 4, hipHostMallocDefault));
        g_stage.cap = (size_t)pw * 4;
    }
    if (!g_stage.ev) URE_HIP(hipEventCreateWithFlags(&g_stage.ev, hipEventDisableTiming));
    uint32_t *plan = static_cast<uint32_t *>(g_stage.ptr);
    __builtin_memcpy(plan, desc.data(), sizeof(init_shard) * (size_t)n_shards);
    uint32_t *lv = plan + n_shards * kDescWords;
    std::vector<uint16_t> sup16((size_t)kDeg);
    for (int t = 0; t < levels; ++t) {
        int32_t n_sup = 0;
        if (const int r = ure_host_mt_jump_support(((int64_t)kSegBlocks) << t, sup16.data(), kDeg, &n_sup)) return r;
        uint32_t *L = lv + (int64_t)t * kLevelWords;
        int32_t *part_at = reinterpret_cast<int32_t *>(L);
        int at = 0;
        for (int p = 0; p <= kJumpParts; ++p) {
            while (at < n_sup && (int)sup16[(size_t)at] < p * kJumpPartSpan) ++at;
            part_at[p] = p == kJumpParts ? n_sup : at;
        }
        for (int k = 0; k < n_sup; ++k) L[8 + k] = sup16[(size_t)k];
    }
    uint32_t *b0 = lv + (int64_t)levels * kLevelWords;
    __builtin_memcpy(b0, seg0.data(), sizeof(uint32_t) * seg0.size());

    uint32_t *dev_plan = scratch;
    uint32_t *dev_states = scratch + ((pw + 63) / 64) * 64;
    URE_HIP(hipMemcpyAsync(dev_plan, plan, (size_t)pw * 4, hipMemcpyHostToDevice, st));
    URE_HIP(hipEventRecord(g_stage.ev, st));
    g_stage.pending = true;
    if (J > 1) URE_HIP(hipMemsetAsync(dev_states, 0, (size_t)n_shards * J * kMtN * 4, st));
    URE_HIP(hipMemcpy2DAsync(dev_states, (size_t)J * kMtN * 4, dev_plan + (b0 - plan), (size_t)kMtN * 4, (size_t)kMtN * 4, (size_t)n_shards,
                             hipMemcpyDeviceToDevice, st));
    for (int t = 0; t < levels; ++t) {
        const int n_src = 1 << t;
        const int n_dst = (int)std::min<int64_t>(n_src, J - n_src);
        const uint32_t *L = dev_plan + n_shards * kDescWords + (int64_t)t * kLevelWords;
        hipLaunchKernelGGL(mt_jump_kernel, dim3((unsigned)(n_shards * n_dst * kJumpParts)), dim3(kJumpLanes), 0, st, dev_states, (int)J, n_src, n_dst, L + 8,
                           reinterpret_cast<const int *>(L));
    }
    hipLaunchKernelGGL(mf_init_fill_kernel, dim3((unsigned)(n_shards * J)), dim3(kFillBlock), 0, st, reinterpret_cast<const init_shard *>(dev_plan), dev_states, (int)J);
    URE_HIP(hipGetLastError());
    return 0;
}
