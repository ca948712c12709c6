// host_rng.cpp -- the epoch permutations of the reference's DataLoader, on host threads.
//
// read.py:133 builds DataLoader(shuffle=True): every epoch torch's RandomSampler draws a
// seed from the global CPU generator, seeds a fresh generator with it and calls
// torch.randperm(N) -- an MT19937-driven Fisher-Yates shuffle
//     r = [0..n);  for i in [0, n-1): z = mt() % (n - i); swap(r[i], r[i + z])
// (ATen randperm_cpu, the n < 2^32/20 branch; the engine is seeded with the low 32 bits
// of the seed).  Results must match the reference draw for draw, so this is a restatement of
// that published algorithm (checked against torch.randperm in tests/test_cpu_host.py), run
// for many epochs at once on a thread pool: each epoch has its own generator, so the
// permutations are independent.  torch.randperm itself costs ~2 ms per 180 k-row epoch on
// one thread, which would dominate a 50-epoch SISA job whose device time is ~10 ms.
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <string>
#include <thread>
#include <vector>

#include "normal_math.h"
#include "ultrare_hip.h"

namespace ure {
int fail(int code, const char *fmt, ...);
int host_threads();
int mt_jump_blocks(uint32_t *st, int64_t blocks);      // mt_jump.cpp
}

namespace {

struct Mt19937 {
    static constexpr int N = 624, M = 397;
    uint32_t st[N];
    int idx = N;
    explicit Mt19937(uint64_t seed)
    {
        st[0] = (uint32_t)(seed & 0xffffffffu);
        for (int j = 1; j < N; ++j) st[j] = 1812433253u * (st[j - 1] ^ (st[j - 1] >> 30)) + (uint32_t)j;
    }
    void refill()
    {
        for (int k = 0; k < N; ++k) {
            const uint32_t y = (st[k] & 0x80000000u) | (st[(k + 1) % N] & 0x7fffffffu);
            st[k] = st[(k + M) % N] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        idx = 0;
    }
    inline uint32_t next()
    {
        if (idx >= N) refill();
        uint32_t y = st[idx++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
};

// One permutation in three passes instead of one dependent loop.  The swap partner of position i,
//     z_i = mt() % (n - i),
// depends on the generator and on i only -- not on the swaps before it -- so all z_i are computed first:
// the MT19937 outputs in bulk (regeneration and tempering vectorise), then the remainders through a
// double-precision quotient (exact: x < 2^32, so x / m rounded to 53 bits never reaches the next integer
// from below nor falls under floor(x / m); see the proof sketch in DESIGN.md 5), which vectorises too,
// where the 32-bit `div` of the direct form costs ~25 cycles per element.  What is left of the loop-carried
// work is the swap itself, with addresses known ahead of time.  The generated sequence is ATen's, draw for draw.
#if defined(__HIP_DEVICE_COMPILE__) || !defined(__x86_64__)
#define URE_HOST_CLONES
#else
#define URE_HOST_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))      // the library is built once, run on any host
#endif
URE_HOST_CLONES void fill_partners(uint32_t *st, int *idx, uint32_t *z, int64_t count, int64_t n)
{
    constexpr int N = Mt19937::N, M = Mt19937::M;
    int64_t done = 0;
    while (done < count) {
        if (*idx >= N) {
            int k = 0;
            for (; k < N - M; ++k) {
                const uint32_t y = (st[k] & 0x80000000u) | (st[k + 1] & 0x7fffffffu);
                st[k] = st[k + M] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            for (; k < N - 1; ++k) {
                const uint32_t y = (st[k] & 0x80000000u) | (st[k + 1] & 0x7fffffffu);
                st[k] = st[k + M - N] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            const uint32_t y = (st[N - 1] & 0x80000000u) | (st[0] & 0x7fffffffu);
            st[N - 1] = st[M - 1] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            *idx = 0;
        }
        const int64_t take = std::min<int64_t>(N - *idx, count - done);
        const uint32_t *src = st + *idx;
        uint32_t *dst = z + done;
        for (int64_t t = 0; t < take; ++t) {
            uint32_t x = src[t];
            x ^= x >> 11;
            x ^= (x << 7) & 0x9d2c5680u;
            x ^= (x << 15) & 0xefc60000u;
            x ^= x >> 18;
            const uint32_t m = (uint32_t)(n - (done + t));
            const uint32_t q = (uint32_t)((double)x / (double)m);
            dst[t] = x - q * m;
        }
        *idx += (int)take;
        done += take;
    }
}

void one_perm(uint64_t seed, int64_t n, int32_t *r, std::vector<uint32_t> &z)
{
    for (int64_t i = 0; i < n; ++i) r[i] = (int32_t)i;
    if (n < 2) return;
    Mt19937 mt(seed);
    z.resize((size_t)n);
    fill_partners(mt.st, &mt.idx, z.data(), n - 1, n);
    for (int64_t i = 0; i < n - 1; ++i) {
        const int64_t j = i + (int64_t)z[(size_t)i];
        const int32_t sav = r[i];
        r[i] = r[j];
        r[j] = sav;
    }
}

// The INVERSE of one_perm's permutation (inv[r[b]] = b), without building r: the shuffle is the product of the transpositions
// (i, i + z[i]), i = 0 .. n-2, applied to the identity from the left end; its inverse is the same transpositions applied in the
// opposite order.  The partners z are all drawn before the first swap, so the chain can run from i = n-2 down to 0.  A batch tag
// is inv / batch: one sequential pass instead of a second scattered one (out[r[b]] = b / batch missed the L1 on every store).
URE_HOST_CLONES static void tags_of_positions(const int32_t *inv, int64_t n, int32_t batch, uint16_t *out)
{
    // floor(v / batch) through the rounded reciprocal: v < 2^31 and |v * RN(1 / batch) - v / batch| < 2^-21, so the truncated
    // product is the quotient or one beside it; the remainder says which
    const double rb = 1.0 / (double)batch;
    for (int64_t i = 0; i < n; ++i) {
        const uint32_t v = (uint32_t)inv[i];
        uint32_t q = (uint32_t)((double)v * rb);
        const int32_t r = (int32_t)(v - q * (uint32_t)batch);
        q += (uint32_t)(r >= batch) - (uint32_t)(r < 0);
        out[i] = (uint16_t)q;
    }
}

void one_perm_tags(uint64_t seed, int64_t n, int32_t batch, int32_t *inv, std::vector<uint32_t> &z, uint16_t *out)
{
    for (int64_t i = 0; i < n; ++i) inv[i] = (int32_t)i;
    if (n >= 2) {
        Mt19937 mt(seed);
        z.resize((size_t)n);
        fill_partners(mt.st, &mt.idx, z.data(), n - 1, n);
        // (two chains interleaved in one loop were tried: no faster -- the chain is not latency bound)
        for (int64_t i = n - 2; i >= 0; --i) {
            const int64_t j = i + (int64_t)z[(size_t)i];
            const int32_t sav = inv[i];
            inv[i] = inv[j];
            inv[j] = sav;
        }
    }
    tags_of_positions(inv, n, batch, out);
}

}  // namespace

extern "C" int ure_host_randperm(const int64_t *seeds, int n_perms, int64_t n, int32_t *out, int n_threads)
{
    if (!seeds || !out || n_perms < 0 || n < 0) return ure::fail(-1, "ure_host_randperm: bad arguments");
    if (n >= (int64_t)(0xffffffffu / 20)) return ure::fail(-1, "ure_host_randperm: n=%lld uses ATen's large-n branch, not restated", (long long)n);
    if (n_perms == 0 || n == 0) return 0;
    int nt = n_threads > 0 ? n_threads : ure::host_threads();
    nt = nt < 1 ? 1 : (nt > n_perms ? n_perms : nt);
    std::atomic<int> next{0};
    auto work = [&]() {
        std::vector<uint32_t> z;
        for (int t = next.fetch_add(1); t < n_perms; t = next.fetch_add(1)) one_perm((uint64_t)seeds[t], n, out + (size_t)t * n, z);
    };
    if (nt == 1) {
        work();
        return 0;
    }
    std::vector<std::thread> pool;
    pool.reserve(nt);
    for (int t = 0; t < nt; ++t) pool.emplace_back(work);
    for (auto &th : pool) th.join();
    return 0;
}

extern "C" int ure_host_randperm_tags(const int64_t *seeds, int n_perms, int64_t n, int32_t batch, uint16_t *tags, int n_threads)
{
    if (!seeds || !tags || n_perms < 0 || n < 0 || batch <= 0) return ure::fail(-1, "ure_host_randperm_tags: bad arguments");
    if (n >= (int64_t)(0xffffffffu / 20)) return ure::fail(-1, "ure_host_randperm_tags: n=%lld uses ATen's large-n branch, not restated", (long long)n);
    if ((n + batch - 1) / batch > 65535) return ure::fail(-1, "ure_host_randperm_tags: more than 65535 steps per epoch");
    if (n_perms == 0 || n == 0) return 0;
    int nt = n_threads > 0 ? n_threads : ure::host_threads();
    nt = nt < 1 ? 1 : (nt > n_perms ? n_perms : nt);
    std::atomic<int> next{0};
    auto work = [&]() {
        std::vector<uint32_t> z;
        std::vector<int32_t> r((size_t)n);
        // file row f trains at position inv[f] of the epoch: its step is inv[f] / batch
        for (int t = next.fetch_add(1); t < n_perms; t = next.fetch_add(1))
            one_perm_tags((uint64_t)seeds[t], n, batch, r.data(), z, tags + (size_t)t * n);
    };
    if (nt == 1) {
        work();
        return 0;
    }
    std::vector<std::thread> pool;
    pool.reserve(nt);
    for (int t = 0; t < nt; ++t) pool.emplace_back(work);
    for (auto &th : pool) th.join();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Skip-ahead of torch's CPU generator.  The reference draws four N(0,1) fills per model
// (utils.py:31-40) and throws two of them away (the nn.Embedding constructors' own fills, overwritten
// by init_weight), and in a multi-rank run every rank replays the whole stream to reach its own
// shards.  The draws are data independent, so the generator can be moved past them
// without computing a single normal: only the MT19937 state has to advance by the number of 32-bit
// outputs those calls would have consumed.  `state` is torch.get_rng_state()'s byte layout
// (CPUGeneratorImplStateLegacy: u64 seed, i32 left, i32 seeded, u64 next, u64 state[624], ...);
// ATen's engine draws with `if (--left == 0) next_state(); y = state[next++]`.
// ---------------------------------------------------------------------------------------------
// `blocks` consecutive next_state() regenerations of an MT19937 state, each in three runs without index wrap-around.
URE_HOST_CLONES static void mt_regenerate(uint32_t *st, int64_t blocks)
{
    constexpr int N = 624, M = 397;
    for (int64_t b = 0; b < blocks; ++b) {
        int k = 0;
        for (; k < N - M; ++k) {
            const uint32_t y = (st[k] & 0x80000000u) | (st[k + 1] & 0x7fffffffu);
            st[k] = st[k + M] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        for (; k < N - 1; ++k) {
            const uint32_t y = (st[k] & 0x80000000u) | (st[k + 1] & 0x7fffffffu);
            st[k] = st[k + M - N] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        const uint32_t y = (st[N - 1] & 0x80000000u) | (st[0] & 0x7fffffffu);
        st[N - 1] = st[M - 1] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
}

constexpr int64_t kJumpMinBlocks = 4096;      // 2.6 M outputs: ~0.8 ms of walking against ~0.1 ms (+ ~1 ms for a distance not seen before)

extern "C" int ure_host_mt_advance(uint8_t *state, int64_t n_bytes, int64_t n_draws)
{
    constexpr int N = 624;
    if (!state || n_bytes < (int64_t)(24 + 8 * N) || n_draws < 0) return ure::fail(-1, "ure_host_mt_advance: bad arguments");
    int32_t left;
    uint64_t next;
    __builtin_memcpy(&left, state + 8, 4);
    __builtin_memcpy(&next, state + 16, 8);
    uint64_t *wide = reinterpret_cast<uint64_t *>(state + 24);
    if (left < 1 || left > N || next > (uint64_t)N) return ure::fail(-1, "ure_host_mt_advance: not a torch CPU generator state (left=%d next=%llu)", left, (unsigned long long)next);
    uint32_t st[N];
    bool loaded = false;
    int64_t n = n_draws;
    while (n > 0) {
        if (left > 1) {                        // outputs left in the current block
            const int64_t c = n < (int64_t)left - 1 ? n : (int64_t)left - 1;
            left -= (int32_t)c;
            next += (uint64_t)c;
            n -= c;
            continue;
        }
        if (!loaded) {
            for (int k = 0; k < N; ++k) st[k] = (uint32_t)wide[k];
            loaded = true;
        }
        // whole blocks that are skipped entirely are regenerated back to back (AVX2 / AVX-512 clones of the loops: the skip-ahead of
        // a request's shards runs on the calling thread before any draw can start -- 1.0 ms of its critical path at 5 shards)
        const int64_t blocks = 1 + (n - 1) / N;                       // regenerations until fewer than N draws remain to take
        // Long distances are JUMPED (mt_jump.cpp: x^J mod the generator's characteristic polynomial, ~0.1 ms whatever J) in
        // multiples of 64 blocks -- so that shards whose distances differ by a block share the memoised polynomial -- and the
        // rest is walked: 56.8 M outputs per shard at BASELINE.json configs[3]'s shape took 19 ms each, on every rank.
        const int64_t jumped = blocks >= kJumpMinBlocks ? blocks & ~(int64_t)63 : 0;
        if (jumped)
            if (const int r = ure::mt_jump_blocks(st, jumped)) return r;
        mt_regenerate(st, blocks - jumped);
        n -= 1 + (blocks - 1) * (int64_t)N;                            // the last block's state[0] taken; all draws of the blocks before it
        left = N;
        next = 1;
    }
    if (loaded)
        for (int k = 0; k < N; ++k) wide[k] = st[k];
    __builtin_memcpy(state + 8, &left, 4);
    __builtin_memcpy(state + 16, &next, 8);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// MF.init_weight's two kept fills (utils.py:31-40) natively.  ATen's `tensor.normal_()` for a contiguous float32 tensor of n >= 16
// elements (aten/src/ATen/native/cpu/DistributionTemplates.h: normal_fill_AVX2) draws n uniforms (24 bits of one 32-bit output each)
// into the tensor with the scalar engine -- 3 of its 4.3 ns per element --, then turns them into normals 16 at a time, and when 16 does
// not divide n draws 16 more uniforms for the LAST 16 elements.  Here the uniforms come off the generator in bulk (regeneration and
// tempering vectorise: ~0.3 ns each), and the 16-blocks -- independent of each other -- go through the installed PyTorch's own
// polynomial kernels (host_normal_avx2.cpp) on n_threads threads.  ultrare_amd/rng.py compares the result with torch once per
// process and keeps torch's own fill when a single bit differs.
// ---------------------------------------------------------------------------------------------
extern "C" int ure_host_normal_blocks(float *data, int64_t n_blocks, float mean, float std_);       // host_normal_avx2.cpp

namespace {

URE_HOST_CLONES void temper_to_uniform(const uint32_t *src, float *dst, int64_t count)
{
    for (int64_t t = 0; t < count; ++t) {
        uint32_t x = src[t];
        x ^= x >> 11;
        x ^= (x << 7) & 0x9d2c5680u;
        x ^= (x << 15) & 0xefc60000u;
        x ^= x >> 18;
        dst[t] = (float)(x & 0xffffffu) * 5.9604644775390625e-8f;          // at::uniform_real_distribution<float>(0, 1): exact
    }
}

// `count` uniforms off a generator in ATen's engine order: `if (--left == 0) next_state(); y = state[next++]`
void mt_uniforms(uint32_t *st, int32_t &left, uint64_t &next, float *dst, int64_t count)
{
    constexpr int N = 624;
    while (count > 0) {
        if (left > 1) {
            const int64_t c = std::min<int64_t>(count, (int64_t)left - 1);
            temper_to_uniform(st + next, dst, c);
            left -= (int32_t)c;
            next += (uint64_t)c;
            dst += c;
            count -= c;
            continue;
        }
        mt_regenerate(st, 1);
        const int64_t c = std::min<int64_t>(count, N);
        temper_to_uniform(st, dst, c);
        left = (int32_t)(N - (c - 1));
        next = (uint64_t)c;
        dst += c;
        count -= c;
    }
}

int normal_fill(uint32_t *st, int32_t &left, uint64_t &next, float *out, int64_t n, int n_threads)
{
    mt_uniforms(st, left, next, out, n);
    const int64_t blocks = n / 16;
    const int nt = (int)std::min<int64_t>(std::max(1, n_threads), std::max<int64_t>(1, blocks / 512));
    std::atomic<int> rc{0};
    auto work = [&](int t) {
        const int64_t b0 = blocks * t / nt, b1 = blocks * (t + 1) / nt;
        if (const int r = ure_host_normal_blocks(out + 16 * b0, b1 - b0, 0.0f, 1.0f)) rc.store(r);
    };
    if (nt > 1) {
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; ++t) pool.emplace_back(work, t);
        work(0);
        for (auto &th : pool) th.join();
    } else {
        work(0);
    }
    if (rc.load()) return rc.load();
    if (n % 16) {
        mt_uniforms(st, left, next, out + n - 16, 16);
        return ure_host_normal_blocks(out + n - 16, 1, 0.0f, 1.0f);
    }
    return 0;
}

}  // namespace

// The Box-Muller half through the SCALAR restatement the device kernels use (normal_math.h), so that a host without a GPU can hold
// it against `tensor.normal_()`: data [16 n_blocks] uniforms -> normals in place.  variant 0 is the arithmetic of record; 1-3 are the
// other readings of the two ambiguous mul + add pairs (tests show that they are NOT torch's).
template <int kVariant>
static void scalar_blocks(float *data, int64_t n_blocks)
{
    for (int64_t b = 0; b < n_blocks; ++b)
        for (int j = 0; j < 8; ++j) {
            float *d = data + 16 * b + j;
            ure::nm_box_muller<kVariant>(d[0], d[8], d, d + 8);
        }
}

extern "C" int ure_host_normal_blocks_scalar(float *data, int64_t n_blocks, int32_t variant)
{
    if ((!data && n_blocks) || n_blocks < 0 || variant < 0 || variant > 3) return ure::fail(-1, "ure_host_normal_blocks_scalar: bad arguments");
    switch (variant) {
    case 0: scalar_blocks<0>(data, n_blocks); break;
    case 1: scalar_blocks<1>(data, n_blocks); break;
    case 2: scalar_blocks<2>(data, n_blocks); break;
    default: scalar_blocks<3>(data, n_blocks); break;
    }
    return 0;
}

// The per-epoch seeds of scratch.py:78-97 without a generator object: `n` int64 values as `tensor.random_()` draws them (two 32-bit
// outputs each, the first the high word, bit 63 cleared) from a COPY of the state moved past `skip_draws` outputs.
extern "C" int ure_host_draw_int64(const uint8_t *state, int64_t n_bytes, int64_t skip_draws, int64_t n, int64_t *out)
{
    constexpr int N = 624;
    if (!state || n_bytes < (int64_t)(24 + 8 * N) || skip_draws < 0 || n < 0 || (n && !out)) return ure::fail(-1, "ure_host_draw_int64: bad arguments");
    std::vector<uint8_t> copy(state, state + n_bytes);
    if (skip_draws)
        if (const int r = ure_host_mt_advance(copy.data(), n_bytes, skip_draws)) return r;
    int32_t left;
    uint64_t next;
    __builtin_memcpy(&left, copy.data() + 8, 4);
    __builtin_memcpy(&next, copy.data() + 16, 8);
    const uint64_t *wide = reinterpret_cast<const uint64_t *>(copy.data() + 24);
    if (left < 1 || left > N || next > (uint64_t)N) return ure::fail(-1, "ure_host_draw_int64: not a torch CPU generator state (left=%d next=%llu)", left, (unsigned long long)next);
    uint32_t st[N];
    for (int k = 0; k < N; ++k) st[k] = (uint32_t)wide[k];
    auto draw = [&]() {
        if (--left == 0) {
            mt_regenerate(st, 1);
            left = N;
            next = 0;
        }
        uint32_t x = st[next++];
        x ^= x >> 11;
        x ^= (x << 7) & 0x9d2c5680u;
        x ^= (x << 15) & 0xefc60000u;
        x ^= x >> 18;
        return x;
    };
    for (int64_t k = 0; k < n; ++k) {
        const uint64_t hi = draw(), lo = draw();
        out[k] = (int64_t)(((hi << 32) | lo) & 0x7fffffffffffffffull);
    }
    return 0;
}

extern "C" int ure_host_mf_init(uint8_t *state, int64_t n_bytes, int64_t skip_draws, float *U0, int64_t nu, float *V0, int64_t nv, int n_threads)
{
    constexpr int N = 624;
    if (!state || n_bytes < (int64_t)(24 + 8 * N) || skip_draws < 0 || nu < 0 || nv < 0 || (nu && !U0) || (nv && !V0))
        return ure::fail(-1, "ure_host_mf_init: bad arguments");
    if ((nu && nu < 16) || (nv && nv < 16)) return ure::fail(-1, "ure_host_mf_init: a fill of fewer than 16 elements takes ATen's scalar path, not restated");
    if (skip_draws)
        if (const int r = ure_host_mt_advance(state, n_bytes, skip_draws)) return r;
    int32_t left;
    uint64_t next;
    __builtin_memcpy(&left, state + 8, 4);
    __builtin_memcpy(&next, state + 16, 8);
    uint64_t *wide = reinterpret_cast<uint64_t *>(state + 24);
    if (left < 1 || left > N || next > (uint64_t)N) return ure::fail(-1, "ure_host_mf_init: not a torch CPU generator state (left=%d next=%llu)", left, (unsigned long long)next);
    uint32_t st[N];
    for (int k = 0; k < N; ++k) st[k] = (uint32_t)wide[k];
    int rc = 0;
    if (nu) rc = normal_fill(st, left, next, U0, nu, n_threads);
    if (!rc && nv) rc = normal_fill(st, left, next, V0, nv, n_threads);
    if (rc) return ure::fail(rc, "ure_host_mf_init: the AVX2 kernels of the installed PyTorch are not available in this build or on this CPU");
    for (int k = 0; k < N; ++k) wide[k] = st[k];
    __builtin_memcpy(state + 8, &left, 4);
    __builtin_memcpy(state + 16, &next, 8);
    return 0;
}

// The model inits of all shards of a request in ONE call: shard s from its own generator state (in / out), on n_threads threads side by
// side (a request of 16 shards started 16 Python workers for this; each spent as long under the interpreter lock as in here).
extern "C" int ure_host_mf_init_batch(int32_t n_shards, uint8_t *const *states, int64_t n_bytes, const int64_t *skip_draws, float *const *U0, int64_t nu,
                                      float *const *V0, int64_t nv, int n_threads)
{
    if (n_shards < 0 || (n_shards > 0 && (!states || !skip_draws || !U0 || !V0))) return ure::fail(-1, "ure_host_mf_init_batch: bad arguments");
    if (n_shards == 0) return 0;
    int nt = n_threads > 0 ? n_threads : ure::host_threads();
    nt = std::max(1, std::min(nt, (int)n_shards));
    const int inner = std::max(1, (n_threads > 0 ? n_threads : ure::host_threads()) / (int)n_shards);      // threads to spare go into each fill
    std::atomic<int> next{0}, rc{0};
    std::vector<std::string> why((size_t)n_shards);
    auto work = [&]() {
        for (int s = next.fetch_add(1); s < n_shards; s = next.fetch_add(1)) {
            const int r = ure_host_mf_init(states[s], n_bytes, skip_draws[s], U0[s], nu, V0[s], nv, inner);
            if (r) { why[(size_t)s] = ure_last_error(); rc.store(r); }
        }
    };
    if (nt == 1) work();
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; ++t) pool.emplace_back(work);
        for (auto &th : pool) th.join();
    }
    if (rc.load())
        for (int s = 0; s < n_shards; ++s)
            if (!why[(size_t)s].empty()) return ure::fail(rc.load(), "ure_host_mf_init_batch: shard %d: %s", s, why[(size_t)s].c_str());
    return 0;
}

