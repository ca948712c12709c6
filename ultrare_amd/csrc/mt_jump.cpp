// mt_jump.cpp -- MT19937 jump-ahead: a generator moved past J outputs in the time of ~20,000 of them, whatever J.
//
// Why it is here.  The reference's randomness is torch's one global MT19937 stream: shard s of a Sisa.learn starts
// where shard s - 1 stopped, 4 model fills + 4 seeds per epoch further on (utils.py:31-40, scratch.py:78-97).  The draws are data
// independent, so a rank can START every shard at its own state -- but reaching those states by stepping the generator
// (ure_host_mt_advance, rounds 2-4) is a sequential walk of 56.8 M outputs per shard at BASELINE.json configs[3]'s shape: 0.2-0.6 s
// in front of every cold request of every rank, 3-8x the device work a rank of an 8-GPU run has.
//
// The generator is linear over GF(2): its 19,937-bit state moves by a fixed matrix T whose characteristic polynomial phi is
// primitive of degree 19,937, so T^J = g(T) for g = x^J mod phi (Haramoto, Matsumoto, Nishimura, Panneton, L'Ecuyer: "Efficient
// jump ahead for F2-linear random number generators", INFORMS J. Comput. 2008).  Two things make that cheap here:
//   * phi has 135 terms and its second-highest is x^19314, so a 39,872-bit square is reduced 64 bits at a time with 134 shifted
//     XORs per word: x^J mod phi for J ~ 6e7 costs ~26 squarings, well under a millisecond, memoised per distance (a request's
//     shards are all the same distance apart);
//   * g(T) applied to a state is NOT evaluated by Horner's rule (19,937 dependent generator steps with a conditional 624-word
//     XOR each): the raw state words x_0, x_1, ... the generator would produce from here are a linear recurring sequence with the
//     same polynomial, so word k of the jumped block is  XOR_{i : g_i = 1} x_{i + k + 1}  -- 33 blocks of plain regeneration and a
//     ~10,000-term XOR of 624-word windows, no dependence between terms: ~0.1 ms.
// (The +1: x_0's low 31 bits are not part of the state -- MT19937 keeps 623 words and one bit -- but every x_j, j >= 1, is a
// linear function of it; with E = 624 b - 1 the window sum over j = i + k + 1 >= 1 yields x_{624 b + k}, k = 0..623: block b.)
//
// phi itself is a constant of the generator.  The table below was computed with Berlekamp-Massey from 40,074 output bits
// (tests/test_cpu_host.py recomputes it the same way and compares; every jump is also checked there against the walk).
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "ultrare_hip.h"

namespace ure {
int fail(int code, const char *fmt, ...);
}

namespace {

constexpr int kDeg = 19937;
constexpr int kWords = 312;              // 64-bit words of a reduced polynomial (19,968 bits; the top 31 stay zero)
constexpr int kMtN = 624, kMtM = 397;
constexpr int kPhiTerms = 134;
// exponents of phi(x) below x^19937
const uint16_t kPhiLow[kPhiTerms] = {
    0, 1189, 1416, 1585, 1643, 1870, 2493, 2773, 3000, 3227, 3454, 3681, 3908, 4135,
    4362, 4753, 5661, 6337, 6569, 7129, 7477, 7525, 7583, 7752, 7979, 8206, 9505, 9901,
    9969, 10128, 10693, 10761, 10920, 11089, 11147, 11157, 11215, 11321, 11374, 11384, 11485, 11611,
    11712, 11717, 11838, 11881, 11944, 11997, 12277, 12335, 12393, 12504, 12509, 12620, 12673, 12731,
    12736, 12789, 12905, 12958, 12963, 13137, 13185, 13190, 13243, 13301, 13412, 13528, 13533, 13639,
    13697, 13760, 13813, 13866, 14093, 14151, 14209, 14320, 14325, 14436, 14547, 14552, 14605, 14721,
    14774, 14779, 14953, 15001, 15006, 15059, 15117, 15228, 15344, 15349, 15455, 15513, 15576, 15629,
    15682, 15909, 15967, 16025, 16136, 16141, 16252, 16363, 16368, 16421, 16537, 16590, 16595, 16817,
    16822, 16875, 16933, 17044, 17160, 17271, 17329, 17445, 17498, 17725, 17783, 17841, 17952, 18068,
    18179, 18237, 18406, 18633, 18691, 18860, 19087, 19314,
};

struct Poly {
    uint64_t w[kWords];
    std::vector<uint16_t> support;       // degrees of the nonzero terms, ascending
};

inline void xor_at(uint64_t *r, uint64_t v, int bitpos)
{
    const int wi = bitpos >> 6, sh = bitpos & 63;
    r[wi] ^= v << sh;
    if (sh) r[wi + 1] ^= v >> (64 - sh);
}

// r [2 kWords] (degree < 39,936) -> r mod phi in r [0, kWords).  Every term of phi but the leading one lies 623 or more degrees below
// it, so a whole 64-bit word of high coefficients folds down at once and never into itself.
void reduce(uint64_t *r)
{
    for (int idx = 2 * kWords - 1; idx >= kWords; --idx) {
        const uint64_t v = r[idx];
        if (!v) continue;
        r[idx] = 0;
        const int base = idx * 64 - kDeg;
        for (int j = 0; j < kPhiTerms; ++j) xor_at(r, v, base + kPhiLow[j]);
    }
    const uint64_t v = r[kWords - 1] >> 33;                 // degrees 19937 .. 19967 sit in bits 33 .. 63 of word 311
    if (v) {
        r[kWords - 1] &= (1ull << 33) - 1;
        for (int j = 0; j < kPhiTerms; ++j) xor_at(r, v, kPhiLow[j]);
    }
}

inline uint64_t spread32(uint32_t x)
{
    uint64_t v = x;
    v = (v | (v << 16)) & 0x0000ffff0000ffffull;
    v = (v | (v << 8)) & 0x00ff00ff00ff00ffull;
    v = (v | (v << 4)) & 0x0f0f0f0f0f0f0f0full;
    v = (v | (v << 2)) & 0x3333333333333333ull;
    v = (v | (v << 1)) & 0x5555555555555555ull;
    return v;
}

void square(uint64_t *p)                                    // p [kWords] <- p^2 mod phi  (squaring over GF(2) spreads the bits)
{
    uint64_t r[2 * kWords];
    for (int i = 0; i < kWords; ++i) {
        r[2 * i] = spread32((uint32_t)p[i]);
        r[2 * i + 1] = spread32((uint32_t)(p[i] >> 32));
    }
    reduce(r);
    std::memcpy(p, r, sizeof(uint64_t) * kWords);
}

void mul_x(uint64_t *p)                                     // p <- p x mod phi
{
    uint64_t carry = 0;
    for (int i = 0; i < kWords; ++i) {
        const uint64_t w = p[i];
        p[i] = (w << 1) | carry;
        carry = w >> 63;
    }
    if (p[kWords - 1] >> 33 & 1ull) {
        p[kWords - 1] &= (1ull << 33) - 1;
        for (int j = 0; j < kPhiTerms; ++j) p[kPhiLow[j] >> 6] ^= 1ull << (kPhiLow[j] & 63);
    }
}

void pow_x(int64_t e, uint64_t *p)                          // p <- x^e mod phi, e >= 0
{
    std::memset(p, 0, sizeof(uint64_t) * kWords);
    int top = 0;
    while (top < 62 && (e >> (top + 1))) ++top;
    int64_t mono = 0;                                       // the power so far is the monomial x^mono while that stays below the degree
    bool is_mono = true;
    for (int b = top; b >= 0; --b) {
        const int bit = (int)((e >> b) & 1);
        if (is_mono) {
            const int64_t next = 2 * mono + bit;
            if (next < kDeg) {
                mono = next;
                continue;
            }
            p[mono >> 6] = 1ull << (mono & 63);
            is_mono = false;
        }
        square(p);
        if (bit) mul_x(p);
    }
    if (is_mono) p[mono >> 6] = 1ull << (mono & 63);
}

void list_support(Poly &g)
{
    g.support.clear();
    for (int i = 0; i < kWords; ++i)
        for (uint64_t w = g.w[i]; w; w &= w - 1) g.support.push_back((uint16_t)(i * 64 + __builtin_ctzll(w)));
}

std::mutex g_memo_lock;
std::map<int64_t, std::shared_ptr<const Poly>> g_memo;      // blocks -> x^(624 blocks - 1) mod phi

// x^(624 blocks - 1) mod phi.  Memoised: a request's shards are all the same distance apart, and the next request's too.  A
// doubled distance (the sub-stream trees of csrc/mf_init.hip) costs one squaring of the half's polynomial: x^(2E + 1) = (x^E)^2 x.
std::shared_ptr<const Poly> poly_of_blocks(int64_t blocks)
{
    std::shared_ptr<const Poly> half;
    {
        std::lock_guard<std::mutex> hold(g_memo_lock);
        auto it = g_memo.find(blocks);
        if (it != g_memo.end()) return it->second;
        if (blocks % 2 == 0) {
            it = g_memo.find(blocks / 2);
            if (it != g_memo.end()) half = it->second;
        }
    }
    auto g = std::make_shared<Poly>();
    if (half) {
        std::memcpy(g->w, half->w, sizeof(g->w));
        square(g->w);
        mul_x(g->w);
    } else {
        pow_x(624 * blocks - 1, g->w);
    }
    list_support(*g);
    std::lock_guard<std::mutex> hold(g_memo_lock);
    if (g_memo.size() >= 256) g_memo.clear();
    g_memo[blocks] = g;
    return g;
}

#if defined(__HIP_DEVICE_COMPILE__) || !defined(__x86_64__)
#define URE_HOST_CLONES
#else
#define URE_HOST_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))
#endif

// acc [624] = XOR over the support of the 624-word windows x [s + 1 ..]: thirteen tiles of 48 words -- three 16-word vectors that
// stay in registers across the ~10,000 terms (written with vector types: the auto-vectoriser's cost model declined the plain loop).
typedef uint32_t v16u __attribute__((vector_size(64), aligned(4)));

URE_HOST_CLONES void window_sum(const uint32_t *x, const uint16_t *sup, int n_sup, uint32_t *out)
{
    constexpr int kTile = 48;
    static_assert(kMtN % kTile == 0, "tiles cover the block");
    for (int t0 = 0; t0 < kMtN; t0 += kTile) {
        v16u a0 = {}, a1 = {}, a2 = {};
        const uint32_t *base = x + 1 + t0;
        for (int s = 0; s < n_sup; ++s) {
            const v16u *p = reinterpret_cast<const v16u *>(base + sup[s]);
            a0 ^= p[0];
            a1 ^= p[1];
            a2 ^= p[2];
        }
        v16u *o = reinterpret_cast<v16u *>(out + t0);
        o[0] = a0;
        o[1] = a1;
        o[2] = a2;
    }
}

URE_HOST_CLONES void raw_words(uint32_t *x, int n_new)             // x [0, 624) given: x [624, 624 + n_new) by the recurrence
{
    for (int n = 0; n < n_new; ++n) {
        const uint32_t y = (x[n] & 0x80000000u) | (x[n + 1] & 0x7fffffffu);
        x[n + kMtN] = x[n + kMtM] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
}

}  // namespace

namespace ure {

// st [624] = the words of a generator block  ->  the block `blocks` regenerations later, in place.
int mt_jump_blocks(uint32_t *st, int64_t blocks)
{
    if (blocks <= 0) return 0;
    if (blocks > (int64_t)1 << 52) return ure::fail(-1, "mt_jump_blocks: distance out of range");
    const std::shared_ptr<const Poly> g = poly_of_blocks(blocks);
    constexpr int kNeed = kDeg + kMtN;                              // the highest word read: x [19936 + 1 + 623]
    std::vector<uint32_t> x((size_t)(kMtN + kNeed + 64));
    std::memcpy(x.data(), st, sizeof(uint32_t) * kMtN);
    raw_words(x.data(), kNeed);
    window_sum(x.data(), g->support.data(), (int)g->support.size(), st);
    return 0;
}

}  // namespace ure

extern "C" int ure_host_mt_jump_blocks(uint32_t *st, int64_t blocks)
{
    if (!st || blocks < 0) return ure::fail(-1, "ure_host_mt_jump_blocks: bad arguments");
    return ure::mt_jump_blocks(st, blocks);
}

extern "C" int ure_host_mt_jump_support(int64_t blocks, uint16_t *support, int32_t capacity, int32_t *n_support)
{
    if (blocks <= 0 || !n_support || (capacity > 0 && !support)) return ure::fail(-1, "ure_host_mt_jump_support: bad arguments");
    const std::shared_ptr<const Poly> g = poly_of_blocks(blocks);
    *n_support = (int32_t)g->support.size();
    if ((int32_t)g->support.size() > capacity) return capacity > 0 ? ure::fail(-1, "ure_host_mt_jump_support: %d terms, room for %d", *n_support, capacity) : 0;
    std::memcpy(support, g->support.data(), sizeof(uint16_t) * g->support.size());
    return 0;
}

extern "C" int ure_host_mt_charpoly(uint16_t *exponents, int32_t capacity)
{
    if (!exponents || capacity < kPhiTerms + 1) return ure::fail(-1, "ure_host_mt_charpoly: room for %d exponents needed", kPhiTerms + 1);
    for (int j = 0; j < kPhiTerms; ++j) exponents[j] = kPhiLow[j];
    exponents[kPhiTerms] = (uint16_t)kDeg;
    return kPhiTerms + 1;
}
