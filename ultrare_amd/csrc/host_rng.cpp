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
#include <atomic>
#include <cstdint>
#include <thread>
#include <vector>

#include "ultrare_hip.h"

namespace ure {
int fail(int code, const char *fmt, ...);
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

void one_perm(uint64_t seed, int64_t n, int32_t *r)
{
    for (int64_t i = 0; i < n; ++i) r[i] = (int32_t)i;
    Mt19937 mt(seed);
    for (int64_t i = 0; i < n - 1; ++i) {
        const int64_t z = (int64_t)(mt.next() % (uint32_t)(n - i));
        const int32_t sav = r[i];
        r[i] = r[i + z];
        r[i + z] = sav;
    }
}

}  // namespace

extern "C" int ure_host_randperm(const int64_t *seeds, int n_perms, int64_t n, int32_t *out, int n_threads)
{
    if (!seeds || !out || n_perms < 0 || n < 0) return ure::fail(-1, "ure_host_randperm: bad arguments");
    if (n >= (int64_t)(0xffffffffu / 20)) return ure::fail(-1, "ure_host_randperm: n=%lld uses ATen's large-n branch, not restated", (long long)n);
    if (n_perms == 0 || n == 0) return 0;
    int nt = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
    nt = nt < 1 ? 1 : (nt > n_perms ? n_perms : nt);
    std::atomic<int> next{0};
    auto work = [&]() {
        for (int t = next.fetch_add(1); t < n_perms; t = next.fetch_add(1)) one_perm((uint64_t)seeds[t], n, out + (size_t)t * n);
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
