// mt_jump_dev.h -- MT19937 jump-ahead ON THE DEVICE by a doubling tree, for a stream that is cut into segments of 1,024 generator blocks:
// level t computes the start blocks of segments [2^t, 2^(t+1)) from those of segments [0, 2^t) with the polynomial of 2^t x 1,024 blocks
// (ure_host_mt_jump_support: x^J modulo the generator's characteristic polynomial, memoised per distance on the host).  The kernel is the
// one csrc/mf_init.hip runs for the model inits' fills (restated here so that csrc/perm_chain.hip can cut a shuffle of millions of rows
// into segments without touching that file: same arithmetic, same layout states [n][J][624]); the level tables of a device are uploaded
// once per process and kept.
#pragma once
#include <map>
#include <mutex>
#include <vector>

extern "C" int ure_host_mt_jump_support(int64_t blocks, uint16_t *support, int32_t capacity, int32_t *n_support);

namespace ure {
namespace jmp {

constexpr int kN = 624, kLag = 227;
constexpr int kDeg = 19937;
constexpr int kSegBlocks = 1024;                // generator blocks per segment
constexpr int64_t kSegWords = (int64_t)kSegBlocks * kN;
constexpr int kLanes = 640;                     // ten wavefronts: lanes 0..623 own a word of the new block
constexpr int kParts = 4;                       // workgroups per jump, each a quarter of the support's degree range
constexpr int kPartSpan = (kDeg + kParts - 1) / kParts;
constexpr int kRing = 8192;                     // circular LDS buffer of raw words (32 KB) >= kPartSpan + 624 + 1
static_assert(kPartSpan + kN + 1 <= kRing, "a part's windows fit the ring");
constexpr int kMaxLevels = 24;
constexpr int64_t kLevelWords = 8 + kDeg + 3;   // part_at [5] (padded to 8) + the support

__device__ __forceinline__ unsigned twist(unsigned a, unsigned b, unsigned far)
{
    const unsigned y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

// states [n][J][624]; level with n_src source segments: workgroup (stream, i, part) adds part `part` of the jump of segment i into
// segment n_src + i (zero before the tree starts).  sup: the polynomial's support, ascending; part_at [kParts + 1]: where the list
// crosses the multiples of kPartSpan.
__global__ __launch_bounds__(kLanes) void jump_kernel(unsigned *__restrict__ states, int J, int n_src, int n_dst, const unsigned *__restrict__ sup,
                                                      const int *__restrict__ part_at)
{
    __shared__ unsigned ring[kRing];
    constexpr unsigned M = kRing - 1;
    const int tid = threadIdx.x;
    const int part = blockIdx.x % kParts;
    const int i = (blockIdx.x / kParts) % n_dst;
    const int stream = blockIdx.x / (kParts * n_dst);
    const int s0 = part_at[part], s1 = part_at[part + 1];
    if (s0 == s1) return;
    const unsigned *src = states + ((size_t)stream * J + i) * kN;
    unsigned *dst = states + ((size_t)stream * J + n_src + i) * kN;
    if (tid < kN) ring[tid] = src[tid];
    __syncthreads();
    // raw words up to the last one a window of this part reads: x[last degree + 1 + 623]
    const int p_end = (int)sup[s1 - 1] + 1 + kN;
    for (int pos = kN; pos < p_end; pos += kLag) {
        const int p = pos + tid;
        if (tid < kLag && p < p_end) ring[p & M] = twist(ring[(p - kN) & M], ring[(p - kN + 1) & M], ring[(p - kLag) & M]);
        __syncthreads();
    }
    if (tid < kN) {
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

inline int levels_of(int64_t J)
{
    int t = 0;
    while (((int64_t)1 << t) < J) ++t;
    return t;
}

// level t's table: part_at [8] | support of the polynomial of (1,024 << t) blocks
inline int fill_level(uint32_t *L, int t)
{
    std::vector<uint16_t> sup16((size_t)kDeg);
    int32_t n_sup = 0;
    if (const int r = ure_host_mt_jump_support(((int64_t)kSegBlocks) << t, sup16.data(), kDeg, &n_sup)) return r;
    int32_t *part_at = reinterpret_cast<int32_t *>(L);
    int at = 0;
    for (int p = 0; p <= kParts; ++p) {
        while (at < n_sup && (int)sup16[(size_t)at] < p * kPartSpan) ++at;
        part_at[p] = p == kParts ? n_sup : at;
    }
    for (int k = kParts + 1; k < 8; ++k) part_at[k] = 0;
    for (int k = 0; k < n_sup; ++k) L[8 + k] = sup16[(size_t)k];
    return 0;
}

// The level tables of the current device: library-owned device memory (1.9 MB), filled level by level on first use and kept for the
// process; a stream that uses them waits for the event behind the newest upload.
struct LevelCache {
    uint32_t *dev = nullptr;
    int levels = 0;
    hipEvent_t ready = nullptr;
    std::vector<std::vector<uint32_t>> host;        // (the uploads' sources stay alive)
};

inline int device_levels(int levels, hipStream_t st, const uint32_t **out)
{
    static std::mutex lock;
    static std::map<int, LevelCache> cache;
    if (levels > kMaxLevels) return -1;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    std::lock_guard<std::mutex> hold(lock);
    LevelCache &C = cache[dev];
    if (!C.dev) {
        if (hipMalloc(reinterpret_cast<void **>(&C.dev), (size_t)kMaxLevels * kLevelWords * 4) != hipSuccess) return -1;
        if (hipEventCreateWithFlags(&C.ready, hipEventDisableTiming) != hipSuccess) return -1;
    }
    bool uploaded = false;
    while (C.levels < levels) {
        C.host.emplace_back((size_t)kLevelWords);
        if (const int r = fill_level(C.host.back().data(), C.levels)) return r;
        if (hipMemcpyAsync(C.dev + (int64_t)C.levels * kLevelWords, C.host.back().data(), (size_t)kLevelWords * 4, hipMemcpyHostToDevice, st) != hipSuccess) return -1;
        ++C.levels;
        uploaded = true;
    }
    if (uploaded && hipEventRecord(C.ready, st) != hipSuccess) return -1;
    if (hipStreamWaitEvent(st, C.ready, 0) != hipSuccess) return -1;
    *out = C.dev;
    return 0;
}

// the tree: states [n][J][624] with segment 0 of every stream filled and the others zero
inline void launch_tree(uint32_t *states, int n, int64_t J, const uint32_t *dev_levels, hipStream_t st)
{
    const int levels = levels_of(J);
    for (int t = 0; t < levels; ++t) {
        const int n_src = 1 << t;
        const int n_dst = (int)std::min<int64_t>(n_src, J - n_src);
        const uint32_t *L = dev_levels + (int64_t)t * kLevelWords;
        hipLaunchKernelGGL(jump_kernel, dim3((unsigned)(n * n_dst * kParts)), dim3(kLanes), 0, st, states, (int)J, n_src, n_dst, L + 8, reinterpret_cast<const int *>(L));
    }
}

}  // namespace jmp
}  // namespace ure
