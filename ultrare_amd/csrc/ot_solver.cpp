// ot_solver.cpp -- exact balanced optimal transport for the OT grouping step.
//
// Replaces the reference's call   trans = ot.emd(ones(n)/n, ones(k)/k, dist.T, 1e-3)
// followed by                     label = np.argmax(trans, axis=1)
// (method/utils.py:640-647).  `ot.emd` is POT 0.9.0's network simplex (a PyPI
// dependency that is not vendored in the reference); the 4th positional argument is
// numItermax and truncates to 0 = "no cap" (SURVEY.md D6), so the call returns an
// exact optimum of the transportation LP.  For costs in general position that optimum
// is unique, i.e. solver independent, which is what this file relies on.
//
// Formulation.  Scale masses by n*k: every point supplies k units, every cluster
// absorbs n units; the polytope is integral, so the optimum is an integer flow
// x[i][c] in 0..k.  Start from the pseudo-flow "all k units of a point on its
// cheapest cluster" (optimal for its own loads), then run successive shortest
// augmenting paths from over-full to under-full clusters on the k-node cluster
// graph, where the weight of edge a->b is min over points i holding units in a of
// cost[i][b] - cost[i][a] (kept in one lazy-deletion heap per ordered pair).  With
// k <= a few dozen clusters a Bellman-Ford pass per augmentation is trivial.
//
// Exactness.  fp32 costs are converted to int64 fixed point with a common power-of-two
// scale chosen so that every cost (and any path sum) is represented exactly; all
// comparisons are then integer comparisons, so the result is the true optimum of the
// LP whose coefficients are the fp32 costs -- no feasibility/optimality tolerances
// as in floating-point simplex codes.  Ties (a degenerate LP) are broken towards the
// lower point index, then the lower cluster index.
//
// This runs on the host by design: the LP is a sequential combinatorial problem with
// n*k <= a few million variables (milliseconds to a second), executed once per OT
// round, next to training steps that run hundreds of thousands of times.  The cost
// matrix (ure_ot_cost) and the centroid update (ure_ot_centroids) stay on the GPU.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <cstdio>
#include <limits>
#include <thread>
#include <vector>

#include "ultrare_hip.h"

namespace ure {
int fail(int code, const char *fmt, ...);
int host_threads();
}

namespace {

struct Entry {
    int64_t diff;
    int32_t i;
    uint32_t ver;
};
// std heap functions build a max-heap; invert to get the smallest (diff, i) on top
struct Later {
    bool operator()(const Entry &a, const Entry &b) const { return a.diff > b.diff || (a.diff == b.diff && a.i > b.i); }
};

constexpr int64_t kInf = std::numeric_limits<int64_t>::max() / 4;

}  // namespace

// [0, n) cut into contiguous pieces, one per host thread (the large instances -- n = 162,000, k = 32 -- spend their
// time in O(n k) passes over the cost matrix; small ones run on the calling thread)
constexpr int kMaxHostThreads = 64;      // (round 2 stopped at 16: on a 256-CPU host the O(n k) passes then ran on 6 % of it)
template <typename F>
static void parallel_ranges(int64_t n, int64_t grain, F &&body)
{
    int nt = (int)std::min<int64_t>(std::max(1u, std::min((unsigned)kMaxHostThreads, (unsigned)ure::host_threads())), (n + grain - 1) / grain);
    if (nt <= 1) { body(0, n, 0); return; }
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; ++t) pool.emplace_back([&, t]() { body(n * t / nt, n * (t + 1) / nt, t); });
    for (auto &th : pool) th.join();
}

// n k entries that are all written before they are read (the fixed-point costs), or nearly all zero (the flows): std::vector would fill 41 + 21 MB on the calling
// thread first -- at n = 162,000, k = 32 a fifth of the solver's fixed passes.  malloc leaves the pages to the threads that write (or zero) them.
template <typename T>
struct RawBuf {
    T *p = nullptr;
    size_t n = 0;
    RawBuf() = default;
    RawBuf(const RawBuf &) = delete;
    RawBuf &operator=(const RawBuf &) = delete;
    ~RawBuf() { std::free(p); }
    bool alloc(size_t m, bool zero) { std::free(p); p = static_cast<T *>(zero ? std::calloc(std::max<size_t>(m, 1), sizeof(T)) : std::malloc(std::max<size_t>(m, 1) * sizeof(T))); n = m; return p != nullptr; }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
    T *begin() { return p; }
    T *end() { return p + n; }
};

// fp32 costs [k][n] -> int64 [n][k] on a common power-of-two scale (exact unless the dynamic range is absurd)
static int to_fixed_point(const float *dist, int64_t n, int k, RawBuf<int64_t> &cost, int *shift_out)
{
    int e_min = std::numeric_limits<int>::max(), e_max = std::numeric_limits<int>::min();
    std::atomic<int64_t> bad{-1};
    int lo[kMaxHostThreads], hi[kMaxHostThreads];
    for (int t = 0; t < kMaxHostThreads; ++t) { lo[t] = e_min; hi[t] = e_max; }
    parallel_ranges(n * k, 1 << 18, [&](int64_t b, int64_t e_, int th) {
        int mn = std::numeric_limits<int>::max(), mx = std::numeric_limits<int>::min();
        for (int64_t t = b; t < e_; ++t) {
            const float v = dist[t];
            if (!(v >= 0.0f) || std::isinf(v)) { bad.store(t); return; }
            if (v == 0.0f) continue;
            uint32_t bits;
            __builtin_memcpy(&bits, &v, 4);
            int e = (int)(bits >> 23) - 126;      // v = m * 2^e, m in [0.5, 1)  (frexp convention; subnormals: below)
            if ((bits >> 23) == 0) (void)std::frexp(v, &e);
            mn = std::min(mn, e - 24);            // ulp(v) = 2^(e-24)
            mx = std::max(mx, e);
        }
        lo[th] = mn;
        hi[th] = mx;
    });
    if (bad.load() >= 0) return ure::fail(-1, "ure_ot_assign: cost %lld is negative, NaN or inf", (long long)bad.load());
    for (int t = 0; t < kMaxHostThreads; ++t) { e_min = std::min(e_min, lo[t]); e_max = std::max(e_max, hi[t]); }
    int shift = 0;                               // cost_int = v * 2^shift
    if (e_max != std::numeric_limits<int>::min()) {
        int guard = 3;
        for (int64_t g = 1; g < 4LL * k + 4; g <<= 1) ++guard;   // room for path sums over <= 2k edges
        const int budget = 62 - guard;           // bits available for a single cost
        shift = -e_min;
        if (e_max + shift > budget) shift = budget - e_max;       // absurd dynamic range: round the tiniest costs
    }
    if (!cost.alloc((size_t)n * k, false)) return ure::fail(-1, "ure_ot_assign: out of memory (%lld x %d costs)", (long long)n, k);
    const double scale = std::ldexp(1.0, shift);
    // transposing write, a block of points per thread, 128 points at a time: their k x 128 costs stay in the cache while the k passes over them fill
    // their lines (a thread's whole share per pass was 32 passes over megabytes of 256-byte strides: half of the solver's time at n = 162,000, k = 32).
    // Rounding: the costs are >= 0 (checked above) and below 2^51 (the budget), and a float times a power of two has 24 significant bits, so v + 0.5 is
    // exact and its truncation is llround(v) -- without the call.
    parallel_ranges(n, 4096, [&](int64_t b, int64_t e_, int) {
        constexpr int64_t kTile = 128;
        for (int64_t i0 = b; i0 < e_; i0 += kTile) {
            const int64_t i1 = std::min(i0 + kTile, e_);
            for (int c = 0; c < k; ++c) {
                const float *row = dist + (size_t)c * n;
                for (int64_t i = i0; i < i1; ++i) cost[(size_t)i * k + c] = (int64_t)((double)row[i] * scale + 0.5);
            }
        }
    });
    *shift_out = shift;
    return 0;
}

extern "C" int ure_ot_assign(const float *dist, int64_t n, int k, int32_t *label, int32_t *plan_nk, double *total_cost)
{
    if (!dist || !label || n <= 0 || k <= 0 || k > 4096 || n > (int64_t)1 << 31)
        return ure::fail(-1, "ure_ot_assign: bad arguments (n=%lld k=%d)", (long long)n, k);

    // ---- fixed-point scale -------------------------------------------------------
    RawBuf<int64_t> cost;
    int shift = 0;
    if (int rc = to_fixed_point(dist, n, k, cost, &shift)) return rc;

    // ---- initial pseudo-flow -----------------------------------------------------
    RawBuf<int32_t> x;
    if (!x.alloc((size_t)n * k, false)) return ure::fail(-1, "ure_ot_assign: out of memory (%lld x %d flows)", (long long)n, k);
    parallel_ranges(n * k, 1 << 20, [&](int64_t b, int64_t e_, int) { std::memset(x.p + b, 0, (size_t)(e_ - b) * sizeof(int32_t)); });     // (zeroed -- and its pages touched -- by the threads)
    std::vector<uint32_t> ver((size_t)n * k, 0);
    std::vector<int64_t> load(k, 0);
    std::vector<std::vector<Entry>> heap((size_t)k * k);
    for (int64_t i = 0; i < n; ++i) {
        const int64_t *ci = &cost[(size_t)i * k];
        int best = 0;
        for (int c = 1; c < k; ++c)
            if (ci[c] < ci[best]) best = c;
        x[(size_t)i * k + best] = k;
        load[best] += k;
        for (int b = 0; b < k; ++b)
            if (b != best) heap[(size_t)best * k + b].push_back({ci[b] - ci[best], (int32_t)i, 0u});
    }
    for (auto &h : heap) std::make_heap(h.begin(), h.end(), Later());

    std::vector<int64_t> w((size_t)k * k, kInf);
    std::vector<int32_t> arg((size_t)k * k, -1);
    std::vector<char> dirty(k, 1);
    auto refresh_row = [&](int a) {
        for (int b = 0; b < k; ++b) {
            if (b == a) continue;
            auto &h = heap[(size_t)a * k + b];
            while (!h.empty()) {
                const Entry &t = h.front();
                if (x[(size_t)t.i * k + a] > 0 && t.ver == ver[(size_t)t.i * k + a]) break;
                std::pop_heap(h.begin(), h.end(), Later());
                h.pop_back();
            }
            if (h.empty()) { w[(size_t)a * k + b] = kInf; arg[(size_t)a * k + b] = -1; }
            else { w[(size_t)a * k + b] = h.front().diff; arg[(size_t)a * k + b] = h.front().i; }
        }
        dirty[a] = 0;
    };

    // ---- successive shortest augmenting paths -------------------------------------
    std::vector<int64_t> dst(k);
    std::vector<int> pred(k), path;
    for (;;) {
        bool any_excess = false;
        for (int c = 0; c < k; ++c) any_excess = any_excess || load[c] > n;
        if (!any_excess) break;
        for (int a = 0; a < k; ++a)
            if (dirty[a]) refresh_row(a);
        for (int c = 0; c < k; ++c) { dst[c] = load[c] > n ? 0 : kInf; pred[c] = -1; }
        for (int pass = 0; pass < k; ++pass) {
            bool changed = false;
            for (int a = 0; a < k; ++a) {
                if (dst[a] >= kInf) continue;
                for (int b = 0; b < k; ++b) {
                    const int64_t wab = w[(size_t)a * k + b];
                    if (b == a || wab >= kInf) continue;
                    if (dst[a] + wab < dst[b]) { dst[b] = dst[a] + wab; pred[b] = a; changed = true; }
                }
            }
            if (!changed) break;
        }
        int tgt = -1;
        for (int c = 0; c < k; ++c)
            if (load[c] < n && dst[c] < kInf && (tgt < 0 || dst[c] < dst[tgt])) tgt = c;
        if (tgt < 0) return ure::fail(-2, "ure_ot_assign: no augmenting path (internal error)");
        path.clear();
        for (int c = tgt; c >= 0; c = pred[c]) {
            path.push_back(c);
            if ((int)path.size() > k) return ure::fail(-2, "ure_ot_assign: predecessor cycle (internal error)");
        }
        std::reverse(path.begin(), path.end());          // src ... tgt
        const int src = path.front();
        int64_t delta = std::min(load[src] - n, n - load[tgt]);
        for (size_t e = 0; e + 1 < path.size(); ++e) {
            const int a = path[e], b = path[e + 1];
            delta = std::min<int64_t>(delta, x[(size_t)arg[(size_t)a * k + b] * k + a]);
        }
        if (delta <= 0) return ure::fail(-2, "ure_ot_assign: zero augmentation (internal error)");
        for (size_t e = 0; e + 1 < path.size(); ++e) {
            const int a = path[e], b = path[e + 1];
            const int64_t i = arg[(size_t)a * k + b];
            x[(size_t)i * k + a] -= (int32_t)delta;
            int32_t &xb = x[(size_t)i * k + b];
            if (xb == 0) {
                const uint32_t v = ++ver[(size_t)i * k + b];
                const int64_t *ci = &cost[(size_t)i * k];
                for (int c = 0; c < k; ++c) {
                    if (c == b) continue;
                    auto &h = heap[(size_t)b * k + c];
                    h.push_back({ci[c] - ci[b], (int32_t)i, v});
                    std::push_heap(h.begin(), h.end(), Later());
                }
            }
            xb += (int32_t)delta;
            dirty[a] = dirty[b] = 1;
        }
        load[src] -= delta;
        load[tgt] += delta;
    }

    // ---- outputs --------------------------------------------------------------------
    // labels by the threads; the objective's terms are then added on this thread in the order they always were (points ascending, clusters ascending):
    // a point's first nonzero flow is noted by the pass, the few points split over clusters are walked again
    long double obj = 0.0L;
    {
        std::vector<int32_t> first_c((size_t)n);
        std::vector<char> more((size_t)n);
        parallel_ranges(n, 8192, [&](int64_t b, int64_t e_, int) {
            for (int64_t i = b; i < e_; ++i) {
                const int32_t *xi = &x[(size_t)i * k];
                int best = 0, fc = -1, nz = 0;
                for (int c = 0; c < k; ++c) {
                    if (xi[c] > xi[best]) best = c;                // np.argmax: first maximum
                    if (xi[c]) { if (fc < 0) fc = c; ++nz; }
                }
                label[i] = best;
                first_c[(size_t)i] = fc;
                more[(size_t)i] = nz > 1;
            }
        });
        for (int64_t i = 0; i < n; ++i) {
            const int fc = first_c[(size_t)i];
            if (fc < 0) continue;
            if (!more[(size_t)i]) { obj += (long double)x[(size_t)i * k + fc] * (long double)dist[(size_t)fc * n + i]; continue; }
            for (int c = fc; c < k; ++c) {
                const int32_t v = x[(size_t)i * k + c];
                if (v) obj += (long double)v * (long double)dist[(size_t)c * n + i];
            }
        }
    }
    if (plan_nk) std::copy(x.begin(), x.end(), plan_nk);
    if (total_cost) *total_cost = (double)(obj / ((long double)n * (long double)k));
    return 0;
}


// ---------------------------------------------------------------------------------------------------
// The same LP from a warm start.  `pi` = potentials of the clusters (any values: ure_ot_potentials finds good
// ones on the GPU, the previous round's serve too).  Every point starts on the cluster of its cheapest REDUCED
// cost cost[i][c] - pi[c]: that pseudo-flow is optimal for its own loads (the residual graph has no negative
// cycle: around a cycle the potentials cancel), which is all successive shortest paths needs.  With loads a few
// points away from balance only tens of augmentations remain, so the cluster graph's edge weights
//     w[a][b] = min over points i holding units in a of cost[i][b] - cost[i][a]
// are kept as plain (min, argmin) pairs and a cluster's row is recomputed from its member list when its argmin
// leaves -- O(n) per augmentation, against the O(n k) heap entries of the cold start.  Falls back to the cold
// start when the warm start is poor (more than `kWarmMaxMoves` points to move).
// ---------------------------------------------------------------------------------------------------
extern "C" int ure_ot_assign_warm(const float *dist, int64_t n, int k, const double *pi, int32_t *label, int32_t *plan_nk,
                                  double *total_cost, int64_t *augmentations)
{
    constexpr int64_t kWarmMaxMoves = 2048;
    if (!dist || !label || n <= 0 || k <= 0 || k > 4096 || n > (int64_t)1 << 31)
        return ure::fail(-1, "ure_ot_assign_warm: bad arguments (n=%lld k=%d)", (long long)n, k);
    if (augmentations) *augmentations = -1;
    if (!pi) return ure_ot_assign(dist, n, k, label, plan_nk, total_cost);
    RawBuf<int64_t> cost;
    int shift = 0;
    if (int rc = to_fixed_point(dist, n, k, cost, &shift)) return rc;
    std::vector<int64_t> pot(k);
    const double scale = std::ldexp(1.0, shift);
    for (int c = 0; c < k; ++c) {
        if (!std::isfinite(pi[c]) || std::fabs(pi[c]) * scale > 4e18 / (4.0 * k + 4)) return ure_ot_assign(dist, n, k, label, plan_nk, total_cost);
        pot[c] = (int64_t)std::llround(pi[c] * scale);
    }

    // ---- initial pseudo-flow: cheapest reduced cost (ties: lowest cluster) ---------------------------
    RawBuf<int32_t> x;
    if (!x.alloc((size_t)n * k, false)) return ure::fail(-1, "ure_ot_assign: out of memory (%lld x %d flows)", (long long)n, k);
    parallel_ranges(n * k, 1 << 20, [&](int64_t b, int64_t e_, int) { std::memset(x.p + b, 0, (size_t)(e_ - b) * sizeof(int32_t)); });     // (zeroed -- and its pages touched -- by the threads)
    std::vector<int64_t> load(k, 0);
    std::vector<std::vector<int32_t>> members(k);
    for (int c = 0; c < k; ++c) members[c].reserve((size_t)(n / k + n / (4 * k) + 16));
    std::vector<int32_t> first(n);
    parallel_ranges(n, 8192, [&](int64_t b, int64_t e_, int) {
        for (int64_t i = b; i < e_; ++i) {
            const int64_t *ci = &cost[(size_t)i * k];
            int best = 0;
            for (int c = 1; c < k; ++c)
                if (ci[c] - pot[c] < ci[best] - pot[best]) best = c;
            first[i] = best;
        }
    });
    for (int64_t i = 0; i < n; ++i) {                            // member lists in ascending point id
        const int best = first[i];
        x[(size_t)i * k + best] = k;
        load[best] += k;
        members[best].push_back((int32_t)i);
    }
    int64_t moves = 0;
    for (int c = 0; c < k; ++c) moves += load[c] > n ? (load[c] - n + k - 1) / k : 0;
    if (moves > kWarmMaxMoves) return ure_ot_assign(dist, n, k, label, plan_nk, total_cost);

    std::vector<int64_t> w((size_t)k * k, kInf);
    std::vector<int32_t> arg((size_t)k * k, -1);
    auto rescan_row = [&](int a) {
        int64_t *wa = &w[(size_t)a * k];
        int32_t *ga = &arg[(size_t)a * k];
        for (int b = 0; b < k; ++b) { wa[b] = kInf; ga[b] = -1; }
        auto &mem = members[a];
        size_t keep = 0;
        for (size_t q = 0; q < mem.size(); ++q) {
            const int32_t i = mem[q];
            if (x[(size_t)i * k + a] <= 0) continue;             // left the cluster: dropped from the list here
            mem[keep++] = i;
            const int64_t *ci = &cost[(size_t)i * k];
            const int64_t base = ci[a];
            for (int b = 0; b < k; ++b) {
                const int64_t dlt = ci[b] - base;
                if (dlt < wa[b] || (dlt == wa[b] && i < ga[b])) { wa[b] = dlt; ga[b] = i; }      // ties: lowest point index
            }
        }
        mem.resize(keep);
        wa[a] = kInf;
        ga[a] = -1;
    };
    {   // all rows once, in parallel (each touches its own row of w / arg and its own member list)
        std::atomic<int> next{0};
        parallel_ranges(std::min<int64_t>(k, kMaxHostThreads), 1, [&](int64_t, int64_t, int) {
            for (int a = next.fetch_add(1); a < k; a = next.fetch_add(1)) rescan_row(a);
        });
    }
    // one edge a -> b again from a's member list (the point that realised it has left a)
    auto rescan_edge = [&](int a, int b) {
        int64_t best = kInf;
        int32_t who = -1;
        for (const int32_t i : members[a]) {
            if (x[(size_t)i * k + a] <= 0) continue;
            const int64_t dlt = cost[(size_t)i * k + b] - cost[(size_t)i * k + a];
            if (dlt < best || (dlt == best && i < who)) { best = dlt; who = i; }
        }
        w[(size_t)a * k + b] = best;
        arg[(size_t)a * k + b] = who;
    };

    std::vector<int64_t> dst(k);
    std::vector<int> pred(k), path;
    std::vector<char> dirty(k, 0);
    int64_t n_aug = 0;
    for (;;) {
        bool any_excess = false;
        for (int c = 0; c < k; ++c) any_excess = any_excess || load[c] > n;
        if (!any_excess) break;
        for (int a = 0; a < k; ++a) {
            if (!dirty[a]) continue;
            // only the edges whose argmin point no longer holds units in a; the member list is compacted now and then
            if (members[a].size() > 2 * (size_t)(load[a] / k + 64)) rescan_row(a);
            else
                for (int b = 0; b < k; ++b) {
                    const int32_t i = arg[(size_t)a * k + b];
                    if (b != a && i >= 0 && x[(size_t)i * k + a] <= 0) rescan_edge(a, b);
                }
            dirty[a] = 0;
        }
        for (int c = 0; c < k; ++c) { dst[c] = load[c] > n ? 0 : kInf; pred[c] = -1; }
        for (int pass = 0; pass < k; ++pass) {
            bool changed = false;
            for (int a = 0; a < k; ++a) {
                if (dst[a] >= kInf) continue;
                const int64_t *wa = &w[(size_t)a * k];
                for (int b = 0; b < k; ++b) {
                    if (b == a || wa[b] >= kInf) continue;
                    if (dst[a] + wa[b] < dst[b]) { dst[b] = dst[a] + wa[b]; pred[b] = a; changed = true; }
                }
            }
            if (!changed) break;
        }
        int tgt = -1;
        for (int c = 0; c < k; ++c)
            if (load[c] < n && dst[c] < kInf && (tgt < 0 || dst[c] < dst[tgt])) tgt = c;
        if (tgt < 0) return ure::fail(-2, "ure_ot_assign_warm: no augmenting path (internal error)");
        path.clear();
        for (int c = tgt; c >= 0; c = pred[c]) {
            path.push_back(c);
            if ((int)path.size() > k) return ure::fail(-2, "ure_ot_assign_warm: predecessor cycle (internal error)");
        }
        std::reverse(path.begin(), path.end());
        const int src = path.front();
        int64_t delta = std::min(load[src] - n, n - load[tgt]);
        for (size_t e = 0; e + 1 < path.size(); ++e) {
            const int a = path[e], b = path[e + 1];
            delta = std::min<int64_t>(delta, x[(size_t)arg[(size_t)a * k + b] * k + a]);
        }
        if (delta <= 0) return ure::fail(-2, "ure_ot_assign_warm: zero augmentation (internal error)");
        for (size_t e = 0; e + 1 < path.size(); ++e) {
            const int a = path[e], b = path[e + 1];
            const int64_t i = arg[(size_t)a * k + b];
            int32_t &xa = x[(size_t)i * k + a];
            int32_t &xb = x[(size_t)i * k + b];
            xa -= (int32_t)delta;
            if (xb == 0) {                                       // the point joins b: it may lower b's outgoing edges
                members[b].push_back((int32_t)i);
                const int64_t *ci = &cost[(size_t)i * k];
                for (int c = 0; c < k; ++c) {
                    if (c == b) continue;
                    const int64_t dlt = ci[c] - ci[b];
                    int64_t &wbc = w[(size_t)b * k + c];
                    int32_t &gbc = arg[(size_t)b * k + c];
                    if (dlt < wbc || (dlt == wbc && (int32_t)i < gbc)) { wbc = dlt; gbc = (int32_t)i; }
                }
            }
            xb += (int32_t)delta;
            if (xa == 0) dirty[a] = 1;                           // it may have been the argmin of any edge out of a
        }
        load[src] -= delta;
        load[tgt] += delta;
        ++n_aug;
    }

    // labels by the threads; the objective's terms are then added on this thread in the order they always were (points ascending, clusters ascending):
    // a point's first nonzero flow is noted by the pass, the few points split over clusters are walked again
    long double obj = 0.0L;
    {
        std::vector<int32_t> first_c((size_t)n);
        std::vector<char> more((size_t)n);
        parallel_ranges(n, 8192, [&](int64_t b, int64_t e_, int) {
            for (int64_t i = b; i < e_; ++i) {
                const int32_t *xi = &x[(size_t)i * k];
                int best = 0, fc = -1, nz = 0;
                for (int c = 0; c < k; ++c) {
                    if (xi[c] > xi[best]) best = c;                // np.argmax: first maximum
                    if (xi[c]) { if (fc < 0) fc = c; ++nz; }
                }
                label[i] = best;
                first_c[(size_t)i] = fc;
                more[(size_t)i] = nz > 1;
            }
        });
        for (int64_t i = 0; i < n; ++i) {
            const int fc = first_c[(size_t)i];
            if (fc < 0) continue;
            if (!more[(size_t)i]) { obj += (long double)x[(size_t)i * k + fc] * (long double)dist[(size_t)fc * n + i]; continue; }
            for (int c = fc; c < k; ++c) {
                const int32_t v = x[(size_t)i * k + c];
                if (v) obj += (long double)v * (long double)dist[(size_t)c * n + i];
            }
        }
    }
    if (plan_nk) std::copy(x.begin(), x.end(), plan_nk);
    if (total_cost) *total_cost = (double)(obj / ((long double)n * (long double)k));
    if (augmentations) *augmentations = n_aug;
    return 0;
}
