// mf_eval.hip -- ensemble scoring, per-user HR@10 / NDCG@10 and the SISA user-row
// merge for gfx950.
//
// Replaces
//   baseTest scoring    method/utils.py:140-148  pred = stack(preds).mean(0); sum (pred-r)^2
//   baseTest ranking    method/utils.py:165-184  per-user argsort top-10, HR, positional NDCG
//   computeNDCG / DCG   method/utils.py:190-210
//   user-row merge      method/sisa.py:52-58, 107-113
//
// The reference walks Python dicts per user; here one wavefront owns one user and
// extracts the two top-10 lists by repeated wave-wide arg-max over (value, position)
// keys -- exactly the order of a stable ascending argsort read backwards -- and one
// lane finishes the float64 DCG in numpy's summation order so that HR is exact and
// NDCG is bit-identical whenever the two rankings agree with the reference's.
#include "ure_internal.h"

#include <cfloat>
#include <climits>

namespace ure {

struct TableList {
    const float *U[URE_MAX_MODELS_PER_CALL];
    const float *V[URE_MAX_MODELS_PER_CALL];
};

template <int LPR>
__global__ __launch_bounds__(kBlock) void score_kernel(TableList T, int n_models, int n_total, int first, int last,
                                                        const int32_t *__restrict__ uid, const int32_t *__restrict__ iid,
                                                        const float *__restrict__ rating, int64_t n,
                                                        float *__restrict__ pred, double *__restrict__ sse)
{
    constexpr int D = LPR * 4;
    constexpr int G = kWave / LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane & (LPR - 1), grp = lane / LPR;
    const int64_t wave_id = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * kBlock) >> 6;
    float sq = 0.f;
    for (int64_t base = wave_id * G; base < n; base += n_waves * G) {
        const int64_t j = base + grp;
        const bool act = j < n;
        const int u = act ? uid[j] : 0, i = act ? iid[j] : 0;
        float acc = (act && !first) ? pred[j] : 0.f;
        // the rows of four models are fetched together (eight independent 16-byte gathers per lane);
        // the sum over models stays in list order (utils.py:145 stack(...).mean(0))
        for (int m0 = 0; m0 < n_models; m0 += 4) {
            float4 a[4], b[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                a[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                b[k] = a[k];
                if (m0 + k < n_models) {
                    a[k] = *reinterpret_cast<const float4 *>(T.U[m0 + k] + (size_t)u * D + sub * 4);
                    b[k] = *reinterpret_cast<const float4 *>(T.V[m0 + k] + (size_t)i * D + sub * 4);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float p = a[k].x * b[k].x;
                p = fmaf(a[k].y, b[k].y, p);
                p = fmaf(a[k].z, b[k].z, p);
                p = fmaf(a[k].w, b[k].w, p);
                p = group_sum<LPR>(p);
                if (m0 + k < n_models) acc += p;
            }
        }
        if (last) {
            acc = acc / (float)n_total;
            if (act && sub == 0 && sse) {
                const float e = acc - rating[j];
                sq = fmaf(e, e, sq);
            }
        }
        if (act && sub == 0) pred[j] = acc;
    }
    if (last && sse) {
        // one partial per workgroup, no atomics: thousands of waves adding to ONE address serialise
        // at ~12 ns each (measured: 69 us for this kernel); the partials are summed in a fixed
        // order by ure_eval_reduce or by the host
        __shared__ float part[kWavesPerBlock];
        sq = wave_sum(sq);
        if (lane == 0) part[threadIdx.x >> 6] = sq;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < kWavesPerBlock; ++k) t += (double)part[k];
            sse[blockIdx.x] = t;
        }
        for (int t = gridDim.x + threadIdx.x; blockIdx.x == 0 && t < URE_SCORE_PARTIALS; t += kBlock) sse[t] = 0.0;
    }
}

// One launch for a SERIES of ensembles that differ in their last model only (scratch.py:83-97 over
// the epochs of a shard: the models trained before it + its own model after epoch e): blockIdx.y = e,
// base[j] = the running sum over the fixed models (ure_score with first = 1, last = 0; NULL when
// there are none), the last model's tables are U + e * stride_u, V + e * stride_v.  The additions
// happen in the same order as in score_kernel over the whole list, so the results are identical.
template <int LPR>
__global__ __launch_bounds__(kBlock) void score_series_kernel(const float *__restrict__ U, const float *__restrict__ V,
                                                               int64_t stride_u, int64_t stride_v, int n_total,
                                                               const int32_t *__restrict__ uid, const int32_t *__restrict__ iid,
                                                               const float *__restrict__ rating, int64_t n,
                                                               const float *__restrict__ base, float *__restrict__ pred,
                                                               double *__restrict__ sse)
{
    constexpr int D = LPR * 4;
    constexpr int G = kWave / LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane & (LPR - 1), grp = lane / LPR;
    const int64_t wave_id = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * kBlock) >> 6;
    U += (size_t)blockIdx.y * stride_u;
    V += (size_t)blockIdx.y * stride_v;
    pred += (size_t)blockIdx.y * n;
    sse += (size_t)blockIdx.y * URE_SCORE_PARTIALS;
    float sq = 0.f;
    for (int64_t j0 = wave_id * G; j0 < n; j0 += n_waves * G) {
        const int64_t j = j0 + grp;
        const bool act = j < n;
        const int u = act ? uid[j] : 0, i = act ? iid[j] : 0;
        float acc = (act && base) ? base[j] : 0.f;
        const float4 a = *reinterpret_cast<const float4 *>(U + (size_t)u * D + sub * 4);
        const float4 b = *reinterpret_cast<const float4 *>(V + (size_t)i * D + sub * 4);
        float p = a.x * b.x;
        p = fmaf(a.y, b.y, p);
        p = fmaf(a.z, b.z, p);
        p = fmaf(a.w, b.w, p);
        acc += group_sum<LPR>(p);
        acc = acc / (float)n_total;
        if (act && sub == 0) {
            const float e = acc - rating[j];
            sq = fmaf(e, e, sq);
            pred[j] = acc;
        }
    }
    __shared__ float part[kWavesPerBlock];
    sq = wave_sum(sq);
    if (lane == 0) part[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < kWavesPerBlock; ++k) t += (double)part[k];
        sse[blockIdx.x] = t;
    }
    for (int t = gridDim.x + threadIdx.x; blockIdx.x == 0 && t < URE_SCORE_PARTIALS; t += kBlock) sse[t] = 0.0;
}

// The scores of a shard's OWN model on a test set for a run of epochs, from compact snapshots (struct ure_shard: snap): member e's
// tables hold only the shard's active rows, in schedule order; every other row is snap_a[e] * w0 by construction and is rebuilt here
// with the product the full snapshot stores (snapshot_kernel: a * w0 per element), so both forms give identical predictions.  The
// other half of a series -- the fixed models, the division, the squared errors -- follows in series_combine_kernel.
// The epochs run INSIDE the pair's loop: a lane group keeps its pair for all members of the series, so a row that
// is not stored in the snapshots (every user of another shard, four fifths of a total test set at five shards) is loaded ONCE -- as
// w0 -- and only scaled per epoch; the stored rows are gathered per epoch as before.  score_series_compact_kernel fetched both rows
// of every pair for every epoch (round 2 / early round 3: one grid row per epoch): 10.2 M row gathers per call on the total test set
// of ml-1m, L2-bound.
template <int LPR>
__global__ __launch_bounds__(kBlock) void score_own_epochs_kernel(const float *__restrict__ snap, int64_t stride, const int32_t *__restrict__ row_slot,
                                                                   const float *__restrict__ U0, const float *__restrict__ V0,
                                                                   const float *__restrict__ snap_a, int n_user_rows, int n_series,
                                                                   const int32_t *__restrict__ uid, const int32_t *__restrict__ iid, int64_t n,
                                                                   float *__restrict__ own)
{
    constexpr int D = LPR * 4;
    constexpr int G = kWave / LPR;
    constexpr int kE = 4;                 // epochs in flight per pair (8: 61.6 -> 65.7 us per call, measured r5)
    const int lane = threadIdx.x & 63;
    const int sub = lane & (LPR - 1), grp = lane / LPR;
    const int64_t wave_id = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * kBlock) >> 6;
    for (int64_t j0 = wave_id * G; j0 < n; j0 += n_waves * G) {
        const int64_t j = j0 + grp;
        const bool act = j < n;
        const int u = act ? uid[j] : 0, i = act ? iid[j] : 0;
        const int su = row_slot[u], si = row_slot[n_user_rows + i];
        // rows that are not stored: w0, once
        const float4 a0 = *reinterpret_cast<const float4 *>(U0 + (size_t)u * D + sub * 4);
        const float4 b0 = *reinterpret_cast<const float4 *>(V0 + (size_t)i * D + sub * 4);
        const float *pa = snap + (size_t)(su >= 0 ? su : 0) * D + sub * 4;
        const float *pb = snap + (size_t)(si >= 0 ? si : 0) * D + sub * 4;
        for (int e0 = 0; e0 < n_series; e0 += kE) {
            float4 a[kE], b[kE];
#pragma unroll
            for (int k = 0; k < kE; ++k) {
                const int e = min(e0 + k, n_series - 1);
                a[k] = a0;
                b[k] = b0;
                if (su >= 0) a[k] = *reinterpret_cast<const float4 *>(pa + (size_t)e * stride);
                if (si >= 0) b[k] = *reinterpret_cast<const float4 *>(pb + (size_t)e * stride);
            }
#pragma unroll
            for (int k = 0; k < kE; ++k) {
                const float ae = snap_a[min(e0 + k, n_series - 1)];
                float4 x = a[k], y = b[k];
                if (su < 0) x = make_float4(ae * x.x, ae * x.y, ae * x.z, ae * x.w);
                if (si < 0) y = make_float4(ae * y.x, ae * y.y, ae * y.z, ae * y.w);
                float p = x.x * y.x;
                p = fmaf(x.y, y.y, p);
                p = fmaf(x.z, y.z, p);
                p = fmaf(x.w, y.w, p);
                p = group_sum<LPR>(p);
                if (act && sub == 0 && e0 + k < n_series) own[(size_t)(e0 + k) * n + j] = p;
            }
        }
    }
}

// pred[e][j] = (base[j] + own[e][j]) / n_total and the squared-error partials of member e -- the additions, the division AND the
// order in which score_kernel sums the squared errors, so that a series member agrees with a single evaluation to the last bit.
// score_kernel's order, for `blocks` workgroups of four waves of G = 64 / LPR lane groups: accumulator (wave w, group g) takes the pairs
// (w G + g) + k (4 blocks G), k = 0, 1, ... in turn; a wave adds its G accumulators as an xor butterfly; a workgroup adds its four
// waves in double, in order.  Here every THREAD is one such accumulator (consecutive threads = consecutive pairs: coalesced, where one
// lane in LPR worked before: 49 -> us per call), G consecutive threads are a logical wave, 4 G a logical workgroup.
template <int LPR>
__global__ __launch_bounds__(kBlock) void series_combine_kernel(const float *own, int n_total, const float *__restrict__ rating, int64_t n,
                                                                const float *__restrict__ base, float *pred, double *__restrict__ sse,
                                                                int blocks)   // (own may be pred: in place)
{
    constexpr int G = kWave / LPR;
    own += (size_t)blockIdx.y * n;
    pred += (size_t)blockIdx.y * n;
    sse += (size_t)blockIdx.y * URE_SCORE_PARTIALS;
    const int64_t n_acc = (int64_t)blocks * kWavesPerBlock * G;          // accumulators = the stride of an accumulator's pairs
    const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;        // this thread's accumulator
    float sq = 0.f;
    if (q < n_acc) {
        for (int64_t j = q; j < n; j += n_acc) {
            float acc = base ? base[j] : 0.f;
            acc += own[j];
            acc = acc / (float)n_total;
            const float e = acc - rating[j];
            sq = fmaf(e, e, sq);
            pred[j] = acc;
        }
    }
    // the logical wave's butterfly over its G accumulators (the other lanes of score_kernel's wave hold zeros there)
#pragma unroll
    for (int o = 1; o < G; o <<= 1) sq += __shfl_xor(sq, o, kWave);
    __shared__ float wave_tot[kBlock];
    wave_tot[threadIdx.x] = sq;
    __syncthreads();
    // the logical workgroup: its four waves, in double, in order
    constexpr int kPerBlock = kWavesPerBlock * G;                        // threads of a logical workgroup
    if (threadIdx.x % kPerBlock == 0 && q < n_acc) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < kWavesPerBlock; ++k) t += (double)wave_tot[threadIdx.x + k * G];
        sse[q / kPerBlock] = t;
    }
    for (int t = blocks + (int)threadIdx.x; blockIdx.x == 0 && t < URE_SCORE_PARTIALS; t += kBlock) sse[t] = 0.0;
}

// (value, position) keys ordered lexicographically; "better" = later in a stable
// ascending argsort, i.e. earlier in its reverse (utils.py:169-170).
__device__ __forceinline__ bool key_gt(float v, int i, float bv, int bi) { return v > bv || (v == bv && i > bi); }

// A model that diverged (the reference's summed-loss SGD does on heavy users, e.g. two of the 16 shards of
// BASELINE.json configs[4] on the synthetic set) predicts NaN.  np.argsort places NaN after every number, so its
// reverse ranks NaN first: NaN is ordered as +inf here, which keeps the keys totally ordered -- without it no entry
// has rank k for some k and the selection returns position -1.
__device__ __forceinline__ float nan_last(float v) { return v != v ? __builtin_inff() : v; }

// The best (value, position) key of the wave, on every lane: four DPP exchanges inside each row of 16 lanes, then the four row
// winners through scalar registers.  (As six __shfl_xor steps -- twelve dependent ds_bpermute -- a selection round took ~0.8 us
// and the ten rounds of the 199 users with more than 64 test entries were the tail of every eval launch: 8 us for one member, r3.)
template <int CTRL>
__device__ __forceinline__ void argmax_step(float &v, int &i)
{
    const float ov = dpp_f<CTRL>(v);
    const int oi = dpp_i<CTRL>(i);
    if (key_gt(ov, oi, v, i)) { v = ov; i = oi; }
}
__device__ __forceinline__ void wave_argmax(float &v, int &i)
{
    argmax_step<kDppQuadXor1>(v, i);
    argmax_step<kDppQuadXor2>(v, i);
    argmax_step<kDppHalfMirror>(v, i);
    argmax_step<kDppRowMirror>(v, i);
    const int vi = __builtin_bit_cast(int, v);
    float bv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 0));
    int bi = __builtin_amdgcn_readlane(i, 0);
#pragma unroll
    for (int r = 1; r < 4; ++r) {
        const float ov = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 16 * r));
        const int oi = __builtin_amdgcn_readlane(i, 16 * r);
        if (key_gt(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    v = bv;
    i = bi;
}

// Positions (within the user's segment) of the top-K keys of `val`, best first.
template <int K>
__device__ __forceinline__ void top_k_positions(const float *__restrict__ val, int cnt, int lane, int (&top)[K])
{
    float pv = FLT_MAX;   // previous pick: everything is "less" than it on the first round
    int pi = INT_MAX;
    bool pinf = true;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float bv = -FLT_MAX;
        int bi = -1;
        for (int t = lane; t < cnt; t += kWave) {
            const float v = nan_last(val[t]);
            const bool below_prev = pinf || v < pv || (v == pv && t < pi);
            if (below_prev && (bi < 0 || key_gt(v, t, bv, bi))) { bv = v; bi = t; }
        }
        // lanes without a candidate carry (−FLT_MAX, −1), which loses to any real key
        wave_argmax(bv, bi);
        top[k] = bi;
        pv = bv; pi = bi; pinf = false;
    }
}

// The same selection when the segment has at most 64 entries: lane t holds entry t.
// Every lane counts the entries that beat its own (val, position) key -- `cnt` independent
// broadcasts instead of ten dependent arg-max rounds -- and the entry of rank k is top[k].
// FAST (predictions): ranks are counted with the strict comparison alone -- three instructions per entry instead of six --
// and are exact unless two of the ten best keys are EQUAL, which shows as two entries with one rank; only then (the wave
// decides as one) the ranks are counted again with the position as the tie-break.  Ratings tie all the time: FAST off.
template <int K, bool FAST = false>
__device__ __forceinline__ void top_k_in_registers(float val, int cnt, int lane, int (&top)[K])
{
    // entry t is broadcast through a scalar register (v_readlane: t is wave-uniform) -- a ds_bpermute per entry, as __shfl
    // compiles to, sends all 64 lanes through the LDS crossbar for it (eval_users: 125 -> 7x us per series call, r3)
    const int vi = __builtin_bit_cast(int, val);
    // the lanes that hold an entry, as a mask: a ballot of (lane < cnt && rank == k) compiles to a select and a second compare
    const unsigned long long valid = __builtin_amdgcn_ballot_w64(lane < cnt);
    int rank = 0;
    bool exact = !FAST;
    if (FAST) {
        for (int t = 0; t < cnt; ++t) rank += __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, t)) > val ? 1 : 0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(rank == k) & valid;
            top[k] = m ? (int)__builtin_ctzll(m) : -1;
            exact = exact || (m & (m - 1)) != 0;
        }
    }
    if (exact) {
        rank = 0;
        for (int t = 0; t < cnt; ++t) {
            const float ov = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, t));
            rank += key_gt(ov, t, val, lane) ? 1 : 0;
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(rank == k) & valid;
            top[k] = m ? (int)__builtin_ctzll(m) : -1;
        }
    }
}

// Segments of up to 64 * kRegItems entries: lane t holds entries t, t+64, ... in registers; every
// round takes the best not-yet-taken key of the wave (no memory access per round).
constexpr int kRegItems = 8;
template <int K, int R>
__device__ __forceinline__ void top_k_multi(const float (&val)[R], int cnt, int lane, int (&top)[K])
{
    unsigned taken = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) taken |= (lane + r * kWave >= cnt ? 1u : 0u) << r;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float bv = -FLT_MAX;
        int bi = -1;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int t = lane + r * kWave;
            if (!((taken >> r) & 1u) && (bi < 0 || key_gt(val[r], t, bv, bi))) { bv = val[r]; bi = t; }
        }
        wave_argmax(bv, bi);
        top[k] = bi;
        if (bi >= 0 && (bi & (kWave - 1)) == lane) taken |= 1u << (bi / kWave);
    }
}

// One rotation of a 16-lane row by N lanes (DPP row_ror): a VALU move, no LDS crossbar.  The entry's position travels with
// its value, so nothing here depends on the direction of the rotation.
template <int N>
__device__ __forceinline__ void quarter_rank_steps(const int vi, const int s, const int cnt, const float val, int &rank)
{
    const float ov = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, vi, 0x120 + N, 0xF, 0xF, false));
    const int ot = __builtin_amdgcn_update_dpp(0, s, 0x120 + N, 0xF, 0xF, false);
    rank += (ot < cnt && key_gt(ov, ot, val, s)) ? 1 : 0;
    if constexpr (N < 15) quarter_rank_steps<N + 1>(vi, s, cnt, val, rank);
}
// The strict comparison alone (lanes past the segment carry -inf, which is greater than nothing): no position travels.
template <int N>
__device__ __forceinline__ void quarter_rank_steps_strict(const int vi, const float val, int &rank)
{
    rank += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, vi, 0x120 + N, 0xF, 0xF, false)) > val ? 1 : 0;
    if constexpr (N < 15) quarter_rank_steps_strict<N + 1>(vi, val, rank);
}

// Rank of every entry of a segment of at most 16 entries held one per lane by a quarter wave (lanes 16 g .. 16 g + 15):
// the in-register selection of top_k_in_registers at width 16, four users per wavefront.  FAST as there: the wave falls
// back to the exact count when any of its four users has equal keys among its ten best.
template <int K, bool FAST = false>
__device__ __forceinline__ void top_k_quarter(float val, int cnt, int lane, int (&top)[K])
{
    const int g16 = lane & ~15, s = lane & 15;
    int rank = 0;
    bool exact = !FAST;
    if (FAST) {
        const float padded = s < cnt ? val : -__builtin_inff();
        quarter_rank_steps_strict<1>(__builtin_bit_cast(int, padded), padded, rank);
        bool tie = false;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const unsigned long long m = __ballot(s < cnt && rank == k);
            const unsigned mg = (unsigned)(m >> g16) & 0xFFFFu;
            top[k] = mg ? (int)__builtin_ctz(mg) : -1;
            tie = tie || (mg & (mg - 1)) != 0;
        }
        exact = __ballot(tie) != 0;
    }
    if (exact) {
        rank = 0;
        quarter_rank_steps<1>(__builtin_bit_cast(int, val), s, cnt, val, rank);      // all 15 other lanes of the row, one rotation each
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const unsigned long long m = __ballot(s < cnt && rank == k);
            const unsigned mg = (unsigned)(m >> g16) & 0xFFFFu;
            top[k] = mg ? (int)__builtin_ctz(mg) : -1;
        }
    }
}

// HR@10 / NDCG@10 of one user from the two top-10 position lists (utils.py:172-184, 190-210).  The ten rank positions are
// worked on by ten lanes side by side -- lane j of the user's lane group (the whole wave, or a quarter of it: QUARTER) looks
// up rating[top_pred[j]], tests the threshold and the membership in top_rating, and divides by log2(j + 1) -- and the
// group's first lane adds the ten terms in numpy's order.  (One lane doing all ten -- nine dependent float64 divisions,
// ten dependent gathers, with the other 63 lanes idle -- was most of eval_users_kernel's time: 114 -> us per series call, r3.)
// Lane 0 of a 16-lane row reads the double of lane N of its row (two DPP row_shl moves; lanes past the row read 0).
template <int N>
__device__ __forceinline__ double row_lane_f64(double v)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)dpp_i<0x100 + N>((int)(unsigned)b), hi = (unsigned)dpp_i<0x100 + N>((int)(unsigned)(b >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template <int N>
__device__ __forceinline__ double wave_lane_f64(double v)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, N), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), N);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// trj: position j of the user's top-10 by RATING on lane j of the group (-1 where there is none).
// own_rating: the rating at the lane's own position of the segment, fetched together with the predictions when the segment
// fits the group (in_regs) -- the rating of a ranked position then comes through the LDS crossbar instead of a third
// dependent memory access.  log2_j = log2_tab[max(j - 1, 0)], idcg = log2_tab[9], fetched before the ranking too.
template <bool QUARTER>
__device__ __forceinline__ void user_metrics(const float *__restrict__ rating, int beg, int cnt, const int (&tp)[10], const int trj,
                                             const float own_rating, const bool in_regs, const double log2_j, const double idcg,
                                             int lane, bool have, int32_t *hits_out, double *ndcg_out)
{
    constexpr int K = 10;
    const int base = QUARTER ? (lane & ~15) : 0;
    const int j = lane - base;
    const int n_top = cnt < K ? cnt : K;
    int tpj = -1;
#pragma unroll
    for (int k = 0; k < K; ++k) tpj = j == k ? tp[k] : tpj;
    const bool ranked = have && j < n_top && tpj >= 0;               // (a position is always found; the guard keeps a bad input from reading out of bounds)
    const float from_regs = __shfl(own_rating, base + (ranked ? tpj : 0), kWave);      // every lane takes part
    double val = 0.0;
    bool hit = false;
    if (ranked) {
        const double rel = (double)(in_regs ? from_regs : rating[beg + tpj]);          // float32 widened (utils.py:132,153)
        hit = rel >= (4.0 / 5.0);                                    // utils.py:175
        bool common = false;                                         // np.in1d(top_rating, top_pred)[j]
#pragma unroll
        for (int q = 0; q < K; ++q) common = common || (q < n_top && trj >= 0 && tp[q] == trj);
        val = (hit && common) ? rel : 0.0;
    }
    const unsigned long long votes = __ballot(hit);
    const int n_hit = __popcll((votes >> base) & 0x3FFull);
    // computeDCG (utils.py:209-210): r[0] + np.sum(r[1:] / log2(2..10)); np.sum of 9 float64 = numpy pairwise: 8 terms
    // combined as a tree, then the 9th added.  The ideal DCG, computeDCG(np.ones(10)), is a constant: the host evaluates
    // it with numpy itself and passes it as log2_tab[9].
    const double term = (j >= 1 && j < K) ? val / log2_j : val;
    // the ten terms to the group's first lane: scalar broadcasts (whole wave) or row shifts (quarter) instead of ten 64-bit
    // ds_bpermute pairs
    double t[K];
    t[0] = term;
    if constexpr (QUARTER) {
        t[1] = row_lane_f64<1>(term); t[2] = row_lane_f64<2>(term); t[3] = row_lane_f64<3>(term);
        t[4] = row_lane_f64<4>(term); t[5] = row_lane_f64<5>(term); t[6] = row_lane_f64<6>(term);
        t[7] = row_lane_f64<7>(term); t[8] = row_lane_f64<8>(term); t[9] = row_lane_f64<9>(term);
    } else {
        t[1] = wave_lane_f64<1>(term); t[2] = wave_lane_f64<2>(term); t[3] = wave_lane_f64<3>(term);
        t[4] = wave_lane_f64<4>(term); t[5] = wave_lane_f64<5>(term); t[6] = wave_lane_f64<6>(term);
        t[7] = wave_lane_f64<7>(term); t[8] = wave_lane_f64<8>(term); t[9] = wave_lane_f64<9>(term);
    }
    if (have && j == 0) {
        const double dcg = t[0] + ((((t[1] + t[2]) + (t[3] + t[4])) + ((t[5] + t[6]) + (t[7] + t[8]))) + t[9]);
        *hits_out = n_hit;
        *ndcg_out = dcg / idcg;
    }
}

template <bool IS_PRED, int R>
__device__ __forceinline__ void multi_items(const float *__restrict__ val, int beg, int cnt, int lane, int (&top)[10])
{
    float v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int t = lane + r * kWave;
        v[r] = t < cnt ? val[beg + t] : 0.f;
        if (IS_PRED) v[r] = nan_last(v[r]);
    }
    top_k_multi<10, R>(v, cnt, lane, top);
}

// Positions of the top-10 of `val` over the segment [beg, beg + cnt) of a wave-per-user segment (any length).
template <bool IS_PRED>
__device__ __forceinline__ void rank_wide(const float *__restrict__ val, int beg, int cnt, int lane, int (&top)[10])
{
    constexpr int K = 10;
    if (cnt <= kWave) {
        // the common case (the items fit one per lane): ten ballots after `cnt` broadcasts, no memory access per round
        float v = lane < cnt ? val[beg + lane] : 0.f;
        if (IS_PRED) v = nan_last(v);
        top_k_in_registers<K, IS_PRED>(v, cnt, lane, top);
    } else if (cnt <= kWave * 2) {
        // heavier users: a few entries per lane (2, or up to 8), loaded once; ten arg-max rounds in registers
        multi_items<IS_PRED, 2>(val, beg, cnt, lane, top);
    } else if (cnt <= kWave * kRegItems) {
        multi_items<IS_PRED, kRegItems>(val, beg, cnt, lane, top);
    } else {
        top_k_positions<K>(val + beg, cnt, lane, top);
    }
}

// The ranking of the RATINGS does not depend on the model: once per test set (ure_eval_rank_ratings).
__global__ __launch_bounds__(kBlock) void eval_rank_ratings_kernel(const int32_t *__restrict__ off, int32_t n_users, const float *__restrict__ rating,
                                                                   int32_t *__restrict__ top_rating)
{
    const int lane = threadIdx.x & 63;
    const int user = (int)(((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6);
    if (user >= n_users) return;
    const int beg = off[user], cnt = off[user + 1] - beg;
    int tr[10];
    rank_wide<false>(rating, beg, cnt, lane, tr);
    if (lane < 10) {
        int v = tr[0];
#pragma unroll
        for (int k = 1; k < 10; ++k) v = lane == k ? tr[k] : v;
        top_rating[(size_t)user * 10 + lane] = v;
    }
}

// Users [0, n_wide): one wavefront each (segments of any length).  Users [n_wide, n_users): segments of at most 16
// entries, four users per wavefront (a quarter wave each) -- on a 10 % hold-out of ml-1m two thirds of the users.
// top_rating (optional): the cached ranking of the ratings, [n_users][10] positions.
__global__ __launch_bounds__(kBlock) void eval_users_kernel(const int32_t *__restrict__ off, int32_t n_users, int32_t n_wide,
                                                            const float *__restrict__ pred, const float *__restrict__ rating,
                                                            const int32_t *__restrict__ top_rating, const double *__restrict__ log2_tab,
                                                            int32_t *__restrict__ hits, double *__restrict__ ndcg, int64_t pred_stride)
{
    constexpr int K = 10;
    const int lane = threadIdx.x & 63;
    const int wave = (int)(((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6);
    // blockIdx.y = member of a series of evaluations that share the test set (ure_eval_series)
    pred += (size_t)blockIdx.y * pred_stride;
    hits += (size_t)blockIdx.y * n_users;
    ndcg += (size_t)blockIdx.y * n_users;
    int tp[K];
    const int jj = (wave < n_wide ? lane : lane & 15) - 1;
    if (wave < n_wide) {
        const int user = wave;
        const int beg = off[user], cnt = off[user + 1] - beg;
        // everything that does not depend on the ranking is requested in front of it
        const double log2_j = log2_tab[jj >= 0 && jj < K - 1 ? jj : 0], idcg = log2_tab[K - 1];
        const bool in_regs = cnt <= kWave;
        const float own_rating = in_regs && lane < cnt ? rating[beg + lane] : 0.f;
        int trj = -1;
        if (top_rating && lane < K) trj = top_rating[(size_t)user * 10 + lane];
        rank_wide<true>(pred, beg, cnt, lane, tp);
        if (!top_rating) {
            int tr[K];
            rank_wide<false>(rating, beg, cnt, lane, tr);
#pragma unroll
            for (int k = 0; k < K; ++k) trj = lane == k ? tr[k] : trj;
        }
        user_metrics<false>(rating, beg, cnt, tp, trj, own_rating, in_regs, log2_j, idcg, lane, true, hits + user, ndcg + user);
        return;
    }
    const int user = n_wide + (wave - n_wide) * 4 + (lane >> 4);
    if (n_wide + (wave - n_wide) * 4 >= n_users) return;
    const bool have = user < n_users;
    const int s = lane & 15;
    const int beg = have ? off[user] : 0, cnt = have ? off[user + 1] - beg : 0;       // cnt <= 16 by the caller's ordering
    const double log2_j = log2_tab[jj >= 0 && jj < K - 1 ? jj : 0], idcg = log2_tab[K - 1];
    const float own_rating = s < cnt ? rating[beg + s] : 0.f;
    int trj = -1;
    if (top_rating && have && s < K) trj = top_rating[(size_t)user * 10 + s];
    const float pv = s < cnt ? nan_last(pred[beg + s]) : 0.f;
    top_k_quarter<K, true>(pv, cnt, lane, tp);
    if (!top_rating) {
        int tr[K];
        top_k_quarter<K>(own_rating, cnt, lane, tr);
#pragma unroll
        for (int k = 0; k < K; ++k) trj = s == k ? tr[k] : trj;
    }
    user_metrics<true>(rating, beg, cnt, tp, trj, own_rating, true, log2_j, idcg, lane, have, hits + (have ? user : 0), ndcg + (have ? user : 0));
}

// ---- The same evaluation as two launches (what ure_eval_users / the series run when the ranking of the ratings is cached) ----
// eval_users_kernel spends most of its VALU time behind the ranking: ten of a wave's 64 lanes (or forty, four users) divide,
// compare and gather for the metrics.  Split: eval_rank_kernel leaves the ten predicted positions of a user PACKED in the
// user's output slots (hits: 32 bits, ndcg: 64 bits), eval_metrics_kernel -- one THREAD per user and member, every lane busy --
// unpacks them, computes HR / NDCG exactly as user_metrics does and overwrites the slots with the results.
// Packing: users [n_wide, n_users) (segments <= 16): 4 bits per position in the ndcg slot; users [0, n_wide) with at most
// kPackMax entries: 9 bits per position, seven in the ndcg slot and three in the hits slot; longer segments (rare) are finished
// by the ranking launch itself, as before, and skipped by the second.
constexpr int kPackMax = 512;

// Sum over the 16 lanes of a DPP row (every lane ends with the total).
__device__ __forceinline__ unsigned row_sum_u32(unsigned v)
{
    v += (unsigned)dpp_i<kDppQuadXor1>((int)v);
    v += (unsigned)dpp_i<kDppQuadXor2>((int)v);
    v += (unsigned)dpp_i<kDppHalfMirror>((int)v);
    v += (unsigned)dpp_i<kDppRowMirror>((int)v);
    return v;
}

// ---- users of 17 .. 32 entries: two per wavefront (a half wave each), round 4.  Lane s of a half holds entry s; its rank is counted
// over its own 16-lane row by DPP rotations and over the half's other row through one cross-row move followed by the same rotations
// (a wavefront each -- `cnt` scalar broadcasts for one user -- was what such users cost before: a fifth of a 10 % hold-out of ml-1m).
template <int N>
__device__ __forceinline__ void half_other_steps_strict(const int ovi, const float val, int &rank)
{
    const int r = N == 0 ? ovi : __builtin_amdgcn_update_dpp(0, ovi, 0x120 + (N == 0 ? 1 : N), 0xF, 0xF, false);
    rank += __builtin_bit_cast(float, r) > val ? 1 : 0;
    if constexpr (N < 15) half_other_steps_strict<N + 1>(ovi, val, rank);
}
template <int N>
__device__ __forceinline__ void half_other_steps(const int ovi, const int os, const int cnt, const float val, const int s, int &rank)
{
    const int rv = N == 0 ? ovi : __builtin_amdgcn_update_dpp(0, ovi, 0x120 + (N == 0 ? 1 : N), 0xF, 0xF, false);
    const int rs = N == 0 ? os : __builtin_amdgcn_update_dpp(0, os, 0x120 + (N == 0 ? 1 : N), 0xF, 0xF, false);
    rank += (rs < cnt && key_gt(__builtin_bit_cast(float, rv), rs, val, s)) ? 1 : 0;
    if constexpr (N < 15) half_other_steps<N + 1>(ovi, os, cnt, val, s, rank);
}
// sum over the 32 lanes of a half wave (every lane ends with the total)
__device__ __forceinline__ unsigned half_sum_u32(unsigned v)
{
    v = row_sum_u32(v);
    return v + (unsigned)__shfl_xor((int)v, 16, kWave);
}

// The ranking half: one member per wave.  What a wave of each class costs was measured on test sets of users with equal
// segment lengths (tools/exp_eval_classes.py, profiles/r03/NOTES.md): the launch sits at the knee of VALU issue (12 cycles per
// entry of an in-register ranking) and of latency times occupancy (two dependent accesses per wave, 8 waves per SIMD).  Several
// members of a series per wave -- bounds fetched once, the M predictions requested together, ranked by one copy of the code --
// lost at every M (2: +22 %, 5: +30 %, 10: +100 %; as unrolled bodies earlier: 69 -> 82 us).  So the instructions are what is
// cut: wave-uniform segment bounds (a scalar loop and scalar ballots instead of exec-masked ones), the ten positions packed on
// the scalar unit; a quarter wave does not search the lane of every rank (ten ballots, each unpacked per row) but lets every
// lane ADD its position into the field of its rank, with a count and a rank sum that expose equal keys.
__global__ __launch_bounds__(kBlock) void eval_rank_kernel(const int32_t *__restrict__ off, int32_t n_users, int32_t n_wide, int32_t n_half,
                                                           const float *__restrict__ pred, const float *__restrict__ rating,
                                                           const int32_t *__restrict__ top_rating, const double *__restrict__ log2_tab,
                                                           int32_t *__restrict__ hits, double *__restrict__ ndcg, int64_t pred_stride)
{
    constexpr int K = 10;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6));
    pred += (size_t)blockIdx.y * pred_stride;
    hits += (size_t)blockIdx.y * n_users;
    ndcg += (size_t)blockIdx.y * n_users;
    if (wave < n_wide) {
        const int user = wave;
        const int beg = __builtin_amdgcn_readfirstlane(off[user]), cnt = __builtin_amdgcn_readfirstlane(off[user + 1]) - beg;
        int tp[K];
        rank_wide<true>(pred, beg, cnt, lane, tp);
        if (cnt > kPackMax) {                                       // too long to pack: finished here
            const int jj = lane - 1;
            const int trj = lane < K ? top_rating[(size_t)user * 10 + lane] : -1;
            user_metrics<false>(rating, beg, cnt, tp, trj, 0.f, false, log2_tab[jj >= 0 && jj < K - 1 ? jj : 0], log2_tab[K - 1], lane, true,
                                hits + user, ndcg + user);
            return;
        }
        unsigned long long lo = 0;
        unsigned hi = 0;
#pragma unroll
        for (int k = 0; k < 7; ++k) lo |= (unsigned long long)(tp[k] & 511) << (9 * k);
#pragma unroll
        for (int k = 7; k < K; ++k) hi |= (unsigned)(tp[k] & 511) << (9 * (k - 7));
        if (lane == 0) {
            hits[user] = (int32_t)hi;
            ndcg[user] = __builtin_bit_cast(double, lo);
        }
        return;
    }
    const int half_waves = (n_half + 1) / 2;
    if (wave < n_wide + half_waves) {
        // ---- two users of at most 32 entries each
        const int first = n_wide + (wave - n_wide) * 2;
        const int user = first + (lane >> 5);
        const bool have = user < n_wide + n_half;
        const int s = lane & 31;
        const int beg = have ? off[user] : 0, cnt = have ? off[user + 1] - beg : 0;       // cnt <= 32 by the caller's ordering
        const float pv = s < cnt ? nan_last(pred[beg + s]) : -__builtin_inff();
        const int n_top = cnt < K ? cnt : K;
        const int pvi = __builtin_bit_cast(int, pv);
        const int ovi = __shfl_xor(pvi, 16, kWave);                                        // the entry at the same place of the half's other row
        int rank = 0;
        quarter_rank_steps_strict<1>(pvi, pv, rank);
        half_other_steps_strict<0>(ovi, pv, rank);
        bool in_top = s < cnt && rank < n_top;
        // positions 0-5 in w0, 6-9 in w1 (5 bits each); chk as below
        unsigned w0 = half_sum_u32(in_top && rank < 6 ? (unsigned)s << (5 * rank) : 0u);
        unsigned w1 = half_sum_u32(in_top && rank >= 6 ? (unsigned)s << (5 * (rank - 6)) : 0u);
        const unsigned chk = half_sum_u32(in_top ? 0x10000u + (unsigned)rank : 0u);
        const bool tie = chk != ((unsigned)n_top << 16) + (unsigned)(n_top * (n_top - 1) / 2);
        if (__builtin_amdgcn_ballot_w64(tie) != 0) {
            rank = 0;
            quarter_rank_steps<1>(pvi, s, cnt, pv, rank);
            half_other_steps<0>(ovi, s ^ 16, cnt, pv, s, rank);
            in_top = s < cnt && rank < n_top;
            w0 = half_sum_u32(in_top && rank < 6 ? (unsigned)s << (5 * rank) : 0u);
            w1 = half_sum_u32(in_top && rank >= 6 ? (unsigned)s << (5 * (rank - 6)) : 0u);
        }
        if (have && s == 0) ndcg[user] = __builtin_bit_cast(double, (unsigned long long)w0 | ((unsigned long long)w1 << 30));
        return;
    }
    const int q0 = n_wide + n_half;                                                       // first user of the quarter class
    const int qwave = wave - n_wide - half_waves;
    const int user = q0 + qwave * 4 + (lane >> 4);
    if (q0 + qwave * 4 >= n_users) return;
    const bool have = user < n_users;
    const int s = lane & 15;
    const int beg = have ? off[user] : 0, cnt = have ? off[user + 1] - beg : 0;       // cnt <= 16 by the caller's ordering
    const float pv = s < cnt ? nan_last(pred[beg + s]) : -__builtin_inff();           // lanes past the segment: greater than nothing
    const int n_top = cnt < K ? cnt : K;
    int rank = 0;
    quarter_rank_steps_strict<1>(__builtin_bit_cast(int, pv), pv, rank);
    // positions 0-4 in w0, 5-9 in w1 (4 bits each); chk = (lanes with one of the first n_top ranks) << 16 | the sum of those ranks
    bool in_top = s < cnt && rank < n_top;
    unsigned w0 = row_sum_u32(in_top && rank < 5 ? (unsigned)s << (4 * rank) : 0u);
    unsigned w1 = row_sum_u32(in_top && rank >= 5 ? (unsigned)s << (4 * (rank - 5)) : 0u);
    const unsigned chk = row_sum_u32(in_top ? 0x10000u + (unsigned)rank : 0u);
    // distinct keys: n_top lanes with the ranks 0 .. n_top - 1.  Equal keys among the first n_top leave a rank out (the sum falls
    // short) or bring more lanes in (the count exceeds n_top): the wave then counts again with the position as the tie-break.
    const bool tie = chk != ((unsigned)n_top << 16) + (unsigned)(n_top * (n_top - 1) / 2);
    if (__builtin_amdgcn_ballot_w64(tie) != 0) {
        rank = 0;
        quarter_rank_steps<1>(__builtin_bit_cast(int, pv), s, cnt, pv, rank);
        in_top = s < cnt && rank < n_top;
        w0 = row_sum_u32(in_top && rank < 5 ? (unsigned)s << (4 * rank) : 0u);
        w1 = row_sum_u32(in_top && rank >= 5 ? (unsigned)s << (4 * (rank - 5)) : 0u);
    }
    if (have && s == 0) ndcg[user] = __builtin_bit_cast(double, (unsigned long long)w0 | ((unsigned long long)w1 << 20));
}

__global__ __launch_bounds__(kBlock) void eval_metrics_kernel(const int32_t *__restrict__ off, int32_t n_users, int32_t n_wide, int32_t n_half,
                                                              const float *__restrict__ rating, const int32_t *__restrict__ top_rating,
                                                              const double *__restrict__ log2_tab, int32_t *hits, double *ndcg)
{
    constexpr int K = 10;
    const int user = (int)blockIdx.x * kBlock + (int)threadIdx.x;
    if (user >= n_users) return;
    hits += (size_t)blockIdx.y * n_users;
    ndcg += (size_t)blockIdx.y * n_users;
    const int beg = off[user], cnt = off[user + 1] - beg;
    if (user < n_wide && cnt > kPackMax) return;                    // finished by eval_rank_kernel
    const unsigned long long lo = __builtin_bit_cast(unsigned long long, ndcg[user]);
    int tp[K], tr[K];
    if (user < n_wide) {
        const unsigned hi = (unsigned)hits[user];
#pragma unroll
        for (int k = 0; k < 7; ++k) tp[k] = (int)((lo >> (9 * k)) & 511);
#pragma unroll
        for (int k = 7; k < K; ++k) tp[k] = (int)((hi >> (9 * (k - 7))) & 511);
    } else if (user < n_wide + n_half) {
#pragma unroll
        for (int k = 0; k < K; ++k) tp[k] = (int)((lo >> (5 * k)) & 31);
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k) tp[k] = (int)((lo >> (4 * k)) & 15);
    }
    const int2 *__restrict__ trp = reinterpret_cast<const int2 *>(top_rating + (size_t)user * 10);     // 40 bytes per user: 8-aligned
#pragma unroll
    for (int k = 0; k < K; k += 2) {
        const int2 v = trp[k / 2];
        tr[k] = v.x;
        tr[k + 1] = v.y;
    }
    const int n_top = cnt < K ? cnt : K;
    double t[K];
    int n_hit = 0;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        double val = 0.0;
        if (j < n_top) {
            const double rel = (double)rating[beg + (tp[j] < cnt ? tp[j] : 0)];       // float32 widened (utils.py:132,153)
            const bool hit = rel >= (4.0 / 5.0);                                       // utils.py:175
            bool common = false;                                                       // np.in1d(top_rating, top_pred)[j]
#pragma unroll
            for (int q = 0; q < K; ++q) common = common || (q < n_top && tr[j] >= 0 && tp[q] == tr[j]);
            n_hit += hit ? 1 : 0;
            val = (hit && common) ? rel : 0.0;
        }
        t[j] = j >= 1 ? val / log2_tab[j - 1] : val;
    }
    // computeDCG (utils.py:209-210) in numpy's pairwise order, as in user_metrics
    const double dcg = t[0] + ((((t[1] + t[2]) + (t[3] + t[4])) + ((t[5] + t[6]) + (t[7] + t[8]))) + t[9]);
    hits[user] = n_hit;
    ndcg[user] = dcg / log2_tab[K - 1];
}

// utils.py:163-184 tail: rmse = sqrt(sse / n_rows), ndcg = mean(ndcg), hr = mean(hits / 10), reduced on
// the device by ONE workgroup in a fixed order (reproducible) so that a caller can queue many
// evaluations and read all results once.
// The sums of a workgroup's 1,024 values per array in the order of the tree  for (o = 512; o > 0; o >>= 1) s[i] += s[i + o]  -- the order every
// reduction of the user metrics has had since round 1, which results are pinned to --: the first four levels through LDS (add(i, j): s[i] += s[j]
// for every array of the reduction, one barrier per level for all of them), the last six inside wavefront 0 with shuffles (tree_tail:
// v + shfl_down(v, o) is s[i] + s[i + o]): six barriers fewer per reduction.
template <class F>
__device__ __forceinline__ void tree_head_1024(F add)
{
    __syncthreads();
#pragma unroll
    for (int o = 512; o >= 64; o >>= 1) {
        if ((int)threadIdx.x < o) add((int)threadIdx.x, (int)threadIdx.x + o);
        __syncthreads();
    }
}
template <typename T>
__device__ __forceinline__ T tree_tail(const T *s)              // (threads 0 .. 63; the total is thread 0's)
{
    T r = s[threadIdx.x & 63];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) r += __shfl_down(r, o, 64);
    return r;
}

__global__ __launch_bounds__(1024) void eval_reduce_kernel(const int32_t *__restrict__ hits, const double *__restrict__ ndcg,
                                                           int32_t n_users, const double *__restrict__ sse, int64_t n_rows,
                                                           double *__restrict__ out3)
{
    // blockIdx.x = member of a series (ure_eval_series); a single evaluation is a series of one
    hits += (size_t)blockIdx.x * n_users;
    ndcg += (size_t)blockIdx.x * n_users;
    sse += (size_t)blockIdx.x * URE_SCORE_PARTIALS;
    out3 += (size_t)blockIdx.x * 3;
    __shared__ double sn[1024];
    __shared__ long long sh[1024];
    double an = 0.0;
    long long ah = 0;
    for (int t = threadIdx.x; t < n_users; t += 1024) {
        an += ndcg[t];
        ah += hits[t];
    }
    __shared__ double ss[1024];
    double as = 0.0;
    for (int t = threadIdx.x; t < URE_SCORE_PARTIALS; t += 1024) as += sse[t];
    sn[threadIdx.x] = an;
    sh[threadIdx.x] = ah;
    ss[threadIdx.x] = as;
    tree_head_1024([&](int i, int j) { sn[i] += sn[j]; sh[i] += sh[j]; ss[i] += ss[j]; });
    if (threadIdx.x >= 64) return;
    const double tn = tree_tail(sn), ts = tree_tail(ss);
    const long long th = tree_tail(sh);
    if (threadIdx.x == 0) {
        out3[0] = sqrt(ts / (double)n_rows);
        out3[1] = n_users > 0 ? tn / (double)n_users : 0.0;
        out3[2] = n_users > 0 ? ((double)th / 10.0) / (double)n_users : 0.0;
    }
}

// The same three numbers for a SUBSET of a test set's users, from the per-pair predictions and per-user metrics a series on the whole set
// left behind (ure_eval_subset).  scratch.py:83-97 tests every epoch's ensemble on the shard's own test set AND on the total test set; the
// reference builds the total set as the shards' test sets side by side (config.py:144-148), so the shard's set is the total set's rows of
// the shard's users: same models, same pairs, same per-user rankings -- a fifth of the evaluation work at five shards was computing them
// twice.  One workgroup per member, reduced in a fixed order.
__global__ __launch_bounds__(1024) void eval_subset_kernel(const int32_t *__restrict__ sub_users, int32_t n_sub, const int32_t *__restrict__ sub_pairs,
                                                           int32_t n_pairs, const float *__restrict__ pred, const float *__restrict__ rating,
                                                           const int32_t *__restrict__ hits, const double *__restrict__ ndcg, int64_t pred_stride,
                                                           int32_t n_users, double *__restrict__ out3)
{
    pred += (size_t)blockIdx.x * pred_stride;
    hits += (size_t)blockIdx.x * n_users;
    ndcg += (size_t)blockIdx.x * n_users;
    out3 += (size_t)blockIdx.x * 3;
    __shared__ double sn[1024], ss[1024];
    __shared__ long long sh[1024], sr[1024];
    double an = 0.0, as = 0.0;
    long long ah = 0;
    const long long ar = threadIdx.x == 0 ? n_pairs : 0;
    for (int t = threadIdx.x; t < n_sub; t += 1024) {
        const int u = sub_users[t];
        an += ndcg[u];
        ah += hits[u];
    }
    // (a thread per pair, not a thread per user walking its pairs: 50 workgroups of dependent loads took 44 us per call)
    constexpr int kPairs = 24;                                     // pairs in flight per thread (a fifth of ml-1m's total test set: 20 per thread, one round); added in index order
    for (int t0 = threadIdx.x; t0 < n_pairs; t0 += kPairs * 1024) {
        float e[kPairs];
#pragma unroll
        for (int k = 0; k < kPairs; ++k) {
            const int t = t0 + k * 1024;
            const int p = t < n_pairs ? sub_pairs[t] : -1;
            e[k] = p >= 0 ? pred[p] - rating[p] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < kPairs; ++k) as += (double)(e[k] * e[k]);
    }
    sn[threadIdx.x] = an; ss[threadIdx.x] = as; sh[threadIdx.x] = ah; sr[threadIdx.x] = ar;
    tree_head_1024([&](int i, int j) { sn[i] += sn[j]; ss[i] += ss[j]; sh[i] += sh[j]; sr[i] += sr[j]; });
    if (threadIdx.x >= 64) return;
    const double tn = tree_tail(sn), ts = tree_tail(ss);
    const long long th = tree_tail(sh), tr = tree_tail(sr);
    if (threadIdx.x == 0) {
        out3[0] = tr > 0 ? sqrt(ts / (double)tr) : 0.0;
        out3[1] = n_sub > 0 ? tn / (double)n_sub : 0.0;
        out3[2] = n_sub > 0 ? ((double)th / 10.0) / (double)n_sub : 0.0;
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void merge_rows_kernel(T *__restrict__ dst, const T *__restrict__ src,
                                                            const int64_t *__restrict__ rows, int64_t n_rows, int width)
{
    const int64_t total = n_rows * width;
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
        const int64_t r = rows[t / width];
        const int c = (int)(t % width);
        dst[r * width + c] = src[r * width + c];
    }
}

template <int LPR>
static void launch_score(const TableList &T, int nm, int nt, int first, int last, const int32_t *uid, const int32_t *iid,
                         const float *rating, int64_t n, float *pred, double *sse, hipStream_t st)
{
    constexpr int G = kWave / LPR;
    const int64_t waves = (n + G - 1) / G;
    const unsigned blocks = (unsigned)std::min<int64_t>((waves + kWavesPerBlock - 1) / kWavesPerBlock, URE_SCORE_PARTIALS);
    hipLaunchKernelGGL(score_kernel<LPR>, dim3(blocks ? blocks : 1), dim3(kBlock), 0, st, T, nm, nt, first, last, uid, iid,
                       rating, n, pred, sse);
}

template <int LPR>
static void launch_score_series(const float *U, const float *V, int64_t su, int64_t sv, int n_series, int nt, const int32_t *uid,
                                const int32_t *iid, const float *rating, int64_t n, const float *base, float *pred, double *sse,
                                hipStream_t st)
{
    constexpr int G = kWave / LPR;
    const int64_t waves = (n + G - 1) / G;
    const unsigned blocks = (unsigned)std::min<int64_t>((waves + kWavesPerBlock - 1) / kWavesPerBlock, URE_SCORE_PARTIALS);
    hipLaunchKernelGGL(score_series_kernel<LPR>, dim3(blocks ? blocks : 1, (unsigned)n_series), dim3(kBlock), 0, st, U, V, su, sv, nt,
                       uid, iid, rating, n, base, pred, sse);
}

}  // namespace ure

using namespace ure;

namespace ure {

// out[j] = the vectors' sum in list order, as score_kernel adds the models of a list (acc = 0; acc += p_0; acc += p_1; ...): the running
// sum over an ensemble's FIXED models from score vectors made once per model (ure_score with one model, first = 1, last = 0).
struct VectorList {
    const float *v[URE_MAX_MODELS_PER_CALL];
};

__global__ __launch_bounds__(kBlock) void sum_vectors_kernel(VectorList L, int n_vec, int first, int64_t n, float *__restrict__ out)
{
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += (int64_t)gridDim.x * kBlock) {
        float acc = first ? 0.f : out[j];
        for (int k = 0; k < n_vec; ++k) acc += L.v[k][j];
        out[j] = acc;
    }
}

}  // namespace ure

extern "C" int ure_sum_vectors(const float *const *vectors, int n_vectors, int64_t n, float *out, void *stream)
{
    using namespace ure;
    URE_ARG(vectors && n_vectors > 0 && n >= 0 && out);
    if (n == 0) return 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const unsigned blocks = (unsigned)std::min<int64_t>((n + kBlock - 1) / kBlock, 8192);
    for (int c0 = 0; c0 < n_vectors; c0 += URE_MAX_MODELS_PER_CALL) {
        VectorList L;
        const int c = std::min(n_vectors - c0, URE_MAX_MODELS_PER_CALL);
        for (int k = 0; k < c; ++k) {
            URE_ARG(vectors[c0 + k]);
            L.v[k] = vectors[c0 + k];
        }
        hipLaunchKernelGGL(sum_vectors_kernel, dim3(blocks), dim3(kBlock), 0, st, L, c, c0 == 0, n, out);
    }
    URE_HIP(hipGetLastError());
    return 0;
}

extern "C" {

int ure_score(const float *const *U_tables, const float *const *V_tables, int n_models, int n_models_total, int first,
              int last, const int32_t *uid, const int32_t *iid, const float *rating, int64_t n, int d, float *pred,
              double *sse, void *stream)
{
    URE_ARG(U_tables && V_tables && n_models > 0 && n_models <= URE_MAX_MODELS_PER_CALL && n_models_total >= n_models);
    URE_ARG(uid && iid && pred && n >= 0 && pow2(d) && d >= 4 && d <= 256 && (!sse || rating));
    if (n == 0) return 0;
    TableList T;
    for (int m = 0; m < n_models; ++m) {
        URE_ARG(U_tables[m] && V_tables[m]);
        T.U[m] = U_tables[m];
        T.V[m] = V_tables[m];
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (d / 4) {
        case 1: launch_score<1>(T, n_models, n_models_total, first, last, uid, iid, rating, n, pred, sse, st); break;
        case 2: launch_score<2>(T, n_models, n_models_total, first, last, uid, iid, rating, n, pred, sse, st); break;
        case 4: launch_score<4>(T, n_models, n_models_total, first, last, uid, iid, rating, n, pred, sse, st); break;
        case 8: launch_score<8>(T, n_models, n_models_total, first, last, uid, iid, rating, n, pred, sse, st); break;
        case 16: launch_score<16>(T, n_models, n_models_total, first, last, uid, iid, rating, n, pred, sse, st); break;
        case 32: launch_score<32>(T, n_models, n_models_total, first, last, uid, iid, rating, n, pred, sse, st); break;
        case 64: launch_score<64>(T, n_models, n_models_total, first, last, uid, iid, rating, n, pred, sse, st); break;
        default: return fail(-1, "ure_score: unsupported d=%d", d);
    }
    URE_HIP(hipGetLastError());
    return 0;
}

// (A wave keeping its user(s) for a run of 5 members of a series -- segment bounds and rating ranks fetched once -- was measured:
// 69 -> 82 us per series call; a fifth of the waves hides less latency than the saved loads cost.  One member per wave.)
static unsigned eval_user_blocks(int32_t n_users, int32_t n_wide, int32_t n_half)
{
    const int64_t waves = (int64_t)n_wide + ((int64_t)n_half + 1) / 2 + ((int64_t)(n_users - n_wide - n_half) + 3) / 4;
    return (unsigned)((waves + kWavesPerBlock - 1) / kWavesPerBlock);
}

// HR / NDCG of every user for n_series members (member m: pred + m * pred_stride, hits / ndcg + m * n_users).
static void launch_eval_users(const int32_t *off, int32_t n_users, int32_t n_wide, int32_t n_half, const float *pred, const float *rating,
                              const int32_t *top_rating, const double *log2_tab, int32_t *hits, double *ndcg, int64_t pred_stride,
                              int n_series, hipStream_t st)
{
    if (!top_rating) {                  // no cached ranking of the ratings: the one-launch form ranks them too (a wavefront for every user of more than 16 entries)
        const dim3 grid(eval_user_blocks(n_users, n_wide + n_half, 0), (unsigned)n_series);
        hipLaunchKernelGGL(eval_users_kernel, grid, dim3(kBlock), 0, st, off, n_users, n_wide + n_half, pred, rating, top_rating, log2_tab, hits, ndcg, pred_stride);
        return;
    }
    const dim3 grid(eval_user_blocks(n_users, n_wide, n_half), (unsigned)n_series);
    hipLaunchKernelGGL(eval_rank_kernel, grid, dim3(kBlock), 0, st, off, n_users, n_wide, n_half, pred, rating, top_rating, log2_tab, hits, ndcg, pred_stride);
    hipLaunchKernelGGL(eval_metrics_kernel, dim3((unsigned)((n_users + kBlock - 1) / kBlock), (unsigned)n_series), dim3(kBlock), 0, st, off, n_users,
                       n_wide, n_half, rating, top_rating, log2_tab, hits, ndcg);
}

int ure_eval_rank_ratings(const int32_t *off, int32_t n_users, const float *rating, int32_t *top_rating, void *stream)
{
    URE_ARG(off && rating && top_rating && n_users >= 0);
    if (n_users == 0) return 0;
    const unsigned blocks = (unsigned)((n_users + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL(eval_rank_ratings_kernel, dim3(blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), off, n_users, rating, top_rating);
    URE_HIP(hipGetLastError());
    return 0;
}

int ure_eval_users(const int32_t *off, int32_t n_users, const float *pred, const float *rating, const double *log2_tab,
                   int32_t *hits, double *ndcg, const int32_t *top_rating, int32_t n_wide, int32_t n_half, void *stream)
{
    URE_ARG(off && pred && rating && log2_tab && hits && ndcg && n_users >= 0 && n_wide >= 0 && n_half >= 0 && n_wide + n_half <= n_users);
    if (n_users == 0) return 0;
    launch_eval_users(off, n_users, n_wide, n_half, pred, rating, top_rating, log2_tab, hits, ndcg, 0, 1, static_cast<hipStream_t>(stream));
    URE_HIP(hipGetLastError());
    return 0;
}

int ure_eval_reduce(const int32_t *hits, const double *ndcg, int32_t n_users, const double *sse, int64_t n_rows, double *out3,
                    void *stream)
{
    URE_ARG(hits && ndcg && sse && out3 && n_users >= 0 && n_rows > 0);
    hipLaunchKernelGGL(eval_reduce_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), hits, ndcg, n_users, sse, n_rows,
                       out3);
    URE_HIP(hipGetLastError());
    return 0;
}

int ure_eval_series(const float *const *U_fixed, const float *const *V_fixed, int n_fixed, const float *U_series,
                    const float *V_series, int64_t stride_u, int64_t stride_v, int n_series, const int32_t *uid, const int32_t *iid,
                    const float *rating, int64_t n, int d, const int32_t *off, int32_t n_users, const double *log2_tab, float *base,
                    float *pred, double *sse, int32_t *hits, double *ndcg, double *out, const int32_t *top_rating, int32_t n_wide, int32_t n_half,
                    void *stream)
{
    URE_ARG(n_wide >= 0 && n_half >= 0 && n_wide + n_half <= n_users);
    URE_ARG(n_fixed >= 0 && (n_fixed == 0 || (base && !U_fixed == !V_fixed)) && U_series && V_series && n_series > 0 && n_series <= 65535);
    URE_ARG(uid && iid && rating && n > 0 && pow2(d) && d >= 4 && d <= 256 && off && n_users >= 0 && log2_tab && pred && sse && hits &&
            ndcg && out);
    hipStream_t st = static_cast<hipStream_t>(stream);
    // the fixed models' running sum, in list order, once for the whole series (U_fixed == NULL: `base` holds it already)
    for (int c0 = 0; U_fixed && c0 < n_fixed; c0 += URE_MAX_MODELS_PER_CALL) {
        const int c = std::min(n_fixed - c0, URE_MAX_MODELS_PER_CALL);
        if (int rc = ure_score(U_fixed + c0, V_fixed + c0, c, n_fixed + 1, c0 == 0, 0, uid, iid, rating, n, d, base, nullptr, stream)) return rc;
    }
    const float *b = n_fixed ? base : nullptr;
    const int nt = n_fixed + 1;
    switch (d / 4) {
        case 1: launch_score_series<1>(U_series, V_series, stride_u, stride_v, n_series, nt, uid, iid, rating, n, b, pred, sse, st); break;
        case 2: launch_score_series<2>(U_series, V_series, stride_u, stride_v, n_series, nt, uid, iid, rating, n, b, pred, sse, st); break;
        case 4: launch_score_series<4>(U_series, V_series, stride_u, stride_v, n_series, nt, uid, iid, rating, n, b, pred, sse, st); break;
        case 8: launch_score_series<8>(U_series, V_series, stride_u, stride_v, n_series, nt, uid, iid, rating, n, b, pred, sse, st); break;
        case 16: launch_score_series<16>(U_series, V_series, stride_u, stride_v, n_series, nt, uid, iid, rating, n, b, pred, sse, st); break;
        case 32: launch_score_series<32>(U_series, V_series, stride_u, stride_v, n_series, nt, uid, iid, rating, n, b, pred, sse, st); break;
        case 64: launch_score_series<64>(U_series, V_series, stride_u, stride_v, n_series, nt, uid, iid, rating, n, b, pred, sse, st); break;
        default: return fail(-1, "ure_eval_series: unsupported d=%d", d);
    }
    if (n_users > 0)
        launch_eval_users(off, n_users, n_wide, n_half, pred, rating, top_rating, log2_tab, hits, ndcg, n, n_series, st);
    hipLaunchKernelGGL(eval_reduce_kernel, dim3((unsigned)n_series), dim3(1024), 0, st, hits, ndcg, n_users, sse, n, out);
    URE_HIP(hipGetLastError());
    return 0;
}

int ure_eval_series_compact(const float *const *U_fixed, const float *const *V_fixed, int n_fixed, const float *snap, int64_t stride,
                            const int32_t *row_slot, const float *U0, const float *V0, const float *snap_a, int32_t n_user_rows,
                            int n_series, const int32_t *uid, const int32_t *iid, const float *rating, int64_t n, int d,
                            const int32_t *off, int32_t n_users, const double *log2_tab, float *base, float *pred, double *sse,
                            int32_t *hits, double *ndcg, double *out, const int32_t *top_rating, int32_t n_wide, int32_t n_half, void *stream)
{
    URE_ARG(snap && row_slot && U0 && V0 && snap_a && n_user_rows > 0 && stride >= 0 && n_series > 0 && n_series <= 65535);
    URE_ARG(uid && iid && n > 0 && pow2(d) && d >= 4 && d <= 256 && pred);
    // the members' own scores, all epochs of a pair together, into pred; then the fixed models, the division, the squared errors
    // and the ranking exactly as ure_eval_series_own does them (in place)
    if (int rc = ure_score_own_compact(snap, stride, row_slot, U0, V0, snap_a, n_user_rows, n_series, uid, iid, n, d, pred, stream)) return rc;
    return ure_eval_series_own(U_fixed, V_fixed, n_fixed, pred, n_series, uid, iid, rating, n, d, off, n_users, log2_tab, base, pred, sse, hits, ndcg,
                               out, top_rating, n_wide, n_half, stream);
}

int ure_score_own_compact(const float *snap, int64_t stride, const int32_t *row_slot, const float *U0, const float *V0, const float *snap_a,
                          int32_t n_user_rows, int n_series, const int32_t *uid, const int32_t *iid, int64_t n, int d, float *own, void *stream)
{
    URE_ARG(snap && row_slot && U0 && V0 && snap_a && n_user_rows > 0 && stride >= 0 && n_series > 0 && n_series <= 65535);
    URE_ARG(uid && iid && own && n > 0 && pow2(d) && d >= 4 && d <= 256);
    hipStream_t st = static_cast<hipStream_t>(stream);
#define URE_OWN(L)                                                                                                                      \
    do {                                                                                                                                \
        constexpr int G = kWave / L;                                                                                                    \
        const int64_t waves = (n + G - 1) / G;                                                                                          \
        const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>((waves + kWavesPerBlock - 1) / kWavesPerBlock, 16384)); \
        hipLaunchKernelGGL(score_own_epochs_kernel<L>, dim3(blocks), dim3(kBlock), 0, st, snap, stride, row_slot, U0, V0, snap_a, n_user_rows,          \
                           n_series, uid, iid, n, own);                                                                                 \
    } while (0)
    switch (d / 4) {
        case 1: URE_OWN(1); break;
        case 2: URE_OWN(2); break;
        case 4: URE_OWN(4); break;
        case 8: URE_OWN(8); break;
        case 16: URE_OWN(16); break;
        case 32: URE_OWN(32); break;
        case 64: URE_OWN(64); break;
        default: return fail(-1, "ure_score_own_compact: unsupported d=%d", d);
    }
#undef URE_OWN
    URE_HIP(hipGetLastError());
    return 0;
}

int ure_eval_series_own(const float *const *U_fixed, const float *const *V_fixed, int n_fixed, const float *own, int n_series, const int32_t *uid,
                        const int32_t *iid, const float *rating, int64_t n, int d, const int32_t *off, int32_t n_users, const double *log2_tab,
                        float *base, float *pred, double *sse, int32_t *hits, double *ndcg, double *out, const int32_t *top_rating, int32_t n_wide,
                        int32_t n_half, void *stream)
{
    URE_ARG(n_wide >= 0 && n_half >= 0 && n_wide + n_half <= n_users);
    URE_ARG(n_fixed >= 0 && (n_fixed == 0 || (base && !U_fixed == !V_fixed)) && own && n_series > 0 && n_series <= 65535);
    URE_ARG(uid && iid && rating && n > 0 && pow2(d) && d >= 4 && d <= 256 && off && n_users >= 0 && log2_tab && pred && sse && hits &&
            ndcg && out);
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int c0 = 0; U_fixed && c0 < n_fixed; c0 += URE_MAX_MODELS_PER_CALL) {
        const int c = std::min(n_fixed - c0, URE_MAX_MODELS_PER_CALL);
        if (int rc = ure_score(U_fixed + c0, V_fixed + c0, c, n_fixed + 1, c0 == 0, 0, uid, iid, rating, n, d, base, nullptr, stream)) return rc;
    }
#define URE_COMBINE(L)                                                                                                                  \
    do {                                                                                                                                \
        constexpr int G = kWave / L;                                                                                                    \
        const int64_t waves = (n + G - 1) / G;                                                                                          \
        const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((waves + kWavesPerBlock - 1) / kWavesPerBlock, URE_SCORE_PARTIALS));   \
        const unsigned phys = (unsigned)(((int64_t)blocks * kWavesPerBlock * G + kBlock - 1) / kBlock);                                 \
        hipLaunchKernelGGL(series_combine_kernel<L>, dim3(phys, (unsigned)n_series), dim3(kBlock), 0, st, own, n_fixed + 1, rating, n,  \
                           n_fixed ? base : nullptr, pred, sse, blocks);                                                                \
    } while (0)
    switch (d / 4) {
        case 1: URE_COMBINE(1); break;
        case 2: URE_COMBINE(2); break;
        case 4: URE_COMBINE(4); break;
        case 8: URE_COMBINE(8); break;
        case 16: URE_COMBINE(16); break;
        case 32: URE_COMBINE(32); break;
        case 64: URE_COMBINE(64); break;
        default: return fail(-1, "ure_eval_series_own: unsupported d=%d", d);
    }
#undef URE_COMBINE
    if (n_users > 0)
        launch_eval_users(off, n_users, n_wide, n_half, pred, rating, top_rating, log2_tab, hits, ndcg, n, n_series, st);
    hipLaunchKernelGGL(eval_reduce_kernel, dim3((unsigned)n_series), dim3(1024), 0, st, hits, ndcg, n_users, sse, n, out);
    URE_HIP(hipGetLastError());
    return 0;
}

int ure_eval_subset(const int32_t *sub_users, int32_t n_sub, const int32_t *sub_pairs, int32_t n_pairs, const float *pred, const float *rating,
                    const int32_t *hits, const double *ndcg, int64_t pred_stride, int32_t n_users, int n_series, double *out, void *stream)
{
    URE_ARG(sub_users && sub_pairs && pred && rating && hits && ndcg && out && n_sub >= 0 && n_pairs >= 0 && n_users >= 0 && n_series > 0 && n_series <= 65535);
    hipLaunchKernelGGL(eval_subset_kernel, dim3((unsigned)n_series), dim3(1024), 0, static_cast<hipStream_t>(stream), sub_users, n_sub, sub_pairs, n_pairs,
                       pred, rating, hits, ndcg, pred_stride, n_users, out);
    URE_HIP(hipGetLastError());
    return 0;
}

int ure_merge_rows(float *dst, const float *src, const int64_t *rows, int64_t n_rows, int d, void *stream)
{
    URE_ARG(dst && src && rows && n_rows >= 0 && d >= 1);
    if (n_rows == 0) return 0;
    const bool wide = d % 4 == 0 && (reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) % 16 == 0;
    const int width = wide ? d / 4 : d;
    const int64_t total = n_rows * width;
    const unsigned blocks = (unsigned)std::min<int64_t>((total + kBlock - 1) / kBlock, 2048);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (wide)
        hipLaunchKernelGGL(merge_rows_kernel<float4>, dim3(blocks), dim3(kBlock), 0, st, reinterpret_cast<float4 *>(dst),
                           reinterpret_cast<const float4 *>(src), rows, n_rows, width);
    else
        hipLaunchKernelGGL(merge_rows_kernel<float>, dim3(blocks), dim3(kBlock), 0, st, dst, src, rows, n_rows, width);
    URE_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
