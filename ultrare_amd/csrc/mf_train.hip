// mf_train.hip -- per-shard MF training for gfx950 (MI355X).
//
// Replaces, for every shard of a job at once, the reference's
//   baseTrain loop            method/utils.py:58-91   (gather, dot, MSE(sum), backward)
//   optim.SGD(...).step()     method/scratch.py:64-69 (dense L2 + momentum over ALL rows)
//   DataLoader batching       read.py:108-133         (batch s = perm[s*B:(s+1)*B])
//
// Design (MI355X-first, not a translation of the autograd graph):
//   * One optimizer step = ONE kernel.  A wavefront owns a destination row (a user
//     row of U or an item row of V).  It walks that row's CSR segment, keeps the
//     entries whose batch number equals this step's, gathers the opposite table's
//     row for each (16 B per lane, d/4 lanes per row, 64/(d/4) entries in flight per
//     wave instruction), recomputes the error e = <u,v> - r and accumulates
//     2e * other_row in registers.  The full gradient row therefore never leaves
//     the wavefront: the SGD-momentum-L2 update is applied immediately and the row
//     is written once into the *other* half of a ping-pong weight pair, so gathers
//     of this step always see step-t weights.  No atomics, no gradient tables, and
//     the result is bitwise reproducible run to run.
//   * Rows with many entries ("heavy", first n_heavy of row_sched) get a whole
//     4-wave workgroup; partial sums meet in LDS in a fixed order.
//   * Batch membership is a 2-byte tag per CSR slot, refreshed once per epoch by
//     assign_batches_kernel from the epoch's permutation.
//   * Shards are independent (sisa.py:33-36), so a job's shards share each launch
//     (blockIdx.y = shard): one tick advances every shard by one optimizer step.
//
// Algorithmic bytes per interaction and step: 16 + 16 d sparse,
// 20 P dense; this kernel moves 16 P dense (no gradient read) + the tag scan.
#include "ure_internal.h"

#include <algorithm>
#include <vector>

namespace ure {

constexpr int kQueue = 128;   // per-wave ring of matched entries (>= 64 + 64/LPR - 1)

struct ure_job {
    std::vector<ure_shard_t> host;
    ure_shard_t *dev = nullptr;
    int64_t ticks = 0;
    int max_blocks = 0;
    int max_n = 0;
    int d = 0;
};

__device__ __forceinline__ int shard_steps(const ure_shard_t &S) { return (S.N + S.batch - 1) / S.batch; }

// Once per epoch and shard: tag every CSR slot with the batch its interaction is
// drawn into.  perm[b] = file-order index of the b-th sample of the epoch.
__global__ __launch_bounds__(kBlock) void assign_batches_kernel(const ure_shard_t *__restrict__ shards, int64_t tick)
{
    const ure_shard_t &S = shards[blockIdx.y];
    const int steps = shard_steps(S);
    if (tick >= (int64_t)steps * S.epochs) return;
    const int epoch = (int)(tick / steps);
    if (tick - (int64_t)epoch * steps != 0) return;
    const int32_t *__restrict__ perm = S.perm + (size_t)epoch * S.N;
    for (int b = blockIdx.x * kBlock + threadIdx.x; b < S.N; b += gridDim.x * kBlock) {
        const int j = perm[b];
        if ((unsigned)j >= (unsigned)S.N) continue;   // malformed permutation: never index outside the shard
        const uint16_t s = (uint16_t)(b / S.batch);
        S.u_b[S.u_pos[j]] = s;
        S.i_b[S.i_pos[j]] = s;
    }
}

template <int LPR>
__global__ __launch_bounds__(kBlock) void mf_step_kernel(const ure_shard_t *__restrict__ shards, int64_t tick)
{
    constexpr int D = LPR * 4;
    constexpr int G = kWave / LPR;           // entries gathered per wave instruction
    __shared__ int q_oid[kWavesPerBlock][kQueue];
    __shared__ float q_r[kWavesPerBlock][kQueue];
    __shared__ float4 part_acc[kWavesPerBlock][LPR];
    __shared__ float part_sse[kWavesPerBlock];

    const ure_shard_t &S = shards[blockIdx.y];
    const int steps = shard_steps(S);
    if (tick >= (int64_t)steps * S.epochs) return;
    const int epoch = (int)(tick / steps);
    const int s = (int)(tick - (int64_t)epoch * steps);
    const int cur = (int)(tick & 1);
    const bool first = tick == 0;
    const int n_rows = S.n_user + S.n_item;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool heavy = (int)blockIdx.x < S.n_heavy;
    const int sched = heavy ? (int)blockIdx.x : S.n_heavy + ((int)blockIdx.x - S.n_heavy) * kWavesPerBlock + wave;
    if (sched >= n_rows) return;             // whole wave (light) or whole block (past the shard's rows)
    const int wpr = heavy ? kWavesPerBlock : 1;
    const int wir = heavy ? wave : 0;

    const int row_id = S.row_sched[sched];
    const bool is_user = row_id < S.n_user;
    const int row = is_user ? row_id : row_id - S.n_user;
    const int32_t *__restrict__ off = is_user ? S.u_off : S.i_off;
    const int32_t *__restrict__ oid = is_user ? S.u_oid : S.i_oid;
    const float *__restrict__ rat = is_user ? S.u_r : S.i_r;
    const uint16_t *__restrict__ tag = is_user ? S.u_b : S.i_b;
    const float *__restrict__ w_cur = is_user ? S.U[cur] : S.V[cur];
    float *__restrict__ w_next = is_user ? S.U[cur ^ 1] : S.V[cur ^ 1];
    float *__restrict__ mom = is_user ? S.mU : S.mV;
    const float *__restrict__ other = is_user ? S.V[cur] : S.U[cur];

    const int sub = lane & (LPR - 1), grp = lane / LPR;
    const size_t row_off = (size_t)row * D + sub * 4;
    const float4 w = *reinterpret_cast<const float4 *>(w_cur + row_off);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float sse = 0.f;

    int *qo = q_oid[wave];
    float *qr = q_r[wave];
    int qh = 0, qt = 0;

    auto round = [&](int head, int tail) {
        const int idx = head + grp;
        const bool act = idx < tail;
        const int slot = idx & (kQueue - 1);
        const int o = act ? qo[slot] : 0;
        const float r = qr[slot];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (act) v = *reinterpret_cast<const float4 *>(other + (size_t)o * D + sub * 4);
        float p = w.x * v.x;
        p = fmaf(w.y, v.y, p);
        p = fmaf(w.z, v.z, p);
        p = fmaf(w.w, v.w, p);
        p = group_sum<LPR>(p);
        const float e = p - r;
        const float ge = act ? 2.0f * e : 0.0f;
        if (act && sub == 0) sse = fmaf(e, e, sse);
        acc.x = fmaf(ge, v.x, acc.x);
        acc.y = fmaf(ge, v.y, acc.y);
        acc.z = fmaf(ge, v.z, acc.z);
        acc.w = fmaf(ge, v.w, acc.w);
    };

    const int beg = off[row], end = off[row + 1];
    for (int base = beg + wir * kWave; base < end; base += kWave * wpr) {
        const int p = base + lane;
        const bool m = p < end && tag[p] == (uint16_t)s;
        const unsigned long long mask = __ballot(m);
        if (mask == 0) continue;
        if (m) {
            const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
            const int slot = (qt + rank) & (kQueue - 1);
            qo[slot] = oid[p];
            qr[slot] = rat[p];
        }
        qt += __popcll(mask);
        __builtin_amdgcn_wave_barrier();
        while (qt - qh >= G) {
            round(qh, qt);
            qh += G;
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (qt > qh) round(qh, qt);

    acc.x = cross_group_sum<LPR>(acc.x);
    acc.y = cross_group_sum<LPR>(acc.y);
    acc.z = cross_group_sum<LPR>(acc.z);
    acc.w = cross_group_sum<LPR>(acc.w);
    sse = wave_sum(sse);

    if (heavy) {   // block-uniform branch: all four waves of a heavy row arrive
        if (lane < LPR) part_acc[wave][lane] = acc;
        if (lane == 0) part_sse[wave] = sse;
        __syncthreads();
        if (wave != 0) return;
        if (lane < LPR) {
            acc = part_acc[0][lane];
#pragma unroll
            for (int k = 1; k < kWavesPerBlock; ++k) {
                const float4 t = part_acc[k][lane];
                acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
            }
        }
        sse = (part_sse[0] + part_sse[1]) + (part_sse[2] + part_sse[3]);
    }

    if (lane < LPR) {
        // torch.optim.SGD single-tensor path: g = g + lam*w ; buf = mu*buf + g (buf = g on
        // the first step) ; w = w - lr*buf
        const float lam = S.lam, mu = S.mu, lr = S.lr[epoch];
        float4 m4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!first) m4 = *reinterpret_cast<const float4 *>(mom + row_off);
        float4 g, wn;
        g.x = fmaf(lam, w.x, acc.x); g.y = fmaf(lam, w.y, acc.y);
        g.z = fmaf(lam, w.z, acc.z); g.w = fmaf(lam, w.w, acc.w);
        if (!first) {
            g.x = __fadd_rn(__fmul_rn(mu, m4.x), g.x); g.y = __fadd_rn(__fmul_rn(mu, m4.y), g.y);
            g.z = __fadd_rn(__fmul_rn(mu, m4.z), g.z); g.w = __fadd_rn(__fmul_rn(mu, m4.w), g.w);
        }
        wn.x = fmaf(-lr, g.x, w.x); wn.y = fmaf(-lr, g.y, w.y);
        wn.z = fmaf(-lr, g.z, w.z); wn.w = fmaf(-lr, g.w, w.w);
        *reinterpret_cast<float4 *>(mom + row_off) = g;
        *reinterpret_cast<float4 *>(w_next + row_off) = wn;
    }
    if (is_user && lane == 0 && sse != 0.f) atomicAdd(&S.sse[epoch], (double)sse);
}

template <int LPR>
static void launch_step(const ure_job *job, int64_t tick, hipStream_t st)
{
    dim3 grid(job->max_blocks, (unsigned)job->host.size());
    hipLaunchKernelGGL(mf_step_kernel<LPR>, grid, dim3(kBlock), 0, st, job->dev, tick);
}

}  // namespace ure

using namespace ure;

extern "C" {

int ure_job_create(const ure_shard_t *shards, int n_shards, ure_job_t **out)
{
    URE_ARG(shards && out && n_shards > 0 && n_shards <= 65535);
    auto *job = new ure::ure_job();
    job->host.assign(shards, shards + n_shards);
    for (int k = 0; k < n_shards; ++k) {
        const ure_shard_t &S = shards[k];
        const bool ok = S.N > 0 && S.n_user > 0 && S.n_item > 0 && S.batch > 0 && S.epochs > 0 && pow2(S.d) && S.d >= 4 &&
                        S.d <= 256 && S.n_heavy >= 0 && S.n_heavy <= S.n_user + S.n_item && S.u_off && S.u_oid && S.u_r &&
                        S.u_b && S.u_pos && S.i_off && S.i_oid && S.i_r && S.i_b && S.i_pos && S.row_sched && S.U[0] &&
                        S.U[1] && S.V[0] && S.V[1] && S.mU && S.mV && S.perm && S.lr && S.sse;
        if (!ok) { delete job; return fail(-1, "ure_job_create: shard %d has an invalid descriptor", k); }
        if (S.d != shards[0].d) { delete job; return fail(-1, "ure_job_create: all shards of a job share d"); }
        const int64_t steps = ((int64_t)S.N + S.batch - 1) / S.batch;
        if (steps > 65535) { delete job; return fail(-1, "ure_job_create: shard %d needs %lld steps/epoch (> 65535)", k, (long long)steps); }
        job->ticks = std::max(job->ticks, steps * S.epochs);
        const int n_rows = S.n_user + S.n_item;
        job->max_blocks = std::max(job->max_blocks, S.n_heavy + (n_rows - S.n_heavy + kWavesPerBlock - 1) / kWavesPerBlock);
        job->max_n = std::max(job->max_n, S.N);
    }
    job->d = shards[0].d;
    hipError_t e = hipMalloc(&job->dev, sizeof(ure_shard_t) * n_shards);
    if (e == hipSuccess) e = hipMemcpy(job->dev, shards, sizeof(ure_shard_t) * n_shards, hipMemcpyHostToDevice);
    if (e != hipSuccess) { if (job->dev) (void)hipFree(job->dev); delete job; return fail((int)e, "ure_job_create: %s", hipGetErrorString(e)); }
    *out = reinterpret_cast<ure_job_t *>(job);
    return 0;
}

int ure_job_destroy(ure_job_t *j)
{
    auto *job = reinterpret_cast<ure::ure_job *>(j);
    if (!job) return 0;
    if (job->dev) (void)hipFree(job->dev);
    delete job;
    return 0;
}

int64_t ure_job_shard_steps(const ure_job_t *j, int s)
{
    auto *job = reinterpret_cast<const ure::ure_job *>(j);
    if (!job || s < 0 || s >= (int)job->host.size()) return -1;
    const ure_shard_t &S = job->host[s];
    return (((int64_t)S.N + S.batch - 1) / S.batch) * S.epochs;
}

int64_t ure_job_ticks(const ure_job_t *j)
{
    auto *job = reinterpret_cast<const ure::ure_job *>(j);
    return job ? job->ticks : -1;
}

static int train_ticks(ure::ure_job *job, int64_t tick0, int64_t tick1, hipStream_t st, std::vector<hipEvent_t> *step_ev,
                       std::vector<hipEvent_t> *assign_ev)
{
    const unsigned n_shards = (unsigned)job->host.size();
    const unsigned assign_blocks = (unsigned)std::min((job->max_n + kBlock - 1) / kBlock, 2048);
    auto mark = [&](std::vector<hipEvent_t> *v) -> int {
        if (!v) return 0;
        hipEvent_t e;
        URE_HIP(hipEventCreate(&e));
        v->push_back(e);
        URE_HIP(hipEventRecord(e, st));
        return 0;
    };
    for (int64_t t = tick0; t < tick1; ++t) {
        bool epoch_start = false;
        for (const ure_shard_t &S : job->host) {
            const int64_t steps = ((int64_t)S.N + S.batch - 1) / S.batch;
            if (t < steps * S.epochs && t % steps == 0) { epoch_start = true; break; }
        }
        if (epoch_start) {
            if (int rc = mark(assign_ev)) return rc;
            hipLaunchKernelGGL(assign_batches_kernel, dim3(assign_blocks, n_shards), dim3(kBlock), 0, st, job->dev, t);
            if (int rc = mark(assign_ev)) return rc;
        }
        if (int rc = mark(step_ev)) return rc;
        switch (job->d / 4) {
            case 1: launch_step<1>(job, t, st); break;
            case 2: launch_step<2>(job, t, st); break;
            case 4: launch_step<4>(job, t, st); break;
            case 8: launch_step<8>(job, t, st); break;
            case 16: launch_step<16>(job, t, st); break;
            case 32: launch_step<32>(job, t, st); break;
            case 64: launch_step<64>(job, t, st); break;
            default: return fail(-1, "ure_job_train: unsupported d=%d", job->d);
        }
        if (int rc = mark(step_ev)) return rc;
    }
    URE_HIP(hipGetLastError());
    return 0;
}

int ure_job_train(ure_job_t *j, int64_t tick0, int64_t tick1, void *stream)
{
    auto *job = reinterpret_cast<ure::ure_job *>(j);
    URE_ARG(job && tick0 >= 0 && tick1 >= tick0);
    return train_ticks(job, tick0, std::min(tick1, job->ticks), static_cast<hipStream_t>(stream), nullptr, nullptr);
}

int ure_job_train_profiled(ure_job_t *j, int64_t tick0, int64_t tick1, void *stream, double *step_ms, int64_t *n_step,
                           double *assign_ms, int64_t *n_assign)
{
    auto *job = reinterpret_cast<ure::ure_job *>(j);
    URE_ARG(job && tick0 >= 0 && tick1 >= tick0 && step_ms && n_step && assign_ms && n_assign);
    hipStream_t st = static_cast<hipStream_t>(stream);
    std::vector<hipEvent_t> se, ae;
    int rc = train_ticks(job, tick0, std::min(tick1, job->ticks), st, &se, &ae);
    if (rc == 0) {
        hipError_t e = hipStreamSynchronize(st);
        if (e != hipSuccess) rc = fail((int)e, "ure_job_train_profiled: %s", hipGetErrorString(e));
    }
    *step_ms = *assign_ms = 0.0;
    *n_step = (int64_t)se.size() / 2;
    *n_assign = (int64_t)ae.size() / 2;
    for (int which = 0; which < 2 && rc == 0; ++which) {
        auto &v = which ? ae : se;
        for (size_t q = 0; q + 1 < v.size(); q += 2) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, v[q], v[q + 1]) == hipSuccess) (which ? *assign_ms : *step_ms) += ms;
        }
    }
    for (hipEvent_t e : se) (void)hipEventDestroy(e);
    for (hipEvent_t e : ae) (void)hipEventDestroy(e);
    return rc;
}

}  // extern "C"
