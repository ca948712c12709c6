// mf_train.hip -- per-shard MF training for gfx950 (MI355X).
//
// Replaces, for every shard of a job at once, the reference's
//   baseTrain loop            method/utils.py:58-91   (gather, dot, MSE(sum), backward)
//   optim.SGD(...).step()     method/scratch.py:64-69 (dense L2 + momentum over ALL rows)
//   DataLoader batching       read.py:108-133         (batch s = perm[s*B:(s+1)*B])
//
// Design (MI355X-first, not a translation of the autograd graph):
//   * One optimizer step = ONE kernel, owner-computes.  Whoever owns a destination row (a user
//     row of U or an item row of V) walks that row's segment of slots, keeps the entries whose
//     batch number equals this step's, gathers the opposite table's row for each (one or two
//     16-byte pieces per lane, LPR = lanes_per_row(d) lanes per row, 64/LPR rows per wave
//     instruction), recomputes the error e = <u,v> - r and accumulates 2e * other_row in
//     registers.  The gradient row never leaves the owner: the SGD-momentum-L2 update is applied
//     immediately and the row is written once into the *other* half of a ping-pong weight pair,
//     so gathers of this step always see step-t weights.  No atomics, no gradient tables, and
//     the result is bitwise reproducible run to run.
//   * The unit of work is one scan pass, not a row: a LANE GROUP (LPR lanes) walks one work unit =
//     8*LPR consecutive slots of a row (ure_host_build_units).  A row cut into several units has
//     them in one workgroup; their partial sums meet in LDS and the row's first unit adds them in
//     unit order.  Every workgroup therefore does the same amount of work -- with row-sized work
//     the heaviest rows set the length of a launch (profiles/r01/NOTES.md).  Rows the shard never
//     touches only decay: they are advanced in closed form when the tables are read (lazy_rows) or
//     updated 256/LPR per workgroup.
//   * The unit loop is branch free: eight unconditional queue writes per lane (matches first, the
//     rest behind), queue reads and row gathers issued back to back, lane-group sums by DPP.
//   * Row segments live in one slot array in schedule order, 8-aligned and padded, so
//     a lane scans 8 slots with one 16-byte load per array.
//   * Batch membership is a 2-byte tag per slot, double-buffered by epoch parity; the steps of an
//     epoch carry the preparation of the next epoch's tags as extra workgroups at the end of the
//     grid (tag_prep.h).
//   * Shards are independent (sisa.py:33-36), so a job's shards share each launch: one tick advances
//     every shard by one optimizer step.  Workgroups are dealt out to (shard, workgroup) so that an XCD
//     works on at most two shards (sliced mapping, see mf_step).
//
// Algorithmic bytes per interaction and step: 16 + 16 d sparse,
// 20 P dense; this kernel moves 16 P dense (no gradient read) + the tag scan.
#include "tag_prep.h"

#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

// Tuning knobs (defaults = the swept optimum; -D overrides are for sweeps only)
#ifndef URE_KGB_NARROW
#define URE_KGB_NARROW 5      // rows a lane group gathers together, d = 32
#endif
// d <= 16 (LPR <= 4): more lane groups per wavefront, each with its own queue positions and gather addresses live at once --
// the d = 32 optimum (5 rows in flight under a 7-wave register budget) leaves the allocator 72 VGPRs where these need ~88,
// and it spilled 15-16 of them (13 scratch loads + 13 stores per launch path, 16.5 us per launch at d = 16, round 2).
// Per-width knobs: URE_KGB_D16 / URE_WAVES_D16 for d = 16, URE_KGB_D8 / URE_WAVES_D8 for d <= 8.
#ifndef URE_KGB_D16
#define URE_KGB_D16 5
#endif
#ifndef URE_WAVES_D16
#define URE_WAVES_D16 5
#endif
#ifndef URE_KGB_D8
#define URE_KGB_D8 5
#endif
#ifndef URE_WAVES_D8
#define URE_WAVES_D8 5
#endif
#ifndef URE_KGB_WIDE
#define URE_KGB_WIDE 4        // the same for d >= 64 (two float4 per lane)
#endif
#ifndef URE_WAVES_NARROW
#define URE_WAVES_NARROW 7    // waves per SIMD the register allocator must leave room for, d <= 32
#endif
#ifndef URE_WAVES_WIDE
#define URE_WAVES_WIDE 4
#endif
#ifndef URE_TOUCH_WAVES
#define URE_TOUCH_WAVES 4     // touch mode (mf_touch.h)
#endif
#ifndef URE_TOUCH_KGB
#define URE_TOUCH_KGB 4       // rows a lane group gathers together in touch mode
#endif

namespace ure {

constexpr int kQueue = 512;     // per-wave match queues: 64/LPR private queues of 8*LPR entries
constexpr int kSegPerLane = 8;  // slots one lane scans per pass over its unit
// Lanes that share a table row.  Up to d = 32 every lane holds one float4; wider rows give every
// lane two (d = 64: 8 lanes, 128: 16, 256: 32), which doubles the rows -- and the bytes in flight --
// per wavefront.  Measured (us per launch): d = 64, 8 shards: 29.3 -> 26.2; d = 128, 25 M workload:
// 1327 -> 1206; four pieces per lane at d = 128: 1530; two pieces at d = 32: 24.9 vs 18.6.
__host__ __device__ constexpr int lanes_per_row(int d) { return d <= URE_NARROW_MAX ? d / 4 : d / 8; }

// per-instantiation tuning of the step kernel: rows gathered together, and the occupancy the register allocator must
// leave room for (tools/isa_report.py + tests/test_cpu_host.py: no instantiation may spill)
__host__ __device__ constexpr int step_kgb(int lpr, int v4) { return v4 > 1 ? URE_KGB_WIDE : lpr >= 8 ? URE_KGB_NARROW : lpr == 4 ? URE_KGB_D16 : URE_KGB_D8; }
__host__ __device__ constexpr int step_waves(int lpr, int v4) { return v4 > 1 ? URE_WAVES_WIDE : lpr >= 8 ? URE_WAVES_NARROW : lpr == 4 ? URE_WAVES_D16 : URE_WAVES_D8; }

__device__ __forceinline__ int shard_steps(const ure_shard_t &S) { return (S.N + S.batch - 1) / S.batch; }

#ifdef URE_TIMELINE
// Diagnostic build (python -m ultrare_amd.build --timeline, tools/exp_timeline.py): every workgroup
// records when it started and ended, and its first lane when it passed the phases of the unit path
// (100 MHz wall clock; 8 values per workgroup), so that the inside of a launch can be read.
__device__ long long *g_timeline = nullptr;
#define URE_STAMP(i)                                                                                          \
    do {                                                                                                      \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                           \
        const size_t lin_ = (size_t)blockIdx.y * gridDim.x + blockIdx.x;                                      \
        if (g_timeline && threadIdx.x == 0 && lin_ < 16384) g_timeline[((size_t)(tick & 15) * 16384 + lin_) * 8 + (i)] = wall_clock64(); \
    } while (0)
#else
#define URE_STAMP(i)
#endif

template <int LPR, int V4>
__device__ __forceinline__ void mf_step(const ure_shard_t *__restrict__ shards, const shard_aux *__restrict__ aux, int64_t tick, int shard_fast)
{
    constexpr int D = LPR * V4 * 4;          // row width: LPR lanes x V4 float4 per lane
    using Row = RowVec<V4>;
    constexpr int G = kWave / LPR;           // lane groups per wavefront = table rows one instruction gathers
    constexpr int UPB = kBlock / LPR;        // lane groups (work units) per workgroup
    constexpr int CAP = kSegPerLane * LPR;   // slots a lane group scans per pass = capacity of its match queue
    // table rows a lane group gathers together on the group path: six while rows are narrow (swept on
    // hardware: 18.5 us vs 18.9 at four, 19.7 at eight for d = 32); four for wide rows, where the
    // extra registers cost occupancy (d = 128: 1.55 ms vs 1.34 ms per launch of the 25 M workload)
    constexpr int kGB = step_kgb(LPR, V4);
    // one raw LDS block: the row paths use it as match queues, the tag riders overlay their own
    // arrays on it (tag_prep.h)
    constexpr int kQueueBytes = kWavesPerBlock * kQueue * 8;
    static_assert(kTagLds <= kQueueBytes, "tag phases must fit in the queue space");
    __shared__ __attribute__((aligned(16))) char lds_raw[kQueueBytes];
    __shared__ float4 part_acc[UPB][V4][LPR];
    int (*q_oid)[kQueue] = reinterpret_cast<int (*)[kQueue]>(lds_raw);
    float (*q_r)[kQueue] = reinterpret_cast<float (*)[kQueue]>(lds_raw + kWavesPerBlock * kQueue * 4);

    // Workgroup -> (shard, workgroup of the shard).  The dispatcher deals consecutive workgroup ids out
    // to the 8 XCDs round robin, and every XCD has its own L2, which starts each launch cold for the
    // rewritten tables: the fewer shards an XCD works on, the fewer times a gathered row is fetched.
    // Default (URE_SHARD_FAST=2): the sliced mapping below -- at most two shards per XCD for ANY
    // shard count (bench, 5 shards: 14.5 -> 13.0 us per launch).  URE_SHARD_FAST=1: grid = (shards,
    // workgroups), shard k on XCD k mod 8 when the count is a multiple of 8; 0: grid = (workgroups, shards).
    // A speed matter only: nothing depends on where a workgroup runs.
    int shard_idx = (int)(shard_fast ? blockIdx.x : blockIdx.y);
    int wg = (int)(shard_fast ? blockIdx.y : blockIdx.x);
    if (shard_fast >> 1) {
        // sliced mapping (1-D grid): the 8 S slices (shard k, workgroups = r mod 8) are dealt out S per XCD
        // in shard order, so an XCD's L2 sees at most two shards' tables for any shard count, and every
        // XCD gets the same number of workgroups of every weight class
        const unsigned n_sh = (unsigned)shard_fast >> 8;
        const unsigned x = blockIdx.x & 7u, j = blockIdx.x >> 3;
        const unsigned jq = j / n_sh, jr = j - jq * n_sh;
        const unsigned slice = n_sh * x + jr;
        shard_idx = (int)(slice >> 3);
        wg = (int)(jq * 8 + (slice & 7u));
    }
    const ure_shard_t &S = shards[shard_idx];
    const shard_aux &A = aux[shard_idx];
    // One round of scalar loads for everything the unit path reads from the descriptor: left alone,
    // the compiler loads each field in the basic block that first uses it -- eight dependent rounds of
    // s_load + s_waitcnt in the prologue of every workgroup.  The empty asm statement only pins the
    // values (their loads) here, ahead of the first branch.
    {
        const int p_epochs = S.epochs, p_nu = S.n_user, p_ni = S.n_item, p_units = S.n_units, p_act = S.n_active, p_lazy = S.lazy_rows;
        const int64_t p_slots = S.n_slots;
        const float p_lam = S.lam, p_mu = S.mu;
        asm volatile("" ::"s"(p_epochs), "s"(p_nu), "s"(p_ni), "s"(p_units), "s"(p_act), "s"(p_lazy), "s"(p_slots), "s"(p_lam), "s"(p_mu),
                     "s"(S.lr), "s"(S.sched), "s"(S.units), "s"(S.ent_oid), "s"(S.ent_r), "s"(S.ent_tag), "s"(S.U[0]), "s"(S.U[1]),
                     "s"(S.V[0]), "s"(S.V[1]), "s"(S.mU), "s"(S.mV), "s"(S.sse), "s"(A.steps), "s"(A.inv_steps), "s"(A.ride_m),
                     "s"(A.ride_ab), "s"(A.ride_c), "s"(A.ranges), "s"(A.derive_blocks));
    }
    const int steps = A.steps;
    if (tick >= (int64_t)steps * S.epochs) return;
    const int epoch = (int)epoch_of(A, tick);
    const int s = (int)(tick - (int64_t)epoch * steps);
    const int cur = (int)(tick & 1);
    const bool first = tick == 0;
    const int n_rows = S.n_user + S.n_item;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int sub = lane & (LPR - 1), grp = lane / LPR;
    const float lam = S.lam, mu = S.mu, lr = ldg(S.lr + epoch);
    const int32_t *__restrict__ ent_oid = S.ent_oid;
    const float *__restrict__ ent_r = S.ent_r;
    const uint16_t *__restrict__ ent_tag = S.ent_tag + (size_t)(epoch & 1) * S.n_slots;
    int *qo = q_oid[wave];
    float *qr = q_r[wave];

    // torch.optim.SGD single-tensor path: g = g + lam*w ; buf = mu*buf + g (buf = g on the
    // first step) ; w = w - lr*buf.   Writes buf and the step-(t+1) weights of one row slice.
    auto sgd_update = [&](const Row &wr, const Row &mr, const Row &ar, float *mom_row, float *next_row, float *snap_row) {
        Row gr, nr;
#pragma unroll
        for (int i = 0; i < V4; ++i) {
            const float4 w = wr.q[i], m4 = mr.q[i], acc = ar.q[i];
            float4 g, wn;
            g.x = fmaf(lam, w.x, acc.x); g.y = fmaf(lam, w.y, acc.y);
            g.z = fmaf(lam, w.z, acc.z); g.w = fmaf(lam, w.w, acc.w);
            if (!first) {
                g.x = __fadd_rn(__fmul_rn(mu, m4.x), g.x); g.y = __fadd_rn(__fmul_rn(mu, m4.y), g.y);
                g.z = __fadd_rn(__fmul_rn(mu, m4.z), g.z); g.w = __fadd_rn(__fmul_rn(mu, m4.w), g.w);
            }
            wn.x = fmaf(-lr, g.x, w.x); wn.y = fmaf(-lr, g.y, w.y);
            wn.z = fmaf(-lr, g.z, w.z); wn.w = fmaf(-lr, g.w, w.w);
            gr.q[i] = g;
            nr.q[i] = wn;
        }
        row_store<LPR, V4>(mom_row, sub, gr);
        row_store<LPR, V4>(next_row, sub, nr);
        if (snap_row) row_store<LPR, V4>(snap_row, sub, nr);
    };

    // Workgroup ranges of a shard:
    //   [0, nbU)          work units: one lane group per unit, kBlock / LPR units per workgroup
    //   [nbU, nbU + nbD)  rows without interactions in this shard: decay only (unless lazy_rows)
    //   then              the tag riders
    const int nbU = S.n_units / UPB;
    // lazy_rows: rows [n_active, n_rows) are never gathered by anyone and evolve linearly
    // (w, m)_t = A_t (w, m)_{t-1}; they are then advanced in closed form only when the tables are
    // read (materialize_rows_kernel) instead of being streamed through HBM every step
    const int nbD = S.lazy_rows ? 0 : (n_rows - S.n_active + UPB - 1) / UPB;
    // While epoch e trains, its steps carry the three phases of epoch e+1's batch tags as extra
    // workgroups at the end of the grid (tag_prep.h); the launch boundary between steps orders
    // the phases.
    const TagRide ride = tag_ride(A, s, epoch + 1 < S.epochs);
    const int nbR = ride.count;
    const int nbRows = nbU + nbD;
    const int blk = wg;
    if (blk >= nbRows) {        // the riders come LAST: the row chains start first, the short rider
        const int rb = blk - nbRows;   // workgroups fill the tail of the launch
        if (rb >= nbR) return;
        if (ride.phase == 0) tag_partition(S, epoch + 1, ride.first + rb, lds_raw);
        else if (ride.phase == 1) tag_collect(S, ride.first + rb, lds_raw);
        else tag_derive(S, epoch + 1, ride.first + rb, A.derive_blocks);
        return;
    }
    const int local = wave * G + grp;            // this lane group's unit inside the workgroup
    const bool dense_only = blk >= nbU;
    int4 du = make_int4(-1, 0, 0, local | (1 << 16));
    if (dense_only) {
        const int idx = S.n_active + (blk - nbU) * UPB + local;
        if (idx < n_rows) du.x = ldg(S.sched + 4 * (size_t)idx);
    } else {
        du = ldg_i4(S.units + 4 * ((size_t)blk * UPB + local));
    }
    const bool have = du.x >= 0;
    const int leader = du.w & 0xFFFF, count = (du.w >> 16) & 0x3FFF;
    const bool multi = (du.w >> 30) & 1;         // workgroup-uniform: some row here has several units
    const bool owner = have && local == leader;  // the unit that applies the row's update
    const bool is_user = du.x < S.n_user;
    const int row = is_user ? du.x : du.x - S.n_user;
    const size_t row_off = (size_t)(have ? row : 0) * D;
    Row w = row_zero<V4>(), m4 = w, acc = w;
    if (have) {
        w = row_load<LPR, V4>((is_user ? S.U[cur] : S.V[cur]) + row_off, sub);
    }
    // the momentum row is requested together with the weights: its latency hides behind the scan and the gathers
    // instead of forming a memory level of its own in front of the update (bench: 12.6 -> 12.5 us per launch)
    if (owner && !first) m4 = row_load<LPR, V4>((is_user ? S.mU : S.mV) + row_off, sub);
    // compact end-of-epoch snapshots written by the owners themselves (every active row is rewritten in every step, so the
    // last step of an epoch writes them all): the row's place in the snapshot is requested here, with the row
    const bool snap_here = S.snap != nullptr && S.row_slot != nullptr && s == steps - 1;
    int snap_slot = -1;
    if (snap_here && owner) snap_slot = ldg(S.row_slot + du.x);
    float sse = 0.f;
    URE_STAMP(2);     // unit descriptor and own row arrived
    if (!dense_only) {
        const float *__restrict__ other = is_user ? S.V[cur] : S.U[cur];
        // the group's queue; positions are rotated by the group number so that the G groups of a wave,
        // which write (and read) the same position at the same time, fall into different LDS banks
        int *gq = qo + grp * CAP;
        float *gr = qr + grp * CAP;
        const int beg = du.y, end = have ? du.z : 0;
        for (int seg = beg;; seg += CAP) {
            if (!__any(seg < end)) break;
            // every lane scans 8 consecutive slots of its group's unit (segments are 8-aligned)
            const int p0 = seg + sub * kSegPerLane;
            const bool valid = p0 < end;
            uint4 t4 = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
            int4 o0 = make_int4(0, 0, 0, 0), o1 = o0;
            float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0;
            if (valid) {
                t4 = ldg_u4(ent_tag + p0);
                o0 = ldg_i4(ent_oid + p0);
                o1 = ldg_i4(ent_oid + p0 + 4);
                r0 = ldg_f4(ent_r + p0);
                r1 = ldg_f4(ent_r + p0 + 4);
            }
            const unsigned tw[4] = {t4.x, t4.y, t4.z, t4.w};
            const int ov[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
            const float rv[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
            unsigned mb = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned tg = (tw[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu;
                mb |= (tg == (unsigned)s ? 1u : 0u) << k;
            }
            // position of this lane's matches in its group's queue: CSR order = lane-major
            const int c = __popc(mb);
            const int inc = group_scan<LPR>(c, sub);
            const int qn = (int)group_sum<LPR>((float)c);  // matches of the whole group (<= 256: exact)
            // branch-free compaction: the lane's matches go to the front of the group's queue in slot
            // order, everything else behind the qn matches -- each of the group's CAP slots gets its
            // own queue position, so all eight writes are unconditional
            int mpos = inc - c, upos = qn + sub * kSegPerLane - mpos;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const bool hit = (mb >> k) & 1u;
                const int pos = ((hit ? mpos : upos) + grp) & (CAP - 1);   // rotated by the group: see gq
                gq[pos] = ov[k];
                gr[pos] = rv[k];
                mpos += hit ? 1 : 0;
                upos += hit ? 0 : 1;
            }
            __builtin_amdgcn_wave_barrier();
            if (seg == beg) URE_STAMP(3);     // first pass scanned and compacted
            // every group walks its own queue in order, kGB gathers in flight
            for (int t0 = 0; __any(t0 < qn); t0 += kGB) {
                int o[kGB];
                float r[kGB];
                bool act[kGB];
                Row v[kGB];
                // branch-free on purpose: all queue reads, then all row gathers, are issued back to
                // back (a conditional read or gather costs an exec-mask branch and a full wait each);
                // entries beyond the group's count read a stale queue slot and gather row 0, and
                // contribute nothing (ge = 0)
#pragma unroll
                for (int k = 0; k < kGB; ++k) {
                    const int qi = (min(t0 + k, CAP - 1) + grp) & (CAP - 1);
                    act[k] = t0 + k < qn;
                    o[k] = gq[qi];
                    r[k] = gr[qi];
                }
#pragma unroll
                for (int k = 0; k < kGB; ++k) v[k] = row_load<LPR, V4>(other + (size_t)(act[k] ? o[k] : 0) * D, sub);
#pragma unroll
                for (int k = 0; k < kGB; ++k) {
                    const float p = group_sum<LPR>(row_dot<V4>(w, v[k]));
                    const float e = p - r[k];
                    const float ge = act[k] ? 2.0f * e : 0.0f;
                    if (act[k]) sse = fmaf(e, e, sse);
                    row_axpy<V4>(acc, ge, v[k]);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        URE_STAMP(4);     // all gathers done
        // a row cut into several units: the partial gradient sums meet in LDS and the row's first
        // unit adds them up in unit order (a fixed order: the result does not depend on timing)
        if (multi) {
            if (have && count > 1) {
#pragma unroll
                for (int i = 0; i < V4; ++i) part_acc[local][i][sub] = acc.q[i];
                if (sub == 0) qr[grp * CAP] = sse;       // the group's queue is free by now
            }
            __syncthreads();
            if (owner && count > 1) {
#pragma unroll
                for (int i = 0; i < V4; ++i) {
                    float4 a4 = part_acc[leader][i][sub];
                    for (int k = 1; k < count; ++k) {
                        const float4 t = part_acc[leader + k][i][sub];
                        a4.x += t.x; a4.y += t.y; a4.z += t.z; a4.w += t.w;
                    }
                    acc.q[i] = a4;
                }
                sse = 0.f;
                for (int k = 0; k < count; ++k) sse += q_r[(leader + k) / G][((leader + k) % G) * CAP];
            }
        }
    }
    if (owner) {
        sgd_update(w, m4, acc, (is_user ? S.mU : S.mV) + row_off, (is_user ? S.U[cur ^ 1] : S.V[cur ^ 1]) + row_off,
                   snap_slot >= 0 ? S.snap + ((size_t)epoch * S.n_active + snap_slot) * D : nullptr);
        // train loss (utils.py:82): each user row adds its own squared errors to its own slot of
        // the epoch -- owner-only read-modify-write, so no atomics and a reproducible sum
        if (is_user && sub == 0 && sse != 0.f) {
            float *slot = S.sse + (size_t)epoch * S.n_user + row;
            stg(slot, ldg(slot) + sse);
        }
    }
}

template <int LPR, int V4>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(step_waves(LPR, V4)))) void mf_step_kernel(const ure_shard_t *__restrict__ shards, const shard_aux *__restrict__ aux, int64_t tick, int shard_fast)
{
#ifdef URE_TIMELINE
    const long long t0 = wall_clock64();
#endif
    mf_step<LPR, V4>(shards, aux, tick, shard_fast);
#ifdef URE_TIMELINE
    __syncthreads();
    if (g_timeline && threadIdx.x == 0) {
        const size_t lin = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        if (lin < 16384) {
            long long *e = g_timeline + ((size_t)(tick & 15) * 16384 + lin) * 8;
            e[0] = t0;
            e[1] = wall_clock64();
        }
    }
#endif
}

// End-of-epoch snapshot of a shard's tables into snapU/snapV[epoch] (optional; used to rebuild the
// reference's per-epoch test logs after shards were trained side by side, scratch.py:83-97).
// Rows the step kernel skips (lazy_rows) are written in closed form with the epoch's scalar.
__global__ __launch_bounds__(kBlock) void snapshot_kernel(const ure_shard_t *__restrict__ shards, const shard_aux *__restrict__ aux, int64_t ticks_done)
{
    const ure_shard_t &S = shards[blockIdx.y];
    const unsigned long long *__restrict__ row_mask = nullptr;
    const bool compact = S.snap != nullptr;
    if (!compact && (!S.snapU || !S.snapV)) return;
    if (compact && (S.touch_mode == 0 || S.touch_mode == 2) && S.row_slot) return;      // written by the step kernel's owners (touch_mode 2: and launch B)
    const int steps = shard_steps(S);
    if (ticks_done > (int64_t)steps * S.epochs || ticks_done % steps != 0) return;   // only at an epoch end of this shard
    const int epoch = (int)(ticks_done / steps) - 1;
    const int cur = (int)(ticks_done & 1);
    if (S.touch_mode == 1 || S.touch_mode == 2) row_mask = aux[blockIdx.y].mask[touch_last_window(aux[blockIdx.y], epoch) & 1];      // touch mode: a row's w sits in the buffer of its step-count parity
    const uint8_t *__restrict__ end_par = S.touch_mode == 3 ? aux[blockIdx.y].end_par[epoch & 1] : nullptr;               // touch_mode 3: the same, kept per row
    const float a = S.lazy_rows ? ldg(S.snap_a + epoch) : 0.f;
    const int d4 = S.d / 4;
    const int n_rows = compact ? S.n_active : S.n_user + S.n_item;
    float *__restrict__ su = compact ? nullptr : S.snapU + (size_t)epoch * S.n_user * S.d;
    float *__restrict__ sv = compact ? nullptr : S.snapV + (size_t)epoch * S.n_item * S.d;
    float *__restrict__ sc = compact ? S.snap + (size_t)epoch * S.n_active * S.d : nullptr;      // compact: the active rows only, in schedule order
    const int64_t total = (int64_t)n_rows * d4;
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
        const int idx = (int)(t / d4);
        const int row_id = ldg(S.sched + 4 * (size_t)idx);
        const bool is_user = row_id < S.n_user;
        const size_t o = (size_t)(is_user ? row_id : row_id - S.n_user) * S.d + (size_t)(t % d4) * 4;
        float4 v;
        if (S.lazy_rows && idx >= S.n_active) {
            const float4 w0 = ldg_f4((is_user ? S.U0 : S.V0) + o);
            v = make_float4(a * w0.x, a * w0.y, a * w0.z, a * w0.w);
        } else {
            const int from = end_par ? (int)ldg(end_par + row_id) : row_mask ? (__popcll(ldg(row_mask + row_id)) & 1) : cur;
            v = ldg_f4((is_user ? S.U[from] : S.V[from]) + o);
        }
        if (compact) stg_f4(sc + t * 4, v);
        else stg_f4((is_user ? su : sv) + o, v);
    }
}

// Rows without interactions in their shard, advanced in closed form: after T optimizer steps
// w_T = a_T * w_0 and m_T = b_T * w_0 with scalars (a, b) from the same recurrence the optimizer
// applies to every element (g = lam*w; m = mu*m + g (m = g first); w -= lr*m), evaluated in
// double on the host.  Differs from T fp32 steps by their accumulated rounding only (~1e-6).
__global__ __launch_bounds__(kBlock) void materialize_rows_kernel(const ure_shard_t *__restrict__ shards, const double *__restrict__ ab,
                                                                 int64_t ticks_done)
{
    const ure_shard_t &S = shards[blockIdx.y];
    if (!S.lazy_rows) return;
    const int steps = shard_steps(S);
    const int64_t T = min(ticks_done, (int64_t)steps * S.epochs);
    if (T <= 0) return;
    const int cur = (int)(T & 1);
    const float a = (float)ab[2 * blockIdx.y], b = (float)ab[2 * blockIdx.y + 1];
    const int d4 = S.d / 4;
    const int n_rows = S.n_user + S.n_item;
    const int64_t total = (int64_t)(n_rows - S.n_active) * d4;
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
        const int row_id = ldg(S.sched + 4 * (size_t)(S.n_active + t / d4));
        const bool is_user = row_id < S.n_user;
        const size_t o = (size_t)(is_user ? row_id : row_id - S.n_user) * S.d + (size_t)(t % d4) * 4;
        const float4 w0 = ldg_f4((is_user ? S.U0 : S.V0) + o);
        stg_f4((is_user ? S.U[cur] : S.V[cur]) + o, make_float4(a * w0.x, a * w0.y, a * w0.z, a * w0.w));
        stg_f4((is_user ? S.mU : S.mV) + o, make_float4(b * w0.x, b * w0.y, b * w0.z, b * w0.w));
    }
}

}  // namespace ure
#include "mf_touch.h"
#include "mf_index.h"
namespace ure {

template <int LPR, int V4>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(URE_TOUCH_WAVES))) void mf_touch_step_kernel(const ure_shard_t *__restrict__ shards, const shard_aux *__restrict__ aux, int64_t tick, int shard_fast)
{
    mf_touch_step<LPR, V4>(shards, aux, tick, shard_fast);
}

// Window start of a shard in touch mode (mf_touch.h): phase 1 builds the row masks of the window from the batch tags,
// phase 2 puts the buffer bits into the window's tags and advances every active row to its first step.  One launch per phase.
__device__ __forceinline__ bool touch_window_start(const ure_shard_t &S, const shard_aux &A, int64_t tick, TouchPos *P)
{
    if (tick >= (int64_t)A.steps * S.epochs) return false;
    const int epoch = (int)epoch_of(A, tick);
    const int s = (int)(tick - (int64_t)epoch * A.steps);
    if (s & (kTouchWindow - 1)) return false;
    *P = touch_pos(A, epoch, s);
    return true;
}

// (the phase is a template parameter so that a kernel trace tells the two launches apart)
template <int LPR, int PHASE>
__global__ __launch_bounds__(kBlock) void touch_prep_kernel(const ure_shard_t *__restrict__ shards, const shard_aux *__restrict__ aux, int64_t tick, int piece_blocks)
{
    __shared__ unsigned long long wg_mask[kBlock];
    const ure_shard_t &S = shards[blockIdx.y];
    const shard_aux &A = aux[blockIdx.y];
    TouchPos P;
    if (!touch_window_start(S, A, tick, &P)) return;
    if (PHASE == 1) {
        if ((int)blockIdx.x < touch_piece_blocks<LPR>(S.n_units, S.n_active, S.n_multi)) touch_build_masks<LPR>(S, A, P, (int)blockIdx.x, wg_mask);
    } else if (PHASE == 3) {      // -DURE_TOUCH_SPLIT (profiling only): the two halves of launch C as launches of their own
        touch_mark_tags<LPR>(S, A, P, (int)blockIdx.x);
    } else if (PHASE == 4) {
        touch_advance_rows(S, A, P, (int)blockIdx.x, (int)gridDim.x);
    } else if ((int)blockIdx.x < piece_blocks) touch_mark_tags<LPR>(S, A, P, (int)blockIdx.x);
    else touch_advance_rows(S, A, P, (int)blockIdx.x - piece_blocks, (int)gridDim.x - piece_blocks);
}

constexpr int kAheadRun = 8;        // blocks of single-pass rows one workgroup of touch_ahead_kernel works off

// touch_mode 2 (mf_touch.h): the two launches of an epoch start.  PHASE 1 = launch B (masks of the next epoch, hand-over to the owners,
// orphans), PHASE 2 = launch C (buffer bits of the next epoch's tags).  boot = 1 (tick 0 only): the same for epoch 0 itself, whose
// launch C also carries the one dense pass of the job (every row from the initial tables to its first step).
template <int LPR, int V4, int PHASE>
__global__ __launch_bounds__(kBlock) void touch_ahead_kernel(const ure_shard_t *__restrict__ shards, const shard_aux *__restrict__ aux, int64_t tick, int boot,
                                                             int piece_blocks)
{
    __shared__ unsigned long long wg_mask[kBlock];
    const ure_shard_t &S = shards[blockIdx.y];
    const shard_aux &A = aux[blockIdx.y];
    if (tick >= (int64_t)A.steps * S.epochs) return;
    const int epoch = (int)epoch_of(A, tick);
    if (tick != (int64_t)epoch * A.steps) return;                       // not an epoch start of this shard
    const int e_next = boot ? 0 : epoch + 1;
    const bool has_next = e_next < S.epochs;
    const int n_pieces = touch_piece_blocks<LPR>(S.n_units, S.n_active, S.n_multi);
    // A workgroup takes one block of work units, or kAheadRun consecutive blocks of single-pass rows: 16 rows of one scan pass each
    // are two dependent loads and a store -- at 60.8 k rows per shard and 32 shards the launch was bound by the rate at which
    // workgroups are dispatched (122 k of them), not by its 0.1 GB of tags.
    const int nbU = S.n_units / (kBlock / LPR);
    if ((int)blockIdx.x < piece_blocks) {
        const int first = (int)blockIdx.x < nbU ? (int)blockIdx.x : nbU + ((int)blockIdx.x - nbU) * kAheadRun;
        const int count = (int)blockIdx.x < nbU ? 1 : kAheadRun;
        if ((int)blockIdx.x >= nbU && kAheadRun * (kBlock / LPR) <= kBlock) {
            // the rows of this workgroup, the first of them first (the schedule is heaviest first): all of at most 8 slots -> a lane each
            const int rel0 = (first - nbU) * (kBlock / LPR);
            if (rel0 < S.n_active - S.n_multi) {
                const int4 sc0 = ldg_i4(S.sched + 4 * (size_t)(S.n_multi + rel0));
                if (sc0.z - sc0.y <= kSegPerLane) {
                    if ((int)threadIdx.x < kAheadRun * (kBlock / LPR)) {
                        if (PHASE == 1) touch_ahead_build_tiny<V4>(S, A, e_next, has_next, rel0);
                        else if (has_next) touch_ahead_mark_tiny(S, A, e_next, rel0);
                    }
                    return;
                }
            }
        }
        for (int r = 0; r < count; ++r) {
            const int wg = first + r;
            if (wg >= n_pieces) break;
            if (PHASE == 1) touch_ahead_build<LPR, V4>(S, A, e_next, has_next, wg, wg_mask);
            else if (has_next) touch_ahead_mark<LPR>(S, A, e_next, wg);
        }
    } else if (PHASE == 2 && boot) {
        touch_advance_rows(S, A, touch_pos(A, 0, 0), (int)blockIdx.x - piece_blocks, (int)gridDim.x - piece_blocks);
    }
}

// Tables read at `ticks_done`: a shard must stand at one of its epoch boundaries (or have finished).
__global__ __launch_bounds__(kBlock) void touch_collect_kernel(const ure_shard_t *__restrict__ shards, const shard_aux *__restrict__ aux, int64_t ticks_done)
{
    const ure_shard_t &S = shards[blockIdx.y];
    const shard_aux &A = aux[blockIdx.y];
    const int64_t T = min(ticks_done, (int64_t)A.steps * S.epochs);
    if (T % A.steps != 0) return;                      // refused on the host (ure_job_materialize)
    touch_collect_rows(S, A, (int)(T / A.steps) - 1, (int)(T & 1), (int)blockIdx.x, (int)gridDim.x);
}

static bool touch_prep_needed(const ure_job *job, int64_t tick)
{
    if (!job->touch || job->index) return false;
    for (size_t k = 0; k < job->host.size(); ++k) {
        const int64_t steps = job->aux_host[k].steps;
        if (tick < steps * job->host[k].epochs && (tick % steps) % kTouchWindow == 0) return true;      // (mode 2: steps <= 63, one window)
    }
    return false;
}

// touch_mode 2: the launches of an epoch start (after the tags of the next epoch are complete)
template <int LPR, int V4>
static void launch_touch_ahead(const ure_job *job, int64_t tick, hipStream_t st)
{
    const unsigned n_sh = (unsigned)job->host.size();
    int pieces = 1;                     // workgroups of the row-structured passes: one per block of units, one per kAheadRun blocks of single-pass rows
    for (const ure_shard_t &S : job->host) {
        const int nbU = S.n_units / (kBlock / LPR);
        pieces = std::max(pieces, nbU + (touch_piece_blocks<LPR>(S.n_units, S.n_active, S.n_multi) - nbU + kAheadRun - 1) / kAheadRun);
    }
    const unsigned adv_b = (unsigned)std::max<int64_t>(1, std::min<int64_t>((job->max_active4 + kBlock - 1) / kBlock, 8192));
    if (tick == 0) {
        hipLaunchKernelGGL((touch_ahead_kernel<LPR, V4, 1>), dim3((unsigned)pieces, n_sh), dim3(kBlock), 0, st, job->dev, job->dev_aux, tick, 1, pieces);
        hipLaunchKernelGGL((touch_ahead_kernel<LPR, V4, 2>), dim3((unsigned)pieces + adv_b, n_sh), dim3(kBlock), 0, st, job->dev, job->dev_aux, tick, 1, pieces);
    }
    hipLaunchKernelGGL((touch_ahead_kernel<LPR, V4, 1>), dim3((unsigned)pieces, n_sh), dim3(kBlock), 0, st, job->dev, job->dev_aux, tick, 0, pieces);
    hipLaunchKernelGGL((touch_ahead_kernel<LPR, V4, 2>), dim3((unsigned)pieces, n_sh), dim3(kBlock), 0, st, job->dev, job->dev_aux, tick, 0, pieces);
}

// the two launches of a window start in touch mode (an epoch start: after the epoch's batch tags are complete)
template <int LPR>
static void launch_touch_prep(const ure_job *job, int64_t tick, hipStream_t st)
{
    const unsigned n_sh = (unsigned)job->host.size();
    int pieces = 1;
    for (const ure_shard_t &S : job->host) pieces = std::max(pieces, touch_piece_blocks<LPR>(S.n_units, S.n_active, S.n_multi));
    const unsigned adv_b = (unsigned)std::max<int64_t>(1, std::min<int64_t>((job->max_active4 + kBlock - 1) / kBlock, 8192));
    hipLaunchKernelGGL((touch_prep_kernel<LPR, 1>), dim3((unsigned)pieces, n_sh), dim3(kBlock), 0, st, job->dev, job->dev_aux, tick, pieces);
#ifdef URE_TOUCH_SPLIT
    hipLaunchKernelGGL((touch_prep_kernel<LPR, 3>), dim3((unsigned)pieces, n_sh), dim3(kBlock), 0, st, job->dev, job->dev_aux, tick, pieces);
    hipLaunchKernelGGL((touch_prep_kernel<LPR, 4>), dim3(adv_b, n_sh), dim3(kBlock), 0, st, job->dev, job->dev_aux, tick, pieces);
#else
    hipLaunchKernelGGL((touch_prep_kernel<LPR, 2>), dim3((unsigned)pieces + adv_b, n_sh), dim3(kBlock), 0, st, job->dev, job->dev_aux, tick, pieces);
#endif
}

// touch_mode 3 (mf_index.h): does an epoch of some shard start at `tick`?
static bool index_epoch_start_needed(const ure_job *job, int64_t tick)
{
    if (!job->index) return false;
    for (size_t k = 0; k < job->host.size(); ++k) {
        const int64_t steps = job->aux_host[k].steps;
        if (tick < steps * job->host[k].epochs && tick % steps == 0) return true;
    }
    return false;
}

// ... then the epoch's slot index is built: masks, sort by step, items, and the one dense pass of the epoch
static void launch_index_epoch_start(const ure_job *job, int64_t tick, hipStream_t st)
{
    const unsigned n_sh = (unsigned)job->host.size();
    int chunks = 1, steps = 1;
    int64_t slots = 8;
    for (size_t k = 0; k < job->host.size(); ++k) {
        chunks = std::max(chunks, job->aux_host[k].idx_chunks);
        steps = std::max(steps, job->aux_host[k].steps);
        slots = std::max(slots, job->host[k].n_slots);
    }
    const unsigned step_blocks = (unsigned)((steps + kBlock - 1) / kBlock);
    const unsigned flag_blocks = (unsigned)((slots + kIdxFlagBlock - 1) / kIdxFlagBlock);
    const unsigned wave_blocks = (unsigned)((chunks + kWavesPerBlock - 1) / kWavesPerBlock);
    const unsigned row_blocks = (unsigned)std::max(1, std::min((job->max_rows + kBlock - 1) / kBlock, 4096));
    const unsigned adv_b = (unsigned)std::max<int64_t>(1, std::min<int64_t>((job->max_active4 + kBlock - 1) / kBlock, 8192));
    if (tick == 0) hipLaunchKernelGGL(idx_grp_row_kernel, dim3(std::min<unsigned>((unsigned)((slots / 8 + kBlock - 1) / kBlock), 8192u), n_sh), dim3(kBlock), 0, st, job->dev, job->dev_aux);
    const unsigned mask_chunks = (unsigned)((slots + kIdxMaskChunk - 1) / kIdxMaskChunk);
    hipLaunchKernelGGL(idx_clear_kernel, dim3(std::min<unsigned>((2 * mask_chunks + kBlock - 1) / kBlock, 1024u), n_sh), dim3(kBlock), 0, st, job->dev, job->dev_aux, tick);
    hipLaunchKernelGGL(idx_masks_kernel, dim3(mask_chunks, n_sh), dim3(kBlock), 0, st, job->dev, job->dev_aux, tick);
    hipLaunchKernelGGL(idx_parity_kernel, dim3(row_blocks, n_sh), dim3(kBlock), 0, st, job->dev, job->dev_aux, tick);
    hipLaunchKernelGGL(idx_hist_kernel, dim3(wave_blocks, n_sh), dim3(kBlock), 0, st, job->dev, job->dev_aux, tick);
    hipLaunchKernelGGL(idx_scan1_kernel, dim3(step_blocks, kIdxSeg, n_sh), dim3(kBlock), 0, st, job->dev, job->dev_aux, tick);
    hipLaunchKernelGGL(idx_scan2_kernel, dim3(n_sh), dim3(1024), 0, st, job->dev, job->dev_aux, tick);
    hipLaunchKernelGGL(idx_scan3_kernel, dim3(step_blocks, kIdxSeg, n_sh), dim3(kBlock), 0, st, job->dev, job->dev_aux, tick);
    // (epochs of at most 63 steps: a step's share of 1,024 slots is a run worth sorting in LDS first)
    constexpr bool staged_ok = true;
    if (steps <= kIdxWin && staged_ok)
        hipLaunchKernelGGL(idx_scatter_short_kernel, dim3(wave_blocks, n_sh), dim3(kBlock), 0, st, job->dev, job->dev_aux, tick);
    else if (job->scatter_staged)
        hipLaunchKernelGGL((idx_scatter_staged_kernel<kIdxStagedWaves>), dim3((unsigned)chunks, n_sh), dim3(kIdxStagedWaves * kWave), 0, st, job->dev, job->dev_aux, tick);
    else
        hipLaunchKernelGGL(idx_scatter_kernel, dim3(wave_blocks, n_sh), dim3(kBlock), 0, st, job->dev, job->dev_aux, tick);
    hipLaunchKernelGGL(idx_mark_kernel, dim3(flag_blocks, n_sh), dim3(kBlock), 0, st, job->dev, job->dev_aux, tick);
    hipLaunchKernelGGL(idx_blkscan_kernel, dim3(n_sh), dim3(1024), 0, st, job->dev, job->dev_aux, tick);
    hipLaunchKernelGGL(idx_emit_kernel, dim3(flag_blocks, n_sh), dim3(kBlock), 0, st, job->dev, job->dev_aux, tick);
    hipLaunchKernelGGL(idx_heavy_kernel, dim3((unsigned)steps, n_sh), dim3(kIdxHeavyMax), 0, st, job->dev, job->dev_aux, tick);
    hipLaunchKernelGGL(idx_advance_kernel, dim3(adv_b, n_sh), dim3(kBlock), 0, st, job->dev, job->dev_aux, tick);
}

template <int LPR, int V4>
static void launch_index_step(const ure_job *job, int64_t tick, hipStream_t st)
{
    constexpr int UPB = kBlock / LPR;
    const unsigned n_sh = (unsigned)job->host.size();
    int blocks = 1, split = 0;
    for (size_t k = 0; k < job->host.size(); ++k) {
        const shard_aux &A = job->aux_host[k];
        if (tick >= (int64_t)A.steps * job->host[k].epochs) continue;
        blocks = std::max(blocks, A.idx_hw + (A.idx_light + UPB - 1) / UPB);
        split = std::max(split, job->host[k].n_split);
    }
    const unsigned per = ((unsigned)blocks + 7u) / 8u;
    hipLaunchKernelGGL((mf_index_step_kernel<LPR, V4>), dim3(8u * n_sh * per), dim3(kBlock), 0, st, job->dev, job->dev_aux, tick, n_sh, per);
    if (split > 0) hipLaunchKernelGGL((idx_combine_kernel<LPR, V4>), dim3((unsigned)split, n_sh), dim3(kWave), 0, st, job->dev, job->dev_aux, tick);
}

template <int LPR, int V4>
static void launch_step(const ure_job *job, int64_t tick, hipStream_t st)
{
    if (job->index) { launch_index_step<LPR, V4>(job, tick, st); return; }
    // grid.x = the largest need of any shard AT THIS TICK (row workgroups + the tag riders its
    // current step carries); finished shards need nothing
    int blocks = 1;
    for (size_t k = 0; k < job->host.size(); ++k) {
        const ure_shard_t &S = job->host[k];
        const int64_t steps = ((int64_t)S.N + S.batch - 1) / S.batch;
        if (tick >= steps * S.epochs) continue;
        const int64_t epoch = tick / steps;
        blocks = std::max(blocks, job->row_blocks[k] + tag_ride(job->aux_host[k], (int)(tick - epoch * steps), epoch + tag_ahead(S) < S.epochs).count);
    }
    int shard_fast = job->shard_fast && blocks <= 65535;
    const unsigned n_sh = (unsigned)job->host.size();
    dim3 grid = shard_fast ? dim3(n_sh, (unsigned)blocks) : dim3((unsigned)blocks, n_sh);
    if (job->shard_sliced && n_sh < (1u << 16) && (uint64_t)((blocks + 7) / 8) * 8 * n_sh < (1ull << 31)) {
        shard_fast = 2 | (int)(n_sh << 8);
        grid = dim3((unsigned)((blocks + 7) / 8) * 8 * n_sh);
    }
    if (job->touch) {
        hipLaunchKernelGGL((mf_touch_step_kernel<LPR, V4>), grid, dim3(kBlock), 0, st, job->dev, job->dev_aux, tick, shard_fast);
        return;
    }
    hipLaunchKernelGGL((mf_step_kernel<LPR, V4>), grid, dim3(kBlock), 0, st, job->dev, job->dev_aux, tick, shard_fast);
}

}  // namespace ure

using namespace ure;

#ifdef URE_TIMELINE
extern "C" int ure_debug_timeline(void *buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(ure::g_timeline), &buf, sizeof(buf)); }
#endif

extern "C" {

// Library-owned device memory of touch mode (masks, tables, the slot index) comes from a small cache of blocks instead of hipMalloc / hipFree
// per job: a 32-shard job makes 32-128 allocations, each a driver call and a synchronous fill -- 24 ms of a 110 ms request at configs[3]'s
// shape (k = 16), 28 ms at k = 128 --, and every request of a process asks for the same sizes again.  Blocks go back when their job is destroyed
// (after the device has been waited for); up to kBlockCacheBytes are kept per device, the rest is freed.
namespace ure {
namespace {
struct BlockCache {
    std::multimap<size_t, void *> idle;
    std::map<void *, size_t> size_of;
    size_t held = 0;
};
std::mutex g_block_lock;
std::map<int, BlockCache> g_block_cache;
constexpr size_t kBlockCacheBytes = (size_t)8 << 30;

hipError_t block_malloc(void **out, size_t bytes)
{
    bytes = (std::max<size_t>(bytes, 1) + ((size_t)256 << 10) - 1) / ((size_t)256 << 10) * ((size_t)256 << 10);
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> hold(g_block_lock);
        BlockCache &C = g_block_cache[dev];
        auto it = C.idle.lower_bound(bytes);
        if (it != C.idle.end() && it->first <= bytes + bytes / 4 + ((size_t)1 << 20)) {
            *out = it->second;
            C.held -= it->first;
            C.idle.erase(it);
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(out, bytes);
    if (e != hipSuccess) {                                   // (out of memory with idle blocks held: give them back and ask once more)
        std::vector<void *> drop;
        {
            std::lock_guard<std::mutex> hold(g_block_lock);
            BlockCache &C = g_block_cache[dev];
            for (auto &kv : C.idle) { drop.push_back(kv.second); C.size_of.erase(kv.second); }
            C.idle.clear();
            C.held = 0;
        }
        for (void *p : drop) (void)hipFree(p);
        (void)hipGetLastError();
        e = hipMalloc(out, bytes);
    }
    if (e == hipSuccess) {
        std::lock_guard<std::mutex> hold(g_block_lock);
        g_block_cache[dev].size_of[*out] = bytes;
    }
    return e;
}

void block_free(void *p)
{
    if (!p) return;
    // (the device the block lives on, not the calling thread's current one: a job is destroyed on a worker thread, whose current device is 0
    // whatever GPU its rank trains on)
    int dev = 0;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) == hipSuccess) dev = attr.device;
    else { (void)hipGetLastError(); (void)hipGetDevice(&dev); }
    {
        std::lock_guard<std::mutex> hold(g_block_lock);
        BlockCache &C = g_block_cache[dev];
        auto it = C.size_of.find(p);
        if (it != C.size_of.end() && C.held + it->second <= kBlockCacheBytes) {
            C.idle.emplace(it->second, p);
            C.held += it->second;
            return;
        }
        if (it != C.size_of.end()) C.size_of.erase(it);
    }
    (void)hipFree(p);
}
}  // namespace
}  // namespace ure

int ure_job_create(const ure_shard_t *shards, int n_shards, ure_job_t **out)
{
    URE_ARG(shards && out && n_shards > 0 && n_shards <= 65535);
    auto *job = new ure::ure_job();
    job->host.assign(shards, shards + n_shards);
    for (int k = 0; k < n_shards; ++k) {
        const ure_shard_t &S = shards[k];
        const int n_rows = S.n_user + S.n_item;
        const bool ok = S.N > 0 && S.n_user > 0 && S.n_item > 0 && S.batch > 0 && S.epochs > 0 && pow2(S.d) && S.d >= 4 &&
                        S.d <= 256 && S.n_active >= 0 && S.n_active <= n_rows && S.units && S.n_units >= 0 &&
                        S.n_units % (kBlock / lanes_per_row(S.d)) == 0 && S.n_slots >= S.N && S.ent_oid && S.ent_r && S.ent_tag && S.ent_src &&
                        S.file_tag && S.sched && S.U[0] && S.U[1] && S.V[0] && S.V[1] && S.mU && S.mV && (S.perm || S.file_tags) && S.lr && S.sse &&
                        (!S.lazy_rows || (S.U0 && S.V0 && S.lr_host));
        if (!ok) { delete job; return fail(-1, "ure_job_create: shard %d has an invalid descriptor", k); }
        if (S.d != shards[0].d) { delete job; return fail(-1, "ure_job_create: all shards of a job share d"); }
        const int64_t steps = ((int64_t)S.N + S.batch - 1) / S.batch;
        if (steps > 65534) { delete job; return fail(-1, "ure_job_create: shard %d needs %lld steps/epoch (> 65534)", k, (long long)steps); }
        job->ticks = std::max(job->ticks, steps * S.epochs);
        const int per_block = kBlock / lanes_per_row(S.d);
        if (S.touch_mode && (S.n_multi < 0 || S.n_multi > S.n_active)) { delete job; return fail(-1, "ure_job_create: shard %d: n_multi outside [0, n_active]", k); }
        const int blocks = S.touch_mode ? S.n_units / per_block + (S.n_active - S.n_multi + kBlock - 1) / kBlock      // units + candidate workgroups
                                        : S.n_units / per_block + (S.lazy_rows ? 0 : (n_rows - S.n_active + per_block - 1) / per_block);
        job->row_blocks.push_back(blocks);
        job->max_n = std::max(job->max_n, S.N);
        job->max_slots = std::max(job->max_slots, S.n_slots);
        const bool small = tag_partitioned(S.N);
        if (small && !(S.inv_stage && S.inv_off)) { delete job; return fail(-1, "ure_job_create: shard %d lacks inv_stage / inv_off", k); }
        (small ? job->small_shards : job->large_shards) = true;
        if (small) job->max_small_n = std::max(job->max_small_n, S.N);
    }
    job->d = shards[0].d;
    if (const char *e = std::getenv("URE_SHARD_FAST")) { job->shard_fast = e[0] != '0'; job->shard_sliced = e[0] == '2'; }
    if (const char *e = std::getenv("URE_INDEX_STAGED")) job->scatter_staged = e[0] != '0';
    for (int k = 0; k < n_shards; ++k) {
        const ure_shard_t &S = shards[k];
        job->lr_host.emplace_back(S.lr_host ? std::vector<float>(S.lr_host, S.lr_host + S.epochs) : std::vector<float>());
        job->host[k].lr_host = nullptr;                       // the caller's array need not outlive this call
        job->max_lazy = std::max<int64_t>(job->max_lazy, S.lazy_rows ? (int64_t)(S.n_user + S.n_item - S.n_active) * (S.d / 4) : 0);
        if (S.snapU || S.snapV || S.snap) {
            const bool full = S.snapU && S.snapV && !S.snap, compact = S.snap && !S.snapU && !S.snapV && S.lazy_rows;
            if (!(full || compact) || (S.lazy_rows && !S.snap_a)) { delete job; return fail(-1, "ure_job_create: shard %d has an incomplete snapshot set (full: snapU + snapV; compact: snap with lazy_rows; snap_a with lazy_rows)", k); }
            job->snapshots = true;
            const int64_t rows = compact ? S.n_active : S.n_user + S.n_item;
            job->snap_blocks = std::max<unsigned>(job->snap_blocks, (unsigned)std::max<int64_t>(1, std::min<int64_t>((rows * (S.d / 4) + kBlock - 1) / kBlock, 2048)));
        }
    }
    for (int k = 0; k < n_shards; ++k) job->aux_host.push_back(make_shard_aux(shards[k]));
    // ---- touch mode: all shards of the job or none; masks and closed-form tables are library-owned
    for (int k = 0; k < n_shards; ++k) job->touch = job->touch || shards[k].touch_mode != 0;
    job->ahead = shards[0].touch_mode == 2;
    job->index = shards[0].touch_mode == 3;
    for (int k = 0; k < n_shards; ++k) job->all_file_tags = job->all_file_tags && shards[k].file_tags != nullptr;
    if (job->touch) {
        for (int k = 0; k < n_shards; ++k) {
            const ure_shard_t &S = shards[k];
            const char *why = !S.touch_mode ? "every shard of a job must ask for it" :
                              (S.touch_mode < 1 || S.touch_mode > 3) ? "touch_mode is 0, 1, 2 or 3" :
                              S.touch_mode != shards[0].touch_mode ? "every shard of a job must ask for the same touch mode" :
                              (S.touch_mode == 2 && job->aux_host[k].steps > kAheadMaxSteps) ? "touch_mode 2 takes at most 63 steps per epoch (the mask word's top bit is the start buffer)" :
                              (S.touch_mode == 2 && (S.snapU || S.snapV)) ? "touch_mode 2 writes compact snapshots only (snap + row_slot)" :
                              (S.touch_mode == 2 && S.snap && !S.row_slot) ? "touch_mode 2 needs row_slot with snap" :
                              (S.touch_mode == 3 && job->aux_host[k].steps > kIdxMaxSteps) ? "touch_mode 3 takes at most 1008 steps per epoch (16 mask words of 63 steps)" :
                              (S.touch_mode == 3 && (S.n_multi > kIdxHeavyMax || S.n_split < 0 || S.n_split > S.n_multi)) ? "touch_mode 3: 0 <= n_split <= n_multi <= 256" :
                              (S.touch_mode == 3 && S.batch > 200000) ? "touch_mode 3 takes batches of at most 200,000 (the parts of a split row are numbered in 11 bits)" :
                              !S.lazy_rows ? "it needs lazy_rows" :
                              job->aux_host[k].steps > kTouchMaxSteps ? "more than 32000 steps per epoch (the step number shares the 16-bit batch tag with the buffer bit)" :
                              (S.epochs != shards[0].epochs || S.lam != shards[0].lam || S.mu != shards[0].mu ||
                               job->lr_host[k] != job->lr_host[0]) ? "the shards' optimizer schedules differ" : nullptr;
            if (why) { delete job; return fail(-1, "ure_job_create: touch mode refused for shard %d: %s", k, why); }
            job->max_units = std::max(job->max_units, S.n_units);
            job->max_active4 = std::max<int64_t>(job->max_active4, (int64_t)S.n_active * (S.d / 4));
            job->max_rows = std::max(job->max_rows, S.n_user + S.n_item);
        }
    }
    hipError_t e = hipMalloc(&job->dev, sizeof(ure_shard_t) * n_shards);
    if (e == hipSuccess) e = hipMemcpy(job->dev, job->host.data(), sizeof(ure_shard_t) * n_shards, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&job->dev_ab, sizeof(double) * 2 * n_shards);
    if (job->touch && e == hipSuccess) {
        // A_e^j for j = 0..64 (the length of a window) at every epoch's learning rate, in double: (w, m)' = A (w, m),
        // m' = mu m + lam w, w' = w - lr m'
        const int E = shards[0].epochs;
        int tab_n = kTouchTab;                                      // touch_mode 3: a row may wait a whole epoch for its next step
        if (job->index)
            for (int k = 0; k < n_shards; ++k) tab_n = std::max(tab_n, job->aux_host[k].steps + 1);
        for (int k = 0; k < n_shards; ++k) job->aux_host[k].ptab_stride = tab_n;
        std::vector<float> tab((size_t)E * tab_n * 4);
        const double lam = (double)shards[0].lam, mu = (double)shards[0].mu;
        for (int ep = 0; ep < E; ++ep) {
            const double lr = (double)job->lr_host[0][(size_t)ep];
            const double a11 = 1.0 - lr * lam, a12 = -lr * mu, a21 = lam, a22 = mu;
            double p11 = 1.0, p12 = 0.0, p21 = 0.0, p22 = 1.0;
            for (int j = 0; j < tab_n; ++j) {
                float *o = &tab[((size_t)ep * tab_n + j) * 4];
                o[0] = (float)p11; o[1] = (float)p12; o[2] = (float)p21; o[3] = (float)p22;
                const double q11 = a11 * p11 + a12 * p21, q12 = a11 * p12 + a12 * p22;
                const double q21 = a21 * p11 + a22 * p21, q22 = a21 * p12 + a22 * p22;
                p11 = q11; p12 = q12; p21 = q21; p22 = q22;
            }
        }
        void *ptab = nullptr;
        e = block_malloc(&ptab, tab.size() * sizeof(float));
        if (e == hipSuccess) { job->touch_mem.push_back(ptab); e = hipMemcpy(ptab, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice); }
        // (epochs of at most 63 steps in every shard: idx_scatter_short_kernel sorts 1,024 slots at a time in LDS, and with a chunk of that size a
        // wavefront per 1,024 slots instead of 4,096 -- the shards of such a job start their epochs at different ticks, 25 / 26 / 27 steps apart, and a
        // single shard's 340 chunks of 4,096 left three quarters of the chip idle: 100 us per shard and epoch against 26 with all 32 at once)
        bool index_short = true;
        for (int k = 0; k < n_shards; ++k) index_short = index_short && job->aux_host[k].steps <= kIdxWin;
        for (int k = 0; k < n_shards && e == hipSuccess && job->index; ++k) {
            // touch_mode 3: the slot index of the current epoch (mf_index.h), one allocation per shard
            const ure_shard_t &S = shards[k];
            shard_aux &A = job->aux_host[k];
            A.ptab = static_cast<const float4 *>(ptab);
            const size_t n_all = (size_t)S.n_user + S.n_item, steps = (size_t)A.steps, slots = (size_t)S.n_slots;
            A.idx_words = (int32_t)((steps + kIdxWin - 1) / kIdxWin);
            A.idx_chunk = index_short ? kIdxStage : kIdxChunk;
            A.idx_chunks = (int32_t)((slots + A.idx_chunk - 1) / A.idx_chunk);
            A.idx_hw = S.n_multi + 2 * S.batch / kIdxPart + 2;
            A.idx_light = (int32_t)std::min<int64_t>(2 * (int64_t)S.batch, S.n_active);
            size_t at = 0;
            auto take = [&at](size_t bytes) { const size_t o = at; at += (bytes + 255) / 256 * 256; return o; };
            const size_t o_grp = take(slots / 8 * 8), o_w = take((size_t)A.idx_words * n_all * 8), o_par = take(2 * n_all), o_first = take(n_all * 2 + (size_t)A.idx_words * n_all * 2);
            const size_t o_hist = take((size_t)A.idx_chunks * (steps + 1) * 4), o_seg = take((size_t)kIdxSeg * (steps + 1) * 4), o_sb = take((steps + 2) * 4);
            const size_t o_ss = take(slots * 16), o_rf = take((slots / 64 + 2) * 8), o_bc = take((slots / kIdxFlagBlock + 2) * 4);
            const size_t o_it = take(((size_t)std::min<int64_t>(2 * (int64_t)S.N, S.n_slots) + 1) * 16), o_si = take((steps + 2) * 4), o_hc = take(steps * 4);
            const size_t o_it2 = take(((size_t)std::min<int64_t>(2 * (int64_t)S.N, S.n_slots) + 1) * 16);
            const size_t o_cum = take(steps * (kIdxHeavyMax + 1) * 4), o_pa = take((size_t)A.idx_hw * (S.d + 4) * 4);
            const size_t o_map = take(steps * (size_t)A.idx_hw * 4), o_wg = take(steps * 4), o_sd = take(steps * 16);
            void *mem = nullptr;
            e = block_malloc(&mem, at);
            if (e != hipSuccess) break;
            job->touch_mem.push_back(mem);
            char *b = static_cast<char *>(mem);
            e = hipMemset(b + o_w, 0, (o_hist - o_w));                       // masks, end-of-epoch buffers, first steps
            A.grp_row = reinterpret_cast<int32_t *>(b + o_grp);
            A.W = reinterpret_cast<unsigned long long *>(b + o_w);
            A.end_par[0] = reinterpret_cast<uint8_t *>(b + o_par);
            A.end_par[1] = A.end_par[0] + n_all;
            A.first_step = reinterpret_cast<uint16_t *>(b + o_first);
            A.next_first = A.first_step + n_all;
            A.hist = reinterpret_cast<uint32_t *>(b + o_hist);
            A.seg = reinterpret_cast<uint32_t *>(b + o_seg);
            A.step_begin = reinterpret_cast<uint32_t *>(b + o_sb);
            A.sslot = reinterpret_cast<uint4 *>(b + o_ss);
            A.runflag = reinterpret_cast<unsigned long long *>(b + o_rf);
            A.blk_cnt = reinterpret_cast<uint32_t *>(b + o_bc);
            A.items = reinterpret_cast<int4 *>(b + o_it);
            A.items2 = reinterpret_cast<uint4 *>(b + o_it2);
            A.step_item = reinterpret_cast<uint32_t *>(b + o_si);
            A.heavy_cnt = reinterpret_cast<uint32_t *>(b + o_hc);
            A.heavy_cum = reinterpret_cast<uint32_t *>(b + o_cum);
            A.partial = reinterpret_cast<float *>(b + o_pa);
            A.heavy_map = reinterpret_cast<uint32_t *>(b + o_map);
            A.heavy_wg = reinterpret_cast<uint32_t *>(b + o_wg);
            A.step_desc = reinterpret_cast<uint4 *>(b + o_sd);
            job->index_split = job->index_split || S.n_split > 0;
        }
        for (int k = 0; k < n_shards && e == hipSuccess && !job->index; ++k) {
            const size_t bytes = (size_t)(shards[k].n_user + shards[k].n_item) * sizeof(unsigned long long);
            void *mk = nullptr;
            e = block_malloc(&mk, 2 * bytes);
            if (e != hipSuccess) break;
            job->touch_mem.push_back(mk);
            e = hipMemset(mk, 0, 2 * bytes);
            job->aux_host[k].mask[0] = static_cast<unsigned long long *>(mk);
            job->aux_host[k].mask[1] = static_cast<unsigned long long *>(mk) + (shards[k].n_user + shards[k].n_item);
            job->aux_host[k].ptab = static_cast<const float4 *>(ptab);
            // the masks once more in work order: per work unit (multi-pass rows) and per single-pass row of the schedule
            const size_t n_um = (size_t)std::max(shards[k].n_units, 1), n_sm = (size_t)std::max(shards[k].n_active - shards[k].n_multi, 1);
            void *wm = nullptr;
            e = block_malloc(&wm, (2 * n_um + n_sm) * sizeof(unsigned long long));
            if (e != hipSuccess) break;
            job->touch_mem.push_back(wm);
            e = hipMemset(wm, 0, (2 * n_um + n_sm) * sizeof(unsigned long long));
            job->aux_host[k].unit_mask = static_cast<unsigned long long *>(wm);
            job->aux_host[k].unit_own = static_cast<unsigned long long *>(wm) + n_um;
            job->aux_host[k].sched_mask = static_cast<unsigned long long *>(wm) + 2 * n_um;
            job->aux_host[k].n_um = (int32_t)n_um;
            job->aux_host[k].n_sm = (int32_t)n_sm;
            if (!job->ahead && job->aux_host[k].windows > 1 && shards[k].n_units > 0) {
                // epochs of several windows: the steps of every scan pass of the multi-pass units (mf_touch.h: pass skipping)
                void *pm = nullptr;
                const size_t bytes = ((size_t)shards[k].n_slots / 8 + 1) * sizeof(unsigned long long);
                e = block_malloc(&pm, bytes);
                if (e != hipSuccess) break;
                job->touch_mem.push_back(pm);
                e = hipMemset(pm, 0, bytes);
                if (e != hipSuccess) break;
                job->aux_host[k].pass_mask = static_cast<unsigned long long *>(pm);
            }
            if (job->ahead) {
                // touch_mode 2: the work-order masks once per epoch parity (the first set is the one above), and the owners' hand-over
                void *wm2 = nullptr, *nf = nullptr;
                e = block_malloc(&wm2, (2 * n_um + n_sm) * sizeof(unsigned long long));
                if (e != hipSuccess) break;
                job->touch_mem.push_back(wm2);
                e = hipMemset(wm2, 0, (2 * n_um + n_sm) * sizeof(unsigned long long));
                if (e != hipSuccess) break;
                e = block_malloc(&nf, n_um + n_sm);
                if (e != hipSuccess) break;
                job->touch_mem.push_back(nf);
                e = hipMemset(nf, 0xFF, n_um + n_sm);
                job->aux_host[k].ahead_masks[0] = static_cast<unsigned long long *>(wm);
                job->aux_host[k].ahead_masks[1] = static_cast<unsigned long long *>(wm2);
                job->aux_host[k].unit_nf = static_cast<uint8_t *>(nf);
                job->aux_host[k].sched_nf = static_cast<uint8_t *>(nf) + n_um;
            }
        }
    }
    if (e == hipSuccess) e = hipMalloc(&job->dev_aux, sizeof(shard_aux) * n_shards);
    if (e == hipSuccess) e = hipMemcpy(job->dev_aux, job->aux_host.data(), sizeof(shard_aux) * n_shards, hipMemcpyHostToDevice);
    if (e != hipSuccess) { const int rc = fail((int)e, "ure_job_create: %s", hipGetErrorString(e)); ure_job_destroy(reinterpret_cast<ure_job_t *>(job)); return rc; }
    *out = reinterpret_cast<ure_job_t *>(job);
    return 0;
}

int ure_job_destroy(ure_job_t *j)
{
    auto *job = reinterpret_cast<ure::ure_job *>(j);
    if (!job) return 0;
    if (job->dev) (void)hipFree(job->dev);
    if (job->dev_ab) (void)hipFree(job->dev_ab);
    if (job->dev_aux) (void)hipFree(job->dev_aux);
    for (void *p : job->touch_mem) ure::block_free(p);          // (the three frees above have waited for the device)
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
    auto mark = [&](std::vector<hipEvent_t> *v) -> int {
        if (!v) return 0;
        hipEvent_t e;
        URE_HIP(hipEventCreate(&e));
        v->push_back(e);
        URE_HIP(hipEventRecord(e, st));
        return 0;
    };
    for (int64_t t = tick0; t < tick1; ++t) {
        // batch tags: the step kernel prepares the next epoch's itself (riders); epoch 0, shards with
        // fewer than 3 steps per epoch and very large shards get standalone launches here
        if (const int passes = tag_prep_needed(job, t)) {
            if (int rc = mark(assign_ev)) return rc;
            if (passes & 1) launch_tag_prep(job, t, st, 0);
            if (passes & 2) launch_tag_prep(job, t, st, 1);
            if (int rc = mark(assign_ev)) return rc;
        }
        if (job->ahead && touch_prep_needed(job, t)) {
            if (int rc = mark(assign_ev)) return rc;
            switch (job->d) {
                case 4: launch_touch_ahead<1, 1>(job, t, st); break;
                case 8: launch_touch_ahead<2, 1>(job, t, st); break;
                case 16: launch_touch_ahead<4, 1>(job, t, st); break;
                case 32: launch_touch_ahead<lanes_per_row(32), 32 / (4 * lanes_per_row(32))>(job, t, st); break;
                case 64: launch_touch_ahead<8, 2>(job, t, st); break;
                case 128: launch_touch_ahead<16, 2>(job, t, st); break;
                default: launch_touch_ahead<32, 2>(job, t, st); break;
            }
            if (int rc = mark(assign_ev)) return rc;
        } else if (index_epoch_start_needed(job, t)) {
            if (int rc = mark(assign_ev)) return rc;
            launch_index_epoch_start(job, t, st);
            if (int rc = mark(assign_ev)) return rc;
        } else if (touch_prep_needed(job, t)) {
            if (int rc = mark(assign_ev)) return rc;
            switch (lanes_per_row(job->d)) {
                case 1: launch_touch_prep<1>(job, t, st); break;
                case 2: launch_touch_prep<2>(job, t, st); break;
                case 4: launch_touch_prep<4>(job, t, st); break;
                case 8: launch_touch_prep<8>(job, t, st); break;
                case 16: launch_touch_prep<16>(job, t, st); break;
                default: launch_touch_prep<32>(job, t, st); break;
            }
            if (int rc = mark(assign_ev)) return rc;
        }
        if (int rc = mark(step_ev)) return rc;
        switch (job->d) {
            case 4: launch_step<1, 1>(job, t, st); break;
            case 8: launch_step<2, 1>(job, t, st); break;
            case 16: launch_step<4, 1>(job, t, st); break;
            case 32: launch_step<lanes_per_row(32), 32 / (4 * lanes_per_row(32))>(job, t, st); break;
            case 64: launch_step<8, 2>(job, t, st); break;
            case 128: launch_step<16, 2>(job, t, st); break;
            case 256: launch_step<32, 2>(job, t, st); break;
            default: return fail(-1, "ure_job_train: unsupported d=%d", job->d);
        }
        if (int rc = mark(step_ev)) return rc;
        if (job->snapshots) {
            bool epoch_end = false;
            for (const ure_shard_t &S : job->host) {
                const int64_t steps = ((int64_t)S.N + S.batch - 1) / S.batch;
                const bool by_kernel = S.snapU || (S.snap && (S.touch_mode == 1 || S.touch_mode == 3 || !S.row_slot));      // otherwise the step kernel's owners wrote it
                if (by_kernel && t + 1 <= steps * S.epochs && (t + 1) % steps == 0) { epoch_end = true; break; }
            }
            if (epoch_end)
                hipLaunchKernelGGL(snapshot_kernel, dim3(job->snap_blocks, (unsigned)job->host.size()), dim3(kBlock), 0, st, job->dev, job->dev_aux, t + 1);
        }
    }
    URE_HIP(hipGetLastError());
    return 0;
}

int ure_job_materialize(ure_job_t *j, int64_t ticks_done, void *stream)
{
    auto *job = reinterpret_cast<ure::ure_job *>(j);
    URE_ARG(job && ticks_done >= 0);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t n = job->host.size();
    if (job->touch) {
        // rows are kept valid for their NEXT own step: the tables as of `ticks_done` exist only where a shard stands
        // at one of its epoch boundaries (all its rows are then valid for that boundary)
        for (size_t k = 0; k < n; ++k) {
            const int64_t steps = job->aux_host[k].steps;
            const int64_t T = std::min(ticks_done, steps * job->host[k].epochs);
            if (T % steps != 0) return fail(-1, "ure_job_materialize: touch mode: shard %d is inside an epoch at tick %lld (tables are readable at its epoch boundaries only)", (int)k, (long long)ticks_done);
            if (job->ahead && T != 0 && T != steps * job->host[k].epochs)
                return fail(-1, "ure_job_materialize: touch_mode 2: shard %d has not finished at tick %lld (rows are kept valid for their next step across epoch boundaries: tables are readable at the end of training, epoch ends through the compact snapshots)", (int)k, (long long)ticks_done);
        }
        const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>((job->max_active4 + kBlock - 1) / kBlock, 8192));
        hipLaunchKernelGGL(touch_collect_kernel, dim3(blocks, (unsigned)n), dim3(kBlock), 0, st, job->dev, job->dev_aux, ticks_done);
    }
    if (job->max_lazy == 0 || ticks_done == 0) { URE_HIP(hipGetLastError()); return 0; }
    job->ab_host.assign(2 * n, 0.0);
    for (size_t k = 0; k < n; ++k) {
        const ure_shard_t &S = job->host[k];
        if (!S.lazy_rows) continue;
        const int64_t steps = ((int64_t)S.N + S.batch - 1) / S.batch;
        const int64_t T = std::min(ticks_done, steps * S.epochs);
        double a = 1.0, b = 0.0;
        const double lam = (double)S.lam, mu = (double)S.mu;
        for (int64_t t = 0; t < T; ++t) {
            b = t == 0 ? lam * a : mu * b + lam * a;
            a -= (double)job->lr_host[k][(size_t)(t / steps)] * b;
        }
        job->ab_host[2 * k] = a;
        job->ab_host[2 * k + 1] = b;
    }
    // pageable source: the runtime stages the copy, so ab_host may be reused right after
    URE_HIP(hipMemcpyAsync(job->dev_ab, job->ab_host.data(), sizeof(double) * 2 * n, hipMemcpyHostToDevice, st));
    const unsigned blocks = (unsigned)std::min<int64_t>((job->max_lazy + kBlock - 1) / kBlock, 2048);
    hipLaunchKernelGGL(materialize_rows_kernel, dim3(blocks, (unsigned)n), dim3(kBlock), 0, st, job->dev, job->dev_ab, ticks_done);
    URE_HIP(hipGetLastError());
    return 0;
}

int ure_job_touch_rows(ure_job_t *j, int64_t *pairs, int64_t *window_steps)
{
    auto *job = reinterpret_cast<ure::ure_job *>(j);
    URE_ARG(job && pairs && window_steps);
    for (size_t k = 0; k < job->host.size(); ++k) { pairs[k] = -1; window_steps[k] = 0; }
    if (!job->touch || job->next_tick == 0) return 0;
    URE_HIP(hipDeviceSynchronize());
    if (job->index) {
        // touch_mode 3: the (row, step) runs of the shard's current epoch = the items of its index
        for (size_t k = 0; k < job->host.size(); ++k) {
            const shard_aux &A = job->aux_host[k];
            uint32_t n_items = 0;
            URE_HIP(hipMemcpy(&n_items, A.step_item + A.steps, sizeof(n_items), hipMemcpyDeviceToHost));
            pairs[k] = n_items;
            window_steps[k] = A.steps;
        }
        return 0;
    }
    std::vector<unsigned long long> host;
    for (size_t k = 0; k < job->host.size(); ++k) {
        const ure_shard_t &S = job->host[k];
        const shard_aux &A = job->aux_host[k];
        const int64_t steps = A.steps;
        const int64_t last = std::min<int64_t>(job->next_tick, steps * S.epochs) - 1;      // the last step of the shard that was launched
        const int64_t epoch = last / steps, win = (last % steps) / kTouchWindow;
        host.resize((size_t)S.n_user + S.n_item);
        URE_HIP(hipMemcpy(host.data(), A.mask[(epoch * A.windows + win) & 1], host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        int64_t n = 0;
        for (unsigned long long m : host) n += __builtin_popcountll(job->ahead ? m & ~(1ull << 63) : m);
        pairs[k] = n;
        window_steps[k] = std::min<int64_t>(kTouchWindow, steps - win * kTouchWindow);
    }
    return 0;
}

int ure_job_index_read(ure_job_t *j, int shard, int which, void *out, int64_t capacity, int64_t *bytes)
{
    auto *job = reinterpret_cast<ure::ure_job *>(j);
    URE_ARG(job && shard >= 0 && shard < (int)job->host.size() && bytes);
    if (!job->index) return fail(-1, "ure_job_index_read: the job is not in touch_mode 3");
    const ure_shard_t &S = job->host[(size_t)shard];
    const shard_aux &A = job->aux_host[(size_t)shard];
    URE_HIP(hipDeviceSynchronize());
    uint32_t n_items = 0, total = 0;
    URE_HIP(hipMemcpy(&n_items, A.step_item + A.steps, 4, hipMemcpyDeviceToHost));
    URE_HIP(hipMemcpy(&total, A.step_begin + A.steps, 4, hipMemcpyDeviceToHost));
    const void *src = nullptr;
    size_t n = 0;
    switch (which) {
        case 0: src = A.step_begin; n = ((size_t)A.steps + 1) * 4; break;
        case 1: src = A.step_item; n = ((size_t)A.steps + 1) * 4; break;
        case 2: src = A.items; n = (size_t)n_items * 16; break;
        case 3: src = A.sslot; n = (size_t)total * 16; break;
        case 4: src = A.W; n = (size_t)A.idx_words * ((size_t)S.n_user + S.n_item) * 8; break;
        case 5: src = A.heavy_cnt; n = (size_t)A.steps * 4; break;
        case 6: src = A.heavy_cum; n = (size_t)A.steps * (kIdxHeavyMax + 1) * 4; break;
        default: return fail(-1, "ure_job_index_read: which = %d", which);
    }
    *bytes = (int64_t)n;
    if (!out) return 0;
    if ((int64_t)n > capacity) return fail(-1, "ure_job_index_read: %zu bytes do not fit %lld", n, (long long)capacity);
    URE_HIP(hipMemcpy(out, src, n, hipMemcpyDeviceToHost));
    return 0;
}

int ure_job_train(ure_job_t *j, int64_t tick0, int64_t tick1, void *stream)
{
    auto *job = reinterpret_cast<ure::ure_job *>(j);
    URE_ARG(job && tick0 >= 0 && tick1 >= tick0);
    if (tick0 != job->next_tick) return fail(-1, "ure_job_train: tick0=%lld but the job is at tick %lld (steps must run in order)", (long long)tick0, (long long)job->next_tick);
    tick1 = std::min(tick1, job->ticks);
    const int rc = train_ticks(job, tick0, tick1, static_cast<hipStream_t>(stream), nullptr, nullptr);
    if (rc == 0) job->next_tick = std::max(tick0, tick1);
    return rc;
}

int ure_job_train_profiled(ure_job_t *j, int64_t tick0, int64_t tick1, void *stream, double *step_ms, int64_t *n_step,
                           double *assign_ms, int64_t *n_assign)
{
    auto *job = reinterpret_cast<ure::ure_job *>(j);
    URE_ARG(job && tick0 >= 0 && tick1 >= tick0 && step_ms && n_step && assign_ms && n_assign);
    if (tick0 != job->next_tick) return fail(-1, "ure_job_train_profiled: tick0=%lld but the job is at tick %lld", (long long)tick0, (long long)job->next_tick);
    hipStream_t st = static_cast<hipStream_t>(stream);
    std::vector<hipEvent_t> se, ae;
    tick1 = std::min(tick1, job->ticks);
    int rc = train_ticks(job, tick0, tick1, st, &se, &ae);
    if (rc == 0) job->next_tick = std::max(tick0, tick1);
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
