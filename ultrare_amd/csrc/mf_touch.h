// mf_touch.h -- "touch mode" of the step kernel: the HBM-bound regime (BASELINE.json configs[3]:
// 32 shards x 113.7 MB of tables per GPU), where most rows of a shard have NO interaction in a given
// optimizer step and the dense optimizer of the reference (scratch.py:64-69: weight decay + momentum
// touch every row every step) would stream them through HBM just to let them decay.
//
// Between two gradient events a row's (w, m) evolves by the optimizer's own linear recurrence
//     m' = mu m + lam w ;  w' = w - lr m'      <=>   (w, m)' = A (w, m),  A = [[1 - lr lam, -lr mu], [lam, mu]]
// so j steps without a gradient are one 2x2 map A^j (evaluated in double on the host, one table per
// epoch's learning rate: StepLR changes it between epochs only).  Touch mode keeps every row in
// NEXT-TOUCH FORM: the stored (w, m) are valid for the step at which the row has its next interaction
// (or the end of the epoch).  Then
//   * a step visits only the rows it trains: a 64-bit mask per row says in which steps of the current
//     WINDOW of 64 steps the row has interactions.  An epoch of up to 64 steps is one window (24..27 steps
//     at configs[3]); longer epochs (full MF at 25 M rows: 750 steps, config.py:182-188) are cut into
//     windows, and every window start does what an epoch start does (below);
//   * whoever gathers a row reads w only, with no time stamp and no catching up: a row is gathered in
//     exactly the steps in which it is trained itself (interaction (u, i) of step s makes u gather i
//     and i gather u in step s), and for those steps its stored w is current by construction;
//   * after its update the owner advances the row by the table entry of the gap to its next step.
// A row's w alternates between the two table buffers with each of its OWN steps (its k-th step of the
// epoch reads buffer k & 1 and writes the other).  Which buffer a GATHERED row is in at step s -- the
// parity of its steps before s -- is put into bit 15 of the slot's batch tag once per epoch, so a gather
// costs no extra memory level.
//
// Work: rows longer than one scan pass (8 * LPR slots) keep the work units of mf_step (ure_host_build_units
// over the first n_multi rows of the schedule).  The others -- most item rows, the light users; 96 % of the
// rows, trained in 13..60 % of the steps -- are CANDIDATES: a workgroup looks at 256 of them with one lane
// each (mask in schedule order, coalesced), compacts the rows trained in this step into LDS, and its
// lane groups work the compacted list off.  (One lane group per row with an early exit was measured first:
// 125 k workgroups per launch whose fixed cost -- two dependent loads and an exit -- set the launch time,
// 0.94 ms against 1.03 ms for the dense kernel; profiles/r02/NOTES.md.)
//
// At a window start two launches (B) build the row masks of the window from the epoch's batch tags -- a row's
// slots belong to one workgroup, so its mask is combined in registers / LDS and stored once, in row order and
// in work order, without atomics or a clearing pass (round 2 had a third launch for that) -- and (C) put the
// buffer bits into the tags of the window's slots and bring every row from "valid at the window boundary" to
// "valid at its first step of the window", into buffer 0.
//
// Included by mf_train.hip after its constants (kQueue, kSegPerLane, lanes_per_row) and helpers.
#pragma once
#include "tag_prep.h"

namespace ure {

constexpr int kTouchMaxSteps = 32000;       // steps per epoch: the step number shares the 16-bit tag with the buffer bit (padding slots read 0x7FFF)
constexpr int kTouchTab = kTouchWindow + 1;
constexpr unsigned kTagStep = 0x7FFFu;      // touch mode: bits 0..14 of a tag = the step, bit 15 = buffer of the gathered row

// Where an optimizer step stands in touch mode.
struct TouchPos {
    int epoch, s;       // epoch, step of the epoch
    int win, sl;        // window of the epoch, step inside the window (the bit of the row masks)
    int wlen;           // steps of this window
    int64_t gwin;       // global window index (parity = which mask buffer)
};
__device__ __forceinline__ TouchPos touch_pos(const shard_aux &A, int epoch, int s)
{
    TouchPos p;
    p.epoch = epoch; p.s = s;
    p.win = s >> kTouchWindowBits; p.sl = s & (kTouchWindow - 1);
    p.wlen = min(kTouchWindow, A.steps - (p.win << kTouchWindowBits));
    p.gwin = (int64_t)epoch * A.windows + p.win;
    return p;
}

__device__ __forceinline__ unsigned long long mask_below(int s) { return s >= 64 ? ~0ull : ((1ull << s) - 1ull); }
__device__ __forceinline__ int mask_rank_parity(unsigned long long mk, int s) { return __popcll(mk & mask_below(s)) & 1; }

// (w, m) <- P (w, m), P = {p11, p12, p21, p22}
// touch_mode 2 -- MASKS ONE EPOCH AHEAD (epochs of at most 63 steps, tables read at the end of training only).
// The dense pass that brings every row to its first step of a new epoch (launch C, part 2: 6 % of the device time at configs[3])
// disappears when a row's owner knows, at its LAST step of epoch e, the row's first step of epoch e + 1: it then advances the row
// across the boundary itself -- to the epoch's end with epoch e's table (that value is the end-of-epoch snapshot), on to the next first
// step with epoch e + 1's.  For that the batch tags are prepared TWO epochs ahead (tag_prep.h: three tag buffers), and at the start
// of epoch e launch B builds the masks of epoch e + 1 (row order + work order, by epoch parity) and hands the owners of epoch e each
// row's next first step; launch C marks the buffer bits of epoch e + 1's tags.  A row's weights are no longer normalised into buffer 0
// at an epoch start: bit 63 of its mask word says in which buffer they are when the epoch starts (cumulative parity of its steps).
// A row without a step in epoch e (an "orphan": only a permutation that misses interactions leaves one -- every interaction trains once
// per epoch) is valid at the END of e; launch B of that epoch start carries it on to its first step of e + 1 (and writes its snapshot of e).
// Only the very first epoch needs the dense pass.
// The arithmetic is that of mode 1 -- the same two table entries applied to the same values -- so both modes agree to the last bit.
constexpr unsigned long long kStartBit = 1ull << 63;
constexpr unsigned long long kStepBits = ~kStartBit;
constexpr int kAheadMaxSteps = 63;
constexpr int kNoNext = 255;

struct AheadWork {              // the work-order arrays of one epoch parity
    unsigned long long *unit_mask, *unit_own, *sched_mask;
};
__device__ __forceinline__ AheadWork ahead_work(const shard_aux &A, int parity)
{
    unsigned long long *b = A.ahead_masks[parity & 1];
    return AheadWork{b, b + A.n_um, b + 2 * (size_t)A.n_um};
}
__device__ __forceinline__ int ahead_buffer_at(unsigned long long word, int s) { return (int)(word >> 63) ^ (__popcll(word & kStepBits & mask_below(s)) & 1); }

template <int V4>
__device__ __forceinline__ void row_advance(RowVec<V4> &w, RowVec<V4> &m, const float4 P)
{
#pragma unroll
    for (int i = 0; i < V4; ++i) {
        float4 a = w.q[i], b = m.q[i], nw, nm;
        nw.x = fmaf(P.x, a.x, __fmul_rn(P.y, b.x)); nw.y = fmaf(P.x, a.y, __fmul_rn(P.y, b.y));
        nw.z = fmaf(P.x, a.z, __fmul_rn(P.y, b.z)); nw.w = fmaf(P.x, a.w, __fmul_rn(P.y, b.w));
        nm.x = fmaf(P.z, a.x, __fmul_rn(P.w, b.x)); nm.y = fmaf(P.z, a.y, __fmul_rn(P.w, b.y));
        nm.z = fmaf(P.z, a.z, __fmul_rn(P.w, b.z)); nm.w = fmaf(P.z, a.w, __fmul_rn(P.w, b.w));
        w.q[i] = nw;
        m.q[i] = nm;
    }
}

// Workgroups of a shard's row-structured passes (mask building, tag bits): [0, nbU) take the work units of
// the multi-pass rows, [nbU, nbU + nbG) take 256 / LPR single-pass rows each, one lane group per row.
// -> the lane group's piece {row id or -1, first slot, end slot, unit index or -1}; *sched_idx = the row's
// place among the single-pass rows (or -1).
template <int LPR>
__device__ __forceinline__ int4 touch_piece(const ure_shard_t &S, int wg, int *sched_idx, int *unit_word = nullptr)
{
    constexpr int UPB = kBlock / LPR;
    const int nbU = S.n_units / UPB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int local = wave * (kWave / LPR) + lane / LPR;
    *sched_idx = -1;
    if (wg < nbU) {
        const int4 du = ldg_i4(S.units + 4 * ((size_t)wg * UPB + local));
        if (unit_word) *unit_word = du.w;
        return make_int4(du.x, du.y, du.z, wg * UPB + local);
    }
    const int rel = (wg - nbU) * UPB + local;
    if (rel >= S.n_active - S.n_multi) return make_int4(-1, 0, 0, -1);
    *sched_idx = rel;
    const int4 sc = ldg_i4(S.sched + 4 * (size_t)(S.n_multi + rel));
    return make_int4(sc.x, sc.y, sc.z, -1);
}
template <int LPR>
__host__ __device__ inline int touch_piece_blocks(int n_units, int n_active, int n_multi)
{
    constexpr int UPB = kBlock / LPR;
    return n_units / UPB + (n_active - n_multi + UPB - 1) / UPB;
}

// ---- window start, launch B: the step mask of every row from the epoch's batch tags, stored in row order
// (A.mask, for whoever gathers the row) and in work order (unit_mask / sched_mask, for the step kernel).
// OR over the LPR lanes of a group (every lane ends with the result)
template <int LPR>
__device__ __forceinline__ unsigned long long group_or(unsigned long long v)
{
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    if (LPR >= 2) { lo |= (unsigned)dpp_i<kDppQuadXor1>((int)lo); hi |= (unsigned)dpp_i<kDppQuadXor1>((int)hi); }
    if (LPR >= 4) { lo |= (unsigned)dpp_i<kDppQuadXor2>((int)lo); hi |= (unsigned)dpp_i<kDppQuadXor2>((int)hi); }
    if (LPR >= 8) { lo |= (unsigned)dpp_i<kDppHalfMirror>((int)lo); hi |= (unsigned)dpp_i<kDppHalfMirror>((int)hi); }
    if (LPR >= 16) { lo |= (unsigned)dpp_i<kDppRowMirror>((int)lo); hi |= (unsigned)dpp_i<kDppRowMirror>((int)hi); }
    if (LPR >= 32) { lo |= (unsigned)__shfl_xor((int)lo, 16, kWave); hi |= (unsigned)__shfl_xor((int)hi, 16, kWave); }
    if (LPR >= 64) { lo |= (unsigned)__shfl_xor((int)lo, 32, kWave); hi |= (unsigned)__shfl_xor((int)hi, 32, kWave); }
    return ((unsigned long long)hi << 32) | lo;
}

template <int LPR>
__device__ __forceinline__ void touch_build_masks(const ure_shard_t &S, const shard_aux &A, const TouchPos &P, int wg, unsigned long long *wg_mask)
{
    constexpr int CAP = kSegPerLane * LPR;
    constexpr int UPB = kBlock / LPR;
    int si, uw = 0;
    const int4 pc = touch_piece<LPR>(S, wg, &si, &uw);
    const bool in_units = wg < S.n_units / UPB;              // workgroup-uniform
    const int sub = threadIdx.x & (LPR - 1);
    const int local = (int)threadIdx.x / LPR;
    const uint16_t *__restrict__ ent_tag = S.ent_tag + (size_t)(P.epoch & 1) * S.n_slots;
    unsigned long long mk = 0;
    if (pc.x >= 0) {
        // a unit of several scan passes (the pieces of the heaviest rows: 120 k slots each for the top item of the 25 M set) also
        // records which steps every single pass holds: with 750 steps per epoch five passes in six hold none of a given step
        const bool per_pass = A.pass_mask != nullptr && pc.z - pc.y > CAP;
        for (int seg = pc.y; seg < pc.z; seg += CAP) {          // (lane-group uniform: every lane of the group stays for the pass masks)
            const int p0 = seg + sub * kSegPerLane;
            unsigned long long bits = 0;
            if (p0 < pc.z) {
                const uint4 t4 = ldg_u4(ent_tag + p0);
                const unsigned tw[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const unsigned st = (tw[k >> 1] >> ((k & 1) * 16)) & kTagStep;      // (slots of earlier windows carry their buffer bit already)
                    if ((int)(st >> kTouchWindowBits) == P.win) bits |= 1ull << (st & (kTouchWindow - 1));
                }
            }
            mk |= bits;
            if (per_pass) {
                const unsigned long long pm = group_or<LPR>(bits);
                if (sub == 0) stg(A.pass_mask + (seg >> 3), pm);
            }
        }
    }
    mk = group_or<LPR>(mk);
    if (!in_units) {
        // a single-pass row: this lane group has seen all of its slots
        if (pc.x >= 0 && sub == 0) {
            stg(A.mask[P.gwin & 1] + pc.x, mk);
            stg(A.sched_mask + si, mk);
        }
        return;
    }
    // work units: the units of a row sit side by side in this workgroup; the row's first unit combines them.  The unit's OWN
    // mask is kept as well: in a step that trains the row through other units only, this unit has nothing to scan -- with 750
    // steps per epoch (full MF at 25 M rows) a unit of 128 slots has an interaction in one step of six, and the scan of all the
    // others was 450 MB of tag / id / rating loads per step
    const int leader = uw & 0xFFFF, count = (uw >> 16) & 0x3FFF;
    if (pc.x >= 0 && sub == 0) stg(A.unit_own + pc.w, mk);
    if (sub == 0) wg_mask[local] = mk;
    __syncthreads();
    if (pc.x >= 0 && sub == 0 && local == leader) {
        unsigned long long all = mk;
        for (int k = 1; k < count; ++k) all |= wg_mask[leader + k];
        stg(A.mask[P.gwin & 1] + pc.x, all);
        wg_mask[leader] = all;
    }
    __syncthreads();
    if (pc.x >= 0 && sub == 0) stg(A.unit_mask + pc.w, wg_mask[leader]);
}

// ---- window start, launch C, part 1: bit 15 of the tag of every slot trained in this window = the buffer the slot's
// OTHER row is in at the slot's step (parity of that row's steps of the window before it)
template <int LPR>
__device__ __forceinline__ void touch_mark_tags(const ure_shard_t &S, const shard_aux &A, const TouchPos &P, int wg)
{
    constexpr int CAP = kSegPerLane * LPR;
    int si;
    const int4 pc = touch_piece<LPR>(S, wg, &si);
    if (pc.x < 0) return;
    const int sub = threadIdx.x & (LPR - 1);
    const unsigned long long *__restrict__ masks = A.mask[P.gwin & 1];
    uint16_t *__restrict__ ent_tag = S.ent_tag + (size_t)(P.epoch & 1) * S.n_slots;
    const int other_base = pc.x < S.n_user ? S.n_user : 0;
    for (int p0 = pc.y + sub * kSegPerLane; p0 < pc.z; p0 += CAP) {
        const uint4 t4 = ldg_u4(ent_tag + p0);
        const int4 o0 = ldg_i4(S.ent_oid + p0), o1 = ldg_i4(S.ent_oid + p0 + 4);
        const unsigned tw[4] = {t4.x, t4.y, t4.z, t4.w};
        const int ov[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
        unsigned out[8];
        bool any = false;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned tg = (tw[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu;
            const unsigned st = tg & kTagStep;
            out[k] = tg;
            if ((int)(st >> kTouchWindowBits) == P.win) {
                out[k] = st | ((unsigned)mask_rank_parity(ldg(masks + other_base + ov[k]), (int)(st & (kTouchWindow - 1))) << 15);
                any = true;
            }
        }
        if (any) stg_u4(ent_tag + p0, make_uint4(out[0] | (out[1] << 16), out[2] | (out[3] << 16), out[4] | (out[5] << 16), out[6] | (out[7] << 16)));
    }
}

// ---- touch_mode 2, launch B at the start of epoch e = e_next - 1 (and once more at tick 0 for e_next = 0): the masks of epoch e_next
// from its tags, the owners' hand-over for epoch e, the orphans of epoch e.  has_next = false: there is no epoch e_next.
template <int LPR, int V4>
__device__ __forceinline__ void touch_ahead_build(const ure_shard_t &S, const shard_aux &A, int e_next, bool has_next, int wg, unsigned long long *wg_mask)
{
    constexpr int CAP = kSegPerLane * LPR;
    constexpr int UPB = kBlock / LPR;
    constexpr int D = LPR * V4 * 4;
    int si, uw = 0;
    const int4 pc = touch_piece<LPR>(S, wg, &si, &uw);
    const bool in_units = wg < S.n_units / UPB;              // workgroup-uniform
    const int sub = threadIdx.x & (LPR - 1);
    const int local = (int)threadIdx.x / LPR;
    unsigned long long mk = 0;
    if (has_next && pc.x >= 0) {
        const uint16_t *__restrict__ ent_tag = S.ent_tag + tag_buffer(S, e_next);
        for (int p0 = pc.y + sub * kSegPerLane; p0 < pc.z; p0 += CAP) {
            const uint4 t4 = ldg_u4(ent_tag + p0);
            const unsigned tw[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned st = (tw[k >> 1] >> ((k & 1) * 16)) & kTagStep;      // (padding slots read 0x7FFF)
                if (st < (unsigned)kAheadMaxSteps) mk |= 1ull << st;
            }
        }
    }
    mk = group_or<LPR>(mk);
    const unsigned long long own = mk;
    bool row_leader = pc.x >= 0;
    if (in_units) {
        const int leader = uw & 0xFFFF, count = (uw >> 16) & 0x3FFF;
        if (sub == 0) wg_mask[local] = mk;
        __syncthreads();
        if (pc.x >= 0 && sub == 0 && local == leader) {
            unsigned long long all = mk;
            for (int k = 1; k < count; ++k) all |= wg_mask[leader + k];
            wg_mask[leader] = all;
        }
        __syncthreads();
        if (pc.x >= 0) mk = wg_mask[leader];
        row_leader = pc.x >= 0 && local == leader;
    }
    if (pc.x < 0) return;
    // the row's word of the epoch that starts now (e = e_next - 1): where its weights are, and whether it has a step in e at all
    const unsigned long long prev = e_next >= 1 ? ldg(A.mask[(e_next - 1) & 1] + pc.x) : 0ull;
    const int start_e = (int)(prev >> 63);
    const bool orphan = e_next >= 1 && (prev & kStepBits) == 0;
    const unsigned long long start_next = e_next >= 1 ? (unsigned long long)(start_e ^ (__popcll(prev & kStepBits) & 1)) : 0ull;
    const unsigned long long word = mk | (start_next << 63);
    const int nf = mk ? __ffsll((long long)mk) - 1 : kNoNext;
    if (sub == 0) {
        const AheadWork W = ahead_work(A, e_next);
        if (row_leader) stg(A.mask[e_next & 1] + pc.x, word);
        if (in_units) {
            stg(W.unit_mask + pc.w, word);
            stg(W.unit_own + pc.w, own);
            stg(A.unit_nf + pc.w, (uint8_t)nf);
        } else {
            stg(W.sched_mask + si, word);
            stg(A.sched_nf + si, (uint8_t)nf);
        }
    }
    // an orphan of epoch e is valid at the END of e (whoever advanced it last found no step in e): that value is its snapshot of e,
    // and it is carried on to its first step of e_next (or to e_next's end) here, in place -- nobody gathers it during e
    if (orphan && row_leader) {
        const bool is_user = pc.x < S.n_user;
        const size_t off = (size_t)(is_user ? pc.x : pc.x - S.n_user) * D;
        float *wrow = (is_user ? S.U[start_e] : S.V[start_e]) + off, *mrow = (is_user ? S.mU : S.mV) + off;
        RowVec<V4> w = row_load<LPR, V4>(wrow, sub), m = row_load<LPR, V4>(mrow, sub);
        if (S.snap && S.row_slot) {
            const int slot = ldg(S.row_slot + pc.x);
            if (slot >= 0) row_store<LPR, V4>(S.snap + ((size_t)(e_next - 1) * S.n_active + slot) * D, sub, w);
        }
        if (has_next) {
            const int gap = mk ? nf : A.steps;
            if (gap > 0) {
                row_advance<V4>(w, m, A.ptab[(size_t)e_next * kTouchTab + gap]);
                row_store<LPR, V4>(wrow, sub, w);
                row_store<LPR, V4>(mrow, sub, m);
            }
        }
    }
}

// ---- the same for 256 single-pass rows of at most 8 slots each ("tiny": 73 % of the rows of a configs[3] shard), ONE LANE per row
// instead of a lane group: the schedule entries, the tags and the work-order stores of consecutive lanes are consecutive in memory,
// and a wavefront covers 64 rows where it covered 64 / LPR.  rel0 = the first row's place among the single-pass rows.
template <int V4LPR>
__device__ __forceinline__ void touch_ahead_build_tiny(const ure_shard_t &S, const shard_aux &A, int e_next, bool has_next, int rel0)
{
    const int rel = rel0 + (int)threadIdx.x;
    if (rel >= S.n_active - S.n_multi) return;
    const int4 sc = ldg_i4(S.sched + 4 * (size_t)(S.n_multi + rel));           // {row id, first slot, end slot (= first + 8), nnz}
    unsigned long long mk = 0;
    if (has_next) {
        const uint4 t4 = ldg_u4(S.ent_tag + tag_buffer(S, e_next) + sc.y);
        const unsigned tw[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned st = (tw[k >> 1] >> ((k & 1) * 16)) & kTagStep;
            if (st < (unsigned)kAheadMaxSteps) mk |= 1ull << st;
        }
    }
    const unsigned long long prev = e_next >= 1 ? ldg(A.mask[(e_next - 1) & 1] + sc.x) : 0ull;
    const int start_e = (int)(prev >> 63);
    const unsigned long long start_next = e_next >= 1 ? (unsigned long long)(start_e ^ (__popcll(prev & kStepBits) & 1)) : 0ull;
    const unsigned long long word = mk | (start_next << 63);
    const int nf = mk ? __ffsll((long long)mk) - 1 : kNoNext;
    stg(A.mask[e_next & 1] + sc.x, word);
    stg(ahead_work(A, e_next).sched_mask + rel, word);
    stg(A.sched_nf + rel, (uint8_t)nf);
    if (e_next >= 1 && (prev & kStepBits) == 0) {
        // an orphan (only a permutation that misses interactions leaves a row without a step): the lane carries the row on by itself
        const int D = S.d;
        const bool is_user = sc.x < S.n_user;
        const size_t off = (size_t)(is_user ? sc.x : sc.x - S.n_user) * D;
        float *wrow = (is_user ? S.U[start_e] : S.V[start_e]) + off, *mrow = (is_user ? S.mU : S.mV) + off;
        const int slot = (S.snap && S.row_slot) ? ldg(S.row_slot + sc.x) : -1;
        const int gap = mk ? nf : A.steps;
        for (int c = 0; c < D; c += 4) {
            RowVec<1> w, m;
            w.q[0] = ldg_f4(wrow + c);
            m.q[0] = ldg_f4(mrow + c);
            if (slot >= 0) stg_f4(S.snap + ((size_t)(e_next - 1) * S.n_active + slot) * D + c, w.q[0]);
            if (has_next && gap > 0) {
                row_advance<1>(w, m, A.ptab[(size_t)e_next * kTouchTab + gap]);
                stg_f4(wrow + c, w.q[0]);
                stg_f4(mrow + c, m.q[0]);
            }
        }
    }
}

__device__ __forceinline__ void touch_ahead_mark_tiny(const ure_shard_t &S, const shard_aux &A, int e_next, int rel0)
{
    const int rel = rel0 + (int)threadIdx.x;
    if (rel >= S.n_active - S.n_multi) return;
    const int4 sc = ldg_i4(S.sched + 4 * (size_t)(S.n_multi + rel));
    const unsigned long long *__restrict__ masks = A.mask[e_next & 1];
    uint16_t *__restrict__ ent_tag = S.ent_tag + tag_buffer(S, e_next);
    const int other_base = sc.x < S.n_user ? S.n_user : 0;
    const uint4 t4 = ldg_u4(ent_tag + sc.y);
    const int4 o0 = ldg_i4(S.ent_oid + sc.y), o1 = ldg_i4(S.ent_oid + sc.y + 4);
    const unsigned tw[4] = {t4.x, t4.y, t4.z, t4.w};
    const int ov[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
    unsigned out[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const unsigned tg = (tw[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu;
        const unsigned st = tg & kTagStep;
        out[k] = tg;
        if (st < (unsigned)kAheadMaxSteps) out[k] = st | ((unsigned)ahead_buffer_at(ldg(masks + other_base + ov[k]), (int)st) << 15);
    }
    stg_u4(ent_tag + sc.y, make_uint4(out[0] | (out[1] << 16), out[2] | (out[3] << 16), out[4] | (out[5] << 16), out[6] | (out[7] << 16)));
}

// ---- touch_mode 2, launch C: bit 15 of every tag of epoch e_next = the buffer the slot's OTHER row is in at the slot's step
template <int LPR>
__device__ __forceinline__ void touch_ahead_mark(const ure_shard_t &S, const shard_aux &A, int e_next, int wg)
{
    constexpr int CAP = kSegPerLane * LPR;
    int si;
    const int4 pc = touch_piece<LPR>(S, wg, &si);
    if (pc.x < 0) return;
    const int sub = threadIdx.x & (LPR - 1);
    const unsigned long long *__restrict__ masks = A.mask[e_next & 1];
    uint16_t *__restrict__ ent_tag = S.ent_tag + tag_buffer(S, e_next);
    const int other_base = pc.x < S.n_user ? S.n_user : 0;
    for (int p0 = pc.y + sub * kSegPerLane; p0 < pc.z; p0 += CAP) {
        const uint4 t4 = ldg_u4(ent_tag + p0);
        const int4 o0 = ldg_i4(S.ent_oid + p0), o1 = ldg_i4(S.ent_oid + p0 + 4);
        const unsigned tw[4] = {t4.x, t4.y, t4.z, t4.w};
        const int ov[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
        unsigned out[8];
        bool any = false;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned tg = (tw[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu;
            const unsigned st = tg & kTagStep;
            out[k] = tg;
            if (st < (unsigned)kAheadMaxSteps) {
                out[k] = st | ((unsigned)ahead_buffer_at(ldg(masks + other_base + ov[k]), (int)st) << 15);
                any = true;
            }
        }
        if (any) stg_u4(ent_tag + p0, make_uint4(out[0] | (out[1] << 16), out[2] | (out[3] << 16), out[4] | (out[5] << 16), out[6] | (out[7] << 16)));
    }
}

// ---- window start, launch C, part 2: every active row from "valid at the window boundary" (buffer = parity of
// its number of steps in the window that ended) to "valid at its first step of this window", in buffer 0.
// One lane per float4 of a row.
__device__ __forceinline__ void touch_advance_rows(const ure_shard_t &S, const shard_aux &A, const TouchPos &P, int blk, int n_blk)
{
    const int d4 = S.d / 4;
    const int64_t total = (int64_t)S.n_active * d4;
    const unsigned long long *__restrict__ m_new = A.mask[P.gwin & 1];
    const unsigned long long *__restrict__ m_old = A.mask[(P.gwin & 1) ^ 1];      // (all zero before the first window: buffer 0)
    const float4 *__restrict__ tab = A.ptab + (size_t)P.epoch * kTouchTab;
    for (int64_t t = (int64_t)blk * kBlock + threadIdx.x; t < total; t += (int64_t)n_blk * kBlock) {
        const int idx = (int)(t / d4), c4 = (int)(t % d4);
        const int row_id = ldg(S.sched + 4 * (size_t)idx);
        const unsigned long long mo = ldg(m_old + row_id), mn = ldg(m_new + row_id);
        const int from = __popcll(mo) & 1;
        const int j = mn ? __ffsll((long long)mn) - 1 : P.wlen;           // steps that pass before the row's first own step (none: to the window's end)
        const bool is_user = row_id < S.n_user;
        const size_t o = (size_t)(is_user ? row_id : row_id - S.n_user) * S.d + (size_t)c4 * 4;
        float *wsrc = (is_user ? S.U[from] : S.V[from]) + o, *wdst = (is_user ? S.U[0] : S.V[0]) + o;
        float *mom = (is_user ? S.mU : S.mV) + o;
        if (from != 0 || j != 0) {
            RowVec<1> w, m;
            w.q[0] = ldg_f4(wsrc);
            m.q[0] = ldg_f4(mom);
            if (j != 0) row_advance<1>(w, m, tab[j]);
            stg_f4(wdst, w.q[0]);
            if (j != 0) stg_f4(mom, m.q[0]);
        }
    }
}

// ---- one row (or one work unit of a multi-pass row) of a step in touch mode: the scan, compaction and gather
// loop of mf_step (mf_train.hip); what differs is marked [touch].  du = {row id, first slot, end slot, leader |
// count << 16 | multi << 30}, mk = the row's step mask, hit = the row is trained in step s (lane-group uniform).
template <int LPR, int V4>
__device__ __forceinline__ void touch_process(const ure_shard_t &S, const shard_aux &A, const int4 du, const unsigned long long mk, const bool hit,
                                              const bool scan, const int nf, const int local, const TouchPos &P, const float lr, int *qo, float *qr,
                                              float (*q_r)[kQueue], float4 (*part_acc)[V4][LPR])
{
    const int epoch = P.epoch, s = P.s;
    const bool ahead = S.touch_mode == 2;        // [mode 2] bit 63 of mk = the buffer at the epoch's start, nf = the row's first step of epoch + 1
    constexpr int D = LPR * V4 * 4;
    using Row = RowVec<V4>;
    constexpr int G = kWave / LPR;
    constexpr int CAP = kSegPerLane * LPR;
    constexpr int kGB = URE_TOUCH_KGB;
    const int lane = threadIdx.x & 63;
    const int sub = lane & (LPR - 1), grp = lane / LPR;
    const float lam = S.lam, mu = S.mu;
    const int32_t *__restrict__ ent_oid = S.ent_oid;
    const float *__restrict__ ent_r = S.ent_r;
    const uint16_t *__restrict__ ent_tag = S.ent_tag + tag_buffer(S, epoch);
    const int leader = du.w & 0xFFFF, count = (du.w >> 16) & 0x3FFF;
    const bool multi = (du.w >> 30) & 1;
    const bool owner = hit && local == leader;
    const bool is_user = du.x < S.n_user;
    const int row = is_user ? du.x : du.x - S.n_user;
    const size_t row_off = (size_t)(hit ? row : 0) * D;
    // [touch] the row's k-th step of the window reads buffer k & 1 (next-touch form: w is valid for THIS step)
    const unsigned long long step_bits = ahead ? mk & kStepBits : mk;
    const int buf = ahead ? ahead_buffer_at(mk, P.sl) : mask_rank_parity(mk, P.sl);
    Row w = row_zero<V4>(), acc = w, m4 = w;
    float *mom = (is_user ? S.mU : S.mV) + row_off;
    if (hit) w = row_load<LPR, V4>((is_user ? S.U[buf] : S.V[buf]) + row_off, sub);
    if (owner) m4 = row_load<LPR, V4>(mom, sub);       // requested with the row: no memory level of its own at the end
    float sse = 0.f;
    float *const *other_tab = is_user ? S.V : S.U;
    const float *__restrict__ other0 = other_tab[0];
    const float *__restrict__ other1 = other_tab[1];
    int *gq = qo + grp * CAP;
    float *gr = qr + grp * CAP;
    const int beg = du.y, end = (hit && scan) ? du.z : du.y;      // scan: this unit's own slots have an interaction in the step
    // a unit of several passes under per-pass masks (tag_prep.h: pass_mask): its lane group looks LPR passes ahead -- a lane per pass,
    // one coalesced load -- and visits only the passes that hold a slot of this step
    const bool by_pass = A.pass_mask != nullptr && end - beg > CAP;
    int look = beg;                      // first pass not yet looked up
    int look0 = beg;                     // first pass of the batch `pending` describes
    unsigned long long pending = 0;      // passes of that batch still to visit (bit = pass of the batch)
    auto next_pass = [&]() -> int {
        while (true) {
            if (pending) {
                const int bit = __ffsll((long long)pending) - 1;
                pending &= pending - 1;
                return look0 + bit * CAP;
            }
            if (look >= end) return end;
            const int mine = look + sub * CAP;
            const unsigned long long pm = mine < end ? ldg(A.pass_mask + (mine >> 3)) : 0ull;
            pending = group_or<LPR>(((pm >> P.sl) & 1ull) << sub);
            look0 = look;
            look += LPR * CAP;
        }
    };
    for (int seg = by_pass ? next_pass() : beg;; seg = by_pass ? next_pass() : seg + CAP) {
        if (!__any(seg < end)) break;
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
        int ov[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
        const float rv[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
        unsigned mb = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned tg = (tw[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu;
            mb |= ((tg & kTagStep) == (unsigned)s ? 1u : 0u) << k;
            ov[k] |= (int)((tg >> 15) << 31);            // [touch] the gathered row's buffer rides in the id's top bit
        }
        const int c = __popc(mb);
        const int inc = group_scan<LPR>(c, sub);
        const int qn = (int)group_sum<LPR>((float)c);
        int mpos = inc - c, upos = qn + sub * kSegPerLane - mpos;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const bool h = (mb >> k) & 1u;
            const int pos = ((h ? mpos : upos) + grp) & (CAP - 1);
            gq[pos] = ov[k];
            gr[pos] = rv[k];
            mpos += h ? 1 : 0;
            upos += h ? 0 : 1;
        }
        __builtin_amdgcn_wave_barrier();
        for (int t0 = 0; __any(t0 < qn); t0 += kGB) {
            int o[kGB];
            float r[kGB];
            bool act[kGB];
            Row v[kGB];
#pragma unroll
            for (int k = 0; k < kGB; ++k) {
                const int qi = (min(t0 + k, CAP - 1) + grp) & (CAP - 1);
                act[k] = t0 + k < qn;
                o[k] = act[k] ? gq[qi] : 0;
                r[k] = gr[qi];
            }
#pragma unroll
            for (int k = 0; k < kGB; ++k)
                v[k] = row_load<LPR, V4>((o[k] < 0 ? other1 : other0) + (size_t)(o[k] & 0x7FFFFFFF) * D, sub);
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
    if (multi) {
        if (hit && count > 1) {
#pragma unroll
            for (int i = 0; i < V4; ++i) part_acc[local][i][sub] = acc.q[i];
            if (sub == 0) qr[grp * CAP] = sse;
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
    if (owner) {
        Row gr2, nr;
#pragma unroll
        for (int i = 0; i < V4; ++i) {                   // torch.optim.SGD: g += lam w ; buf = mu buf + g ; w -= lr buf
            const float4 ww = w.q[i], mm = m4.q[i], aa = acc.q[i];
            float4 g, wn;
            g.x = fmaf(lam, ww.x, aa.x); g.y = fmaf(lam, ww.y, aa.y); g.z = fmaf(lam, ww.z, aa.z); g.w = fmaf(lam, ww.w, aa.w);
            g.x = __fadd_rn(__fmul_rn(mu, mm.x), g.x); g.y = __fadd_rn(__fmul_rn(mu, mm.y), g.y);
            g.z = __fadd_rn(__fmul_rn(mu, mm.z), g.z); g.w = __fadd_rn(__fmul_rn(mu, mm.w), g.w);
            wn.x = fmaf(-lr, g.x, ww.x); wn.y = fmaf(-lr, g.y, ww.y); wn.z = fmaf(-lr, g.z, ww.z); wn.w = fmaf(-lr, g.w, ww.w);
            gr2.q[i] = g;
            nr.q[i] = wn;
        }
        // [touch] bring the row to its next own step of the window (or to the window's end): the steps in between
        // apply weight decay and momentum only, one 2x2 map for all of them
        const unsigned long long rest = P.sl + 1 < 64 ? step_bits >> (P.sl + 1) : 0ull;
        const int gap = rest ? __ffsll((long long)rest) - 1 : P.wlen - 1 - P.sl;
        if (gap > 0) row_advance<V4>(nr, gr2, A.ptab[(size_t)epoch * kTouchTab + gap]);
        if (ahead && !rest) {
            // [mode 2] the row's last step of the epoch: nr is now valid at the epoch's end -- its snapshot -- and goes on to its first
            // step of the next epoch (none there: to that epoch's end, where launch B of its start picks it up as an orphan)
            if (S.snap && S.row_slot) {
                const int slot = ldg(S.row_slot + du.x);
                if (slot >= 0) row_store<LPR, V4>(S.snap + ((size_t)epoch * S.n_active + slot) * D, sub, nr);
            }
            if (epoch + 1 < S.epochs) {
                const int g2 = nf != kNoNext ? nf : A.steps;
                if (g2 > 0) row_advance<V4>(nr, gr2, A.ptab[(size_t)(epoch + 1) * kTouchTab + g2]);
            }
        }
        row_store<LPR, V4>(mom, sub, gr2);
        row_store<LPR, V4>((is_user ? S.U[buf ^ 1] : S.V[buf ^ 1]) + row_off, sub, nr);
        if (is_user && sub == 0 && sse != 0.f) {
            float *slot = S.sse + (size_t)epoch * S.n_user + row;
            stg(slot, ldg(slot) + sse);
        }
    }
}

// ---- one optimizer step in touch mode.  Workgroups of a shard: [0, nbU) work units of the multi-pass rows |
// [nbU, nbU + nbC) candidates, 256 single-pass rows each | the tag riders of the next epoch.
template <int LPR, int V4>
__device__ __forceinline__ void mf_touch_step(const ure_shard_t *__restrict__ shards, const shard_aux *__restrict__ aux, int64_t tick, int shard_fast)
{
    constexpr int G = kWave / LPR;
    constexpr int UPB = kBlock / LPR;
    constexpr int kQueueBytes = kWavesPerBlock * kQueue * 8;
    static_assert(kTagLds <= kQueueBytes, "tag phases must fit in the queue space");
    __shared__ __attribute__((aligned(16))) char lds_raw[kQueueBytes];
    __shared__ float4 part_acc[UPB][V4][LPR];
    __shared__ int4 cand_row[kBlock];                   // the candidates trained in this step: {row id, first slot, end slot, -}
    __shared__ unsigned long long cand_mask[kBlock];
    __shared__ int cand_count[kWavesPerBlock];
    int (*q_oid)[kQueue] = reinterpret_cast<int (*)[kQueue]>(lds_raw);
    float (*q_r)[kQueue] = reinterpret_cast<float (*)[kQueue]>(lds_raw + kWavesPerBlock * kQueue * 4);

    int shard_idx = (int)(shard_fast ? blockIdx.x : blockIdx.y);
    int wg = (int)(shard_fast ? blockIdx.y : blockIdx.x);
    if (shard_fast >> 1) {        // sliced mapping, see mf_step
        const unsigned n_sh = (unsigned)shard_fast >> 8;
        const unsigned x = blockIdx.x & 7u, j = blockIdx.x >> 3;
        const unsigned jq = j / n_sh, jr = j - jq * n_sh;
        const unsigned slice = n_sh * x + jr;
        shard_idx = (int)(slice >> 3);
        wg = (int)(jq * 8 + (slice & 7u));
    }
    const ure_shard_t &S = shards[shard_idx];
    const shard_aux &A = aux[shard_idx];
    const int steps = A.steps;
    if (tick >= (int64_t)steps * S.epochs) return;
    const int epoch = (int)epoch_of(A, tick);
    const int s = (int)(tick - (int64_t)epoch * steps);
    const TouchPos P = touch_pos(A, epoch, s);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int grp = lane / LPR;
    const float lr = ldg(S.lr + epoch);
    int *qo = q_oid[wave];
    float *qr = q_r[wave];
    const int local = wave * G + grp;

    const int nbU = S.n_units / UPB;
    const int n_single = S.n_active - S.n_multi;
    const int nbC = (n_single + kBlock - 1) / kBlock;
    const bool ahead = S.touch_mode == 2;
    // the masks in work order: mode 1 one set (the current window's), mode 2 one per epoch parity
    const unsigned long long *__restrict__ unit_mask = A.unit_mask, *__restrict__ unit_own = A.unit_own, *__restrict__ sched_mask = A.sched_mask;
    if (ahead) {
        const AheadWork W = ahead_work(A, epoch);
        unit_mask = W.unit_mask; unit_own = W.unit_own; sched_mask = W.sched_mask;
    }
    if (wg < nbU) {
        // ---- multi-pass rows: the unit's mask sits next to its descriptor (no dependent load)
        const size_t u = (size_t)wg * UPB + local;
        const int4 du = ldg_i4(S.units + 4 * u);
        const unsigned long long mk = ldg(unit_mask + u);
        const unsigned long long own = ldg(unit_own + u);
        const int nf = ahead ? (int)ldg(A.unit_nf + u) : 0;
        const bool hit = du.x >= 0 && ((mk >> P.sl) & 1ull);
        const bool multi = (du.w >> 30) & 1;
        if (!multi && !__any(hit)) return;
        touch_process<LPR, V4>(S, A, du, mk, hit, (own >> P.sl) & 1ull, nf, local, P, lr, qo, qr, q_r, part_acc);
        return;
    }
    if (wg < nbU + nbC) {
        // ---- candidates: one lane per row finds out whether the row is trained in this step ...
        const int rel = (wg - nbU) * kBlock + (int)threadIdx.x;
        // (the schedule entry is requested with the mask, not after it: one memory level, and both reads are coalesced)
        const unsigned long long mk = rel < n_single ? ldg(sched_mask + rel) : 0ull;
        int4 sc = rel < n_single ? ldg_i4(S.sched + 4 * (size_t)(S.n_multi + rel)) : make_int4(-1, 0, 0, 0);
        if (ahead && rel < n_single) sc.w = (int)ldg(A.sched_nf + rel);      // [mode 2] the row's first step of the next epoch rides with its entry
        const bool hit = (mk >> P.sl) & 1ull;
        const unsigned long long vote = __ballot(hit);
        if (lane == 0) cand_count[wave] = __popcll(vote);
        __syncthreads();
        int base = 0, total = 0;
#pragma unroll
        for (int k = 0; k < kWavesPerBlock; ++k) {
            const int c = cand_count[k];
            base += k < wave ? c : 0;
            total += c;
        }
        if (hit) {
            const int pos = base + __popcll(vote & ((1ull << lane) - 1ull));
            cand_row[pos] = sc;
            cand_mask[pos] = mk;
        }
        __syncthreads();
        // ... and the lane groups work the compacted list off, one row each per round
        for (int e0 = wave * G; e0 < total; e0 += UPB) {      // wave-uniform bound: a wave's groups take e0 .. e0 + G - 1
            const int e = e0 + grp;
            const bool have = e < total;
            int4 du = make_int4(-1, 0, 0, 0);
            unsigned long long rm = 0ull;
            if (have) { du = cand_row[e]; rm = cand_mask[e]; }
            const int nf = du.w;
            du.w = local | (1 << 16);                          // its own leader, one unit, no partial sums to combine
            touch_process<LPR, V4>(S, A, du, rm, have, true, nf, local, P, lr, qo, qr, q_r, part_acc);
        }
        return;
    }
    // ---- the tag riders of the next epoch, as in mf_step
    const int tgt = epoch + tag_ahead(S);                  // the epoch whose tags this one's steps prepare
    const TagRide ride = tag_ride(A, s, tgt < S.epochs);
    const int rb = wg - nbU - nbC;
    if (rb >= ride.count) return;
    if (ride.phase == 0) tag_partition(S, tgt, ride.first + rb, lds_raw);
    else if (ride.phase == 1) tag_collect(S, ride.first + rb, lds_raw);
    else tag_derive(S, tgt, ride.first + rb, A.derive_blocks);
}

// ---- reading the tables at an epoch boundary of the shard: the active rows' current w sits in the buffer
// given by the parity of their step count in the epoch that ended; copy it where the caller reads (`cur`)
__device__ __forceinline__ void touch_collect_rows(const ure_shard_t &S, const shard_aux &A, int last_epoch, int cur, int blk, int n_blk)
{
    const int d4 = S.d / 4;
    const int64_t total = (int64_t)S.n_active * d4;
    const unsigned long long *__restrict__ mk = A.mask[touch_last_window(A, last_epoch < 0 ? 0 : last_epoch) & 1];      // (touch_mode 3 keeps no such masks: NULL, not read)
    for (int64_t t = (int64_t)blk * kBlock + threadIdx.x; t < total; t += (int64_t)n_blk * kBlock) {
        const int idx = (int)(t / d4), c4 = (int)(t % d4);
        const int row_id = ldg(S.sched + 4 * (size_t)idx);
        const unsigned long long word = (last_epoch < 0 || S.touch_mode == 3) ? 0ull : ldg(mk + row_id);
        const int from = last_epoch < 0 ? 0 : S.touch_mode == 3 ? (int)ldg(A.end_par[last_epoch & 1] + row_id) :
                         (S.touch_mode == 2 ? ahead_buffer_at(word, 64) : (__popcll(word) & 1));
        if (from == cur) continue;
        const bool is_user = row_id < S.n_user;
        const size_t o = (size_t)(is_user ? row_id : row_id - S.n_user) * S.d + (size_t)c4 * 4;
        stg_f4((is_user ? S.U[cur] : S.V[cur]) + o, ldg_f4((is_user ? S.U[from] : S.V[from]) + o));
    }
}

}  // namespace ure
