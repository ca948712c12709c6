// Internal helpers shared by the HIP translation units of libultrare_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "ultrare_hip.h"

// Widest table row whose lanes hold ONE float4 each (d / 4 lanes per row); wider rows give every
// lane two (d / 8 lanes per row).  Shared by the step kernel and ure_host_build_units.
#ifndef URE_NARROW_MAX
#define URE_NARROW_MAX 32
#endif

namespace ure {

constexpr int kWave = 64;            // gfx950 wavefront
constexpr int kBlock = 256;          // 4 wavefronts per workgroup
constexpr int kWavesPerBlock = kBlock / kWave;

char *err_buf();                     // thread-local message buffer (1 KiB)
int fail(int code, const char *fmt, ...);
int host_threads();                 // CPUs this process may really use (affinity and the container's CPU quota)

#define URE_HIP(call)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return ::ure::fail((int)e_, "%s:%d %s -> %s", __FILE__, __LINE__, #call,          \
                               hipGetErrorString(e_));                                        \
    } while (0)

#define URE_ARG(cond)                                                                         \
    do {                                                                                      \
        if (!(cond)) return ::ure::fail(-1, "%s:%d argument check failed: %s", __FILE__,      \
                                        __LINE__, #cond);                                     \
    } while (0)

// A set of shards trained side by side (ure_job_t of the C ABI).
struct shard_aux;
struct ure_job {
    std::vector<ure_shard_t> host;
    ure_shard_t *dev = nullptr;
    int64_t ticks = 0;
    int64_t next_tick = 0;               // ticks run so far (steps must run in order)
    std::vector<int> row_blocks;         // per shard: workgroups its rows need in a step launch
    int max_n = 0, max_small_n = 0;
    int64_t max_slots = 0;
    int d = 0;
    bool small_shards = false, large_shards = false;   // which tag-preparation paths the job needs
    std::vector<std::vector<float>> lr_host;           // per shard: learning rate of each epoch (for the closed form)
    std::vector<double> ab_host;                       // per shard (a, b) of the lazily advanced rows
    double *dev_ab = nullptr;
    std::vector<struct shard_aux> aux_host;            // per shard: derived constants (tag_prep.h)
    struct shard_aux *dev_aux = nullptr;
    bool shard_sliced = true;            // 1-D grid, shards dealt out to XCDs in slices (URE_SHARD_FAST=2, the default)
    bool shard_fast = true;              // shard = fast index of the workgroup id (XCD affinity for 8k shards)
    bool snapshots = false;
    unsigned snap_blocks = 1;
    int64_t max_lazy = 0;                              // float4 slices of lazily advanced rows, max over shards
    bool touch = false;                                // touch mode (mf_touch.h): all shards of the job or none
    bool ahead = false;                                // touch_mode 2 (masks one epoch ahead): all shards of the job or none
    bool index = false;                                // touch_mode 3 (per-step slot index, mf_index.h): all shards of the job or none
    bool index_split = false;                          // ... and some shard has rows split over several workgroups (a combine launch per step)
    bool scatter_staged = true;                        // ... its epochs of 64+ steps sorted a chunk at a time in LDS (idx_scatter_staged_kernel; URE_INDEX_STAGED=0: record by record)
    bool all_file_tags = true;                         // every shard's batch tags come from the host: no partition / collect / scatter launches
    std::vector<void *> touch_mem;                     // library-owned device memory of touch mode (masks, tables)
    int max_units = 0;                                 // work units of the largest shard
    int64_t max_active4 = 0;                           // float4 slices of active rows, max over shards
    int max_rows = 0;
};

// tag_prep.hip: standalone per-epoch tag preparation (the step kernel carries the common case)
int tag_prep_needed(const ure_job *job, int64_t tick);                       // bit mask of the standalone passes needed (tag_prep.hip)
void launch_tag_prep(const ure_job *job, int64_t tick, hipStream_t st, int pass);

// Global-memory accessors.  A pointer that a kernel reads out of a descriptor in memory (struct
// ure_shard) has no address space the compiler can see, so a plain dereference becomes a flat_*
// instruction: it counts in vmcnt AND lgkmcnt, completes out of order and forces every wait to be
// a full `s_waitcnt vmcnt(0) lgkmcnt(0)` -- LDS reads then wait for all loads in flight.  Going
// through address space 1 gives global_* instructions (counted waits, SGPR base where uniform).
#define URE_AS1 __attribute__((address_space(1)))
typedef float ure_f4 __attribute__((ext_vector_type(4)));
typedef int ure_i4 __attribute__((ext_vector_type(4)));
typedef unsigned ure_u4 __attribute__((ext_vector_type(4)));
typedef unsigned ure_u2 __attribute__((ext_vector_type(2)));
typedef int ure_i2 __attribute__((ext_vector_type(2)));
template <typename T>
__device__ __forceinline__ T ldg(const T *p) { return *(const T URE_AS1 *)p; }
template <typename T>
__device__ __forceinline__ void stg(T *p, T v) { *(T URE_AS1 *)p = v; }
__device__ __forceinline__ float4 ldg_f4(const float *p) { const ure_f4 v = *(const ure_f4 URE_AS1 *)p; return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ int4 ldg_i4(const int32_t *p) { const ure_i4 v = *(const ure_i4 URE_AS1 *)p; return make_int4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint4 ldg_u4(const void *p) { const ure_u4 v = *(const ure_u4 URE_AS1 *)p; return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void stg_f4(float *p, float4 v) { ure_f4 t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w; *(ure_f4 URE_AS1 *)p = t; }
__device__ __forceinline__ void stg_i4(void *p, int4 v) { ure_i4 t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w; *(ure_i4 URE_AS1 *)p = t; }
__device__ __forceinline__ void stg_u4(void *p, uint4 v) { ure_u4 t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w; *(ure_u4 URE_AS1 *)p = t; }

// A lane's share of a table row: V4 float4 pieces.  Lane `sub` of the LPR lanes that share a row
// owns float4 columns sub, sub + LPR, ... so that every load / store instruction of the group
// covers one contiguous run of LPR * 16 bytes.  Row width d = LPR * V4 * 4.
template <int V4>
struct RowVec {
    float4 q[V4];
};
template <int V4>
__device__ __forceinline__ RowVec<V4> row_zero()
{
    RowVec<V4> r;
#pragma unroll
    for (int i = 0; i < V4; ++i) r.q[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    return r;
}
template <int LPR, int V4>
__device__ __forceinline__ RowVec<V4> row_load(const float *row, int sub)
{
    RowVec<V4> r;
#pragma unroll
    for (int i = 0; i < V4; ++i) r.q[i] = ldg_f4(row + (i * LPR + sub) * 4);
    return r;
}
template <int LPR, int V4>
__device__ __forceinline__ void row_store(float *row, int sub, const RowVec<V4> &v)
{
#pragma unroll
    for (int i = 0; i < V4; ++i) stg_f4(row + (i * LPR + sub) * 4, v.q[i]);
}
template <int V4>
__device__ __forceinline__ float row_dot(const RowVec<V4> &a, const RowVec<V4> &b)
{
    float p = a.q[0].x * b.q[0].x;
    p = fmaf(a.q[0].y, b.q[0].y, p);
    p = fmaf(a.q[0].z, b.q[0].z, p);
    p = fmaf(a.q[0].w, b.q[0].w, p);
#pragma unroll
    for (int i = 1; i < V4; ++i) {
        p = fmaf(a.q[i].x, b.q[i].x, p);
        p = fmaf(a.q[i].y, b.q[i].y, p);
        p = fmaf(a.q[i].z, b.q[i].z, p);
        p = fmaf(a.q[i].w, b.q[i].w, p);
    }
    return p;
}
template <int V4>
__device__ __forceinline__ void row_axpy(RowVec<V4> &acc, float s, const RowVec<V4> &v)
{
#pragma unroll
    for (int i = 0; i < V4; ++i) {
        acc.q[i].x = fmaf(s, v.q[i].x, acc.q[i].x);
        acc.q[i].y = fmaf(s, v.q[i].y, acc.q[i].y);
        acc.q[i].z = fmaf(s, v.q[i].z, acc.q[i].z);
        acc.q[i].w = fmaf(s, v.q[i].w, acc.q[i].w);
    }
}

inline bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// Cross-lane moves inside a row of 16 lanes as DPP modifiers (one VALU instruction, no trip through
// the LDS crossbar as ds_bpermute / __shfl takes, and no lgkmcnt wait).
constexpr int kDppQuadXor1 = 0xB1;       // quad_perm [1,0,3,2]
constexpr int kDppQuadXor2 = 0x4E;       // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141;    // lane i <-> 7 - i within 8 lanes
constexpr int kDppRowMirror = 0x140;     // lane i <-> 15 - i within 16 lanes
constexpr int kDppRowShr = 0x110;        // + n: lane i reads lane i - n of its 16-lane row (0 when there is none)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}

// Sum over the LPR consecutive lanes that share one table row (LPR = d/4 or d/8 lanes).  Every
// lane of the group ends with the total; the additions pair the same values as an xor butterfly.
template <int LPR>
__device__ __forceinline__ float group_sum(float v)
{
    if (LPR >= 2) v += dpp_f<kDppQuadXor1>(v);
    if (LPR >= 4) v += dpp_f<kDppQuadXor2>(v);
    if (LPR >= 8) v += dpp_f<kDppHalfMirror>(v);
    if (LPR >= 16) v += dpp_f<kDppRowMirror>(v);
    if (LPR >= 32) v += __shfl_xor(v, 16, kWave);
    if (LPR >= 64) v += __shfl_xor(v, 32, kWave);
    return v;
}

// Inclusive prefix sum over the LPR lanes of a group (lane `sub` of the group).
template <int LPR>
__device__ __forceinline__ int group_scan(int v, int sub)
{
    if (LPR > 16) {          // the group spans DPP rows: plain shuffles
#pragma unroll
        for (int o = 1; o < LPR; o <<= 1) {
            const int t = __shfl_up(v, o, LPR);
            if (sub >= o) v += t;
        }
        return v;
    }
    if (LPR >= 2) { const int t = dpp_i<kDppRowShr + 1>(v); if (sub >= 1) v += t; }
    if (LPR >= 4) { const int t = dpp_i<kDppRowShr + 2>(v); if (sub >= 2) v += t; }
    if (LPR >= 8) { const int t = dpp_i<kDppRowShr + 4>(v); if (sub >= 4) v += t; }
    if (LPR >= 16) { const int t = dpp_i<kDppRowShr + 8>(v); if (sub >= 8) v += t; }
    return v;
}

// Sum across the 64/LPR groups of a wavefront (lanes with equal lane % LPR).
template <int LPR>
__device__ __forceinline__ float cross_group_sum(float v)
{
#pragma unroll
    for (int o = LPR; o < kWave; o <<= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

}  // namespace ure
