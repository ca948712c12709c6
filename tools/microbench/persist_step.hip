// persist_step.hip -- what would ONE persistent launch per job cost per optimizer step?  
//
// The ml-1m job of bench.py, reduced to its communication pattern: 5 shards x 51 workgroups (one per CU), every workgroup owns
// 192 table rows of 32 floats (weights + momentum stay in registers for the whole launch), and in every step each row gathers
// P rows that OTHER workgroups of its shard wrote in the previous step, updates itself and publishes its new weights.
//   mode 0: one launch per step (plain loads / stores, the kernel boundary is the hand-off) -- the structure the product has
//   mode 1: one launch for all steps; rows published write-through (sc1), a per-shard arrival counter as the barrier, gathers
//           by sc1 loads (MI355X_MICROARCH.md "Valid forms": every store and every load of the handed-off bytes sc1, every
//           storing wave drained before its workgroup's one counter add)
// Both modes must end with identical tables (the check that the hand-off is sound).  Prints us per step for both.
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -o persist_step persist_step.hip && ./persist_step [steps] [partners]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

constexpr int kShards = 5, kWgPerShard = 51, kThreads = 512, kLpr = 8, kD = 32;
constexpr int kRowsPerWg = 3 * (kThreads / kLpr);          // 192: three rows per lane group
constexpr int kRows = kWgPerShard * kRowsPerWg;            // 9,792 rows per shard (ml-1m: 9,746)
constexpr int kMaxP = 8;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float float4_ __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float group_sum8(float v)
{
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    return v;
}

struct Args {
    float *W[2];                 // [kShards][kRows][kD], double buffered
    const int *partner;          // [steps][kShards][kRows][P]
    unsigned *counter;           // [kShards] arrival counters (mode 1), on lines of their own (x 32 words)
    unsigned *gave_up;
    int steps, P;
};

// base: wave-uniform start of a shard's table (the buffer descriptor lives in scalar registers), at: the lane's float offset
template <bool SC1>
__device__ __forceinline__ float4_ load_row(const float *base, unsigned at)
{
    if constexpr (SC1) {
        // 16 bytes with sc1 (aux 16): bypasses this CU's L1, served beyond it
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, kRows * kD * 4, 0x00020000);
        return __builtin_bit_cast(float4_, __builtin_amdgcn_raw_buffer_load_b128(rs, at * 4, 0, 16));
    } else {
        return *(const float4_ *)(base + at);
    }
}
template <bool SC1>
__device__ __forceinline__ void store_row(float *base, unsigned at, float4_ v)
{
    if constexpr (SC1) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, kRows * kD * 4, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rs, at * 4, 0, 16);
    } else {
        *(float4_ *)(base + at) = v;
    }
}

// one step of one workgroup: w / m are the lane group's three rows (registers); reads buffer `cur`, writes buffer `cur ^ 1`.
// All partner indices of the step are requested first, then all partner rows, then the arithmetic (the memory-level
// parallelism the product's step kernel has).  idx: the step's indices if the caller fetched them ahead (persistent mode
// does, before its barrier: they do not depend on the previous step), else fetched here.
template <bool SC1, int P>
__device__ __forceinline__ void fetch_idx(const Args &A, int shard, int wg, int t, int (&idx)[3][P])
{
    const int g = threadIdx.x / kLpr;
    const int *pt = A.partner + ((size_t)t * kShards + shard) * kRows * P;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int row = wg * kRowsPerWg + q * (kThreads / kLpr) + g;
#pragma unroll
        for (int p = 0; p < P; ++p) idx[q][p] = pt[(size_t)row * P + p];           // written by the host: plain loads
    }
}
template <bool SC1, int P>
__device__ __forceinline__ void step_body(const Args &A, int shard, int wg, int cur, const int (&idx)[3][P], float4_ (&w)[3], float4_ (&m)[3])
{
    const int g = threadIdx.x / kLpr, sub = threadIdx.x % kLpr;
    const float *Wc = A.W[cur] + (size_t)shard * kRows * kD;
    float *Wn = A.W[cur ^ 1] + (size_t)shard * kRows * kD;
    float4_ o[3][P];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int p = 0; p < P; ++p) o[q][p] = load_row<SC1>(Wc, (unsigned)idx[q][p] * kD + sub * 4);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int row = wg * kRowsPerWg + q * (kThreads / kLpr) + g;
        float4_ acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const float dot = group_sum8(o[q][p].x * w[q].x + o[q][p].y * w[q].y + o[q][p].z * w[q].z + o[q][p].w * w[q].w) - 0.5f;
            acc += o[q][p] * dot;
        }
        m[q] = m[q] * 0.9f + acc + w[q] * 0.1f;
        w[q] = w[q] - m[q] * 1e-3f;
        store_row<SC1>(Wn, (unsigned)row * kD + sub * 4, w[q]);
    }
}

template <int P>
__global__ __launch_bounds__(kThreads) void per_step_kernel(Args A, int t)
{
    const int shard = blockIdx.x / kWgPerShard, wg = blockIdx.x % kWgPerShard;
    // weights and momentum come from / go to memory every launch, as in the product's step kernel
    extern __shared__ float unused[];
    const int g = threadIdx.x / kLpr, sub = threadIdx.x % kLpr;
    const int cur = t & 1;
    float4_ w[3], m[3];
    float *M = A.W[0] + (size_t)2 * kShards * kRows * kD;        // momentum stored behind the two weight buffers (see main)
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const size_t at = ((size_t)shard * kRows + wg * kRowsPerWg + q * (kThreads / kLpr) + g) * kD + sub * 4;
        w[q] = *(const float4_ *)(A.W[cur] + at);
        m[q] = *(const float4_ *)(M + at);
    }
    int idx[3][P];
    fetch_idx<false, P>(A, shard, wg, t, idx);
    step_body<false, P>(A, shard, wg, cur, idx, w, m);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const size_t at = ((size_t)shard * kRows + wg * kRowsPerWg + q * (kThreads / kLpr) + g) * kD + sub * 4;
        *(float4_ *)(M + at) = m[q];
    }
}

template <int P>
__global__ __launch_bounds__(kThreads) void persistent_kernel(Args A)
{
    const int shard = blockIdx.x / kWgPerShard, wg = blockIdx.x % kWgPerShard;
    const int g = threadIdx.x / kLpr, sub = threadIdx.x % kLpr;
    float4_ w[3], m[3];
    float *M = A.W[0] + (size_t)2 * kShards * kRows * kD;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const size_t at = ((size_t)shard * kRows + wg * kRowsPerWg + q * (kThreads / kLpr) + g) * kD + sub * 4;
        w[q] = *(const float4_ *)(A.W[0] + at);                  // written by the host before the launch
        m[q] = *(const float4_ *)(M + at);
    }
    unsigned *cnt = A.counter + shard * 32;
    __shared__ int ok;
    int idx[3][P];
    fetch_idx<true, P>(A, shard, wg, 0, idx);
    for (int t = 0; t < A.steps; ++t) {
        step_body<true, P>(A, shard, wg, t & 1, idx, w, m);
        if (t + 1 < A.steps) fetch_idx<true, P>(A, shard, wg, t + 1, idx);          // in flight across the barrier
        // publish: every storing wave drains its write-through stores, the workgroup meets, ONE lane adds to the shard's counter
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (also waits for the prefetched indices: a counted wait would not)
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)kWgPerShard * (unsigned)(t + 1);
            int good = 1;
            unsigned spins = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 4000000u) { good = 0; atomicExch(A.gave_up, 1u); break; }      // bounded: every wave reaches the exit
            }
            ok = good;
        }
        __syncthreads();
        if (!ok) return;
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const size_t at = ((size_t)shard * kRows + wg * kRowsPerWg + q * (kThreads / kLpr) + g) * kD + sub * 4;
        *(float4_ *)(M + at) = m[q];
    }
}

int main(int argc, char **argv)
{
    const int steps = argc > 1 ? atoi(argv[1]) : 200, P = argc > 2 ? atoi(argv[2]) : 6;
    if (steps < 2 || steps % 2 || (P != 3 && P != 6 && P != 8)) { fprintf(stderr, "steps even >= 2, partners 3, 6 or 8\n"); return 2; }
    const size_t table = (size_t)kShards * kRows * kD;
    std::vector<float> init(3 * table, 0.f);
    srand(3);
    for (size_t i = 0; i < table; ++i) init[i] = (float)(rand() % 2001 - 1000) * 1e-3f;
    std::vector<int> partner((size_t)steps * kShards * kRows * P);
    for (auto &x : partner) x = rand() % kRows;
    float *buf[2];
    int *d_partner;
    unsigned *d_counter, *d_gave;
    CHECK(hipMalloc(&d_partner, partner.size() * sizeof(int)));
    CHECK(hipMemcpy(d_partner, partner.data(), partner.size() * sizeof(int), hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_counter, kShards * 32 * sizeof(unsigned) + 64));
    d_gave = d_counter + kShards * 32;
    std::vector<float> result[2];
    double us[2];
    for (int mode = 0; mode < 2; ++mode) {
        CHECK(hipMalloc(&buf[mode], 3 * table * sizeof(float)));
        Args A;
        A.W[0] = buf[mode]; A.W[1] = buf[mode] + table; A.partner = d_partner; A.counter = d_counter; A.gave_up = d_gave; A.steps = steps; A.P = P;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipMemcpy(buf[mode], init.data(), 3 * table * sizeof(float), hipMemcpyHostToDevice));
            CHECK(hipMemset(d_counter, 0, kShards * 32 * sizeof(unsigned) + 64));
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
#define RUN(PP) if (mode == 0) { for (int t = 0; t < steps; ++t) hipLaunchKernelGGL(per_step_kernel<PP>, dim3(kShards * kWgPerShard), dim3(kThreads), 0, 0, A, t); } \
                else hipLaunchKernelGGL(persistent_kernel<PP>, dim3(kShards * kWgPerShard), dim3(kThreads), 0, 0, A)
            if (P == 3) { RUN(3); } else if (P == 6) { RUN(6); } else { RUN(8); }
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        us[mode] = best * 1e3 / steps;
        result[mode].resize(table);
        CHECK(hipMemcpy(result[mode].data(), buf[mode], table * sizeof(float), hipMemcpyDeviceToHost));      // steps even: final weights in W[0]
    }
    unsigned gave = 0;
    CHECK(hipMemcpy(&gave, d_gave, sizeof(gave), hipMemcpyDeviceToHost));
    size_t diff = 0;
    for (size_t i = 0; i < table; ++i) diff += memcmp(&result[0][i], &result[1][i], 4) != 0;
    printf("{\"steps\": %d, \"partners\": %d, \"rows_per_shard\": %d, \"us_per_step_launches\": %.2f, \"us_per_step_persistent\": %.2f, "
           "\"words_that_differ\": %zu, \"gave_up\": %u}\n", steps, P, kRows, us[0], us[1], diff, gave);
    return diff || gave ? 1 : 0;
}
