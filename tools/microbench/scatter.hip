// Microbenchmark (measurement aid, not product code): cost of the per-epoch tag scatter.
//   hipcc -O3 --offload-arch=gfx950 -o scatter scatter.hip && ./scatter
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

template <typename T, int PER>
__global__ void scat(const int* __restrict__ perm, const int* __restrict__ pos, T* __restrict__ out, int n, int batch)
{
    for (int b0 = (blockIdx.x * 256 + threadIdx.x) * PER; b0 < n; b0 += gridDim.x * 256 * PER) {
        int j[PER], p[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) j[k] = b0 + k < n ? perm[b0 + k] : -1;
#pragma unroll
        for (int k = 0; k < PER; ++k) p[k] = j[k] >= 0 ? pos[j[k]] : -1;
#pragma unroll
        for (int k = 0; k < PER; ++k) if (p[k] >= 0) out[p[k]] = (T)((b0 + k) / batch);
    }
}
// the same scatter issued from the workgroups of ONE XCD only (blockIdx % 8 == 0), so that
// partial writes to a line can merge in that XCD's write-back L2
template <typename T, int PER>
__global__ void scat_xcd(const int* __restrict__ perm, const int* __restrict__ pos, T* __restrict__ out, int n, int batch)
{
    if (blockIdx.x % 8 != 0) return;
    const int blk = blockIdx.x / 8, nblk = (gridDim.x + 7) / 8;
    for (int b0 = (blk * 256 + threadIdx.x) * PER; b0 < n; b0 += nblk * 256 * PER) {
        int j[PER], p[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) j[k] = b0 + k < n ? perm[b0 + k] : -1;
#pragma unroll
        for (int k = 0; k < PER; ++k) p[k] = j[k] >= 0 ? pos[j[k]] : -1;
#pragma unroll
        for (int k = 0; k < PER; ++k) if (p[k] >= 0) out[p[k]] = (T)((b0 + k) / batch);
    }
}
// gather form: out[p] = tagj[src[p]] (coalesced writes, random reads)
template <typename T>
__global__ void gath(const int* __restrict__ src, const int* __restrict__ tagj, T* __restrict__ out, int n)
{
    for (int p = blockIdx.x * 256 + threadIdx.x; p < n; p += gridDim.x * 256) out[p] = (T)tagj[src[p]];
}
__global__ void inv_scatter(const int* __restrict__ perm, int* __restrict__ inv, int n)
{
    for (int b = blockIdx.x * 256 + threadIdx.x; b < n; b += gridDim.x * 256) inv[perm[b]] = b;
}

template <typename F> float timeit(F f, int reps = 20)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms * 1000.f / reps;
}

int main()
{
    for (int n : {180000, 900000, 4000000}) {
        std::vector<int> perm(n), pos(n);
        std::iota(perm.begin(), perm.end(), 0); std::iota(pos.begin(), pos.end(), 0);
        std::mt19937 g(1); std::shuffle(perm.begin(), perm.end(), g); std::shuffle(pos.begin(), pos.end(), g);
        int *dperm, *dpos, *dinv; void* dout;
        hipMalloc(&dperm, n * 4); hipMalloc(&dpos, n * 4); hipMalloc(&dinv, n * 4); hipMalloc(&dout, (size_t)n * 4);
        hipMemcpy(dperm, perm.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dpos, pos.data(), n * 4, hipMemcpyHostToDevice);
        for (int blocks : {64, 176, 704, 2048}) {
            float t2 = timeit([&] { scat<uint16_t, 4><<<blocks, 256>>>(dperm, dpos, (uint16_t*)dout, n, 30000); });
            float t4 = timeit([&] { scat<uint32_t, 4><<<blocks, 256>>>(dperm, dpos, (uint32_t*)dout, n, 30000); });
            float t1 = timeit([&] { scat<uint8_t, 4><<<blocks, 256>>>(dperm, dpos, (uint8_t*)dout, n, 30000); });
            float t21 = timeit([&] { scat<uint16_t, 1><<<blocks, 256>>>(dperm, dpos, (uint16_t*)dout, n, 30000); });
            float ti = timeit([&] { inv_scatter<<<blocks, 256>>>(dperm, dinv, n); });
            float tx = timeit([&] { scat_xcd<uint16_t, 4><<<blocks * 8, 256>>>(dperm, dpos, (uint16_t*)dout, n, 30000); });
            printf("   one-XCD scatter u16x4 (%d working blocks) %.2f us\n", blocks, tx);
            float tg = timeit([&] { gath<uint16_t><<<blocks, 256>>>(dpos, dinv, (uint16_t*)dout, n); });
            printf("n=%d blocks=%d  scatter u16x4 %.2f us | u32x4 %.2f | u8x4 %.2f | u16x1 %.2f | inv(dword, no pos) %.2f | gather->u16 %.2f\n", n, blocks, t2, t4, t1, t21, ti, tg);
        }
        hipFree(dperm); hipFree(dpos); hipFree(dinv); hipFree(dout);
    }
    return 0;
}
