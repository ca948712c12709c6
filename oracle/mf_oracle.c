/*
 * oracle/mf_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Scalar CPU restatement of the reference's MF training / scoring arithmetic,
 * used as the parity checker for the HIP path (tests/, __graft_entry__.smoke(),
 * bench.py's cpu_baseline leg).  Nothing under ultrare_amd/ may import, link or
 * call this file.  Parity status: PINNED by tests/golden/full_mf_toy.npz and
 * tests/golden/sisa_toy.npz, which were produced by the real reference
 * (tests/golden/make_golden.py, run in the build container).
 *
 * Follows (file:line under /root/reference):
 *   method/utils.py:42-43    MF.forward         pred_b = sum_j U[u_b,j] * V[i_b,j]
 *   method/utils.py:58-91    baseTrain loop     per batch: loss = sum (pred-r)^2,
 *                                               zero_grad / backward / step
 *   method/scratch.py:64-69  optim.SGD(lr, weight_decay=lam, momentum), dense grads
 *   method/utils.py:140-148  baseTest           pred = mean_m score_m ; sum (pred-r)^2
 *   method/sisa.py:52-58     user-row merge
 *   method/utils.py:637      OT cost            ((X - C[:,None])**2).sum(2), fp32,
 *                                               numpy pairwise (8-accumulator) order
 *   method/utils.py:648      centroid           X[label==i].mean(0), fp32 sequential
 *
 * Summation orders are documented per function; the HIP path is allowed to
 * differ from them within the tolerance BASELINE.json states (1e-4 relative on
 * embeddings / metrics), except where a comment says "bit-exact".
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* One optimizer step over batch perm[b0:b1) (read.py:108-133 defines the batch as
 * a contiguous slice of the epoch's permutation).  gU/gV are caller-provided
 * scratch of the table sizes.  Returns the fp32 batch loss (MSELoss(sum)).
 *   forward   utils.py:63    backward utils.py:89-91 (autograd: d loss/d pred = 2 e)
 *   step      torch.optim.SGD: g += lam*w ; m = first ? g : mu*m + g ; w -= lr*m
 */
float ure_oracle_train_step(float *U, float *V, float *mU, float *mV,
                            float *gU, float *gV,
                            const int32_t *uid, const int32_t *iid, const float *rating,
                            const int32_t *perm, int64_t b0, int64_t b1,
                            int32_t n_user, int32_t n_item, int32_t d,
                            float lr, float lam, float mu, int32_t first_step)
{
    const size_t nu = (size_t)n_user * d, ni = (size_t)n_item * d;
    memset(gU, 0, nu * sizeof(float));
    memset(gV, 0, ni * sizeof(float));
    /* torch's fp32 sum is a cascade (vectorised multi-accumulator) sum, accurate to
     * ~1 ulp; a double accumulator rounded once is the closest simple restatement. */
    double loss = 0.0;
    for (int64_t b = b0; b < b1; ++b) {
        const int64_t j = perm ? perm[b] : b;
        const float *u = U + (size_t)uid[j] * d;
        const float *v = V + (size_t)iid[j] * d;
        float pred = 0.0f;
        for (int32_t c = 0; c < d; ++c) pred += u[c] * v[c];
        const float e = pred - rating[j];
        loss += (double)(e * e);
        const float ge = 2.0f * e;
        float *gu = gU + (size_t)uid[j] * d;
        float *gv = gV + (size_t)iid[j] * d;
        for (int32_t c = 0; c < d; ++c) {
            gu[c] += ge * v[c];
            gv[c] += ge * u[c];
        }
    }
    for (int t = 0; t < 2; ++t) {
        float *w = t ? V : U, *m = t ? mV : mU, *g = t ? gV : gU;
        const size_t n = t ? ni : nu;
        for (size_t k = 0; k < n; ++k) {
            const float gg = g[k] + lam * w[k];
            const float mm = first_step ? gg : mu * m[k] + gg;
            m[k] = mm;
            w[k] = w[k] - lr * mm;
        }
    }
    return (float)loss;
}

/* One epoch = ceil(N/B) steps over consecutive B-slices of perm, last partial
 * batch kept (read.py:133, DataLoader default drop_last=False).  Returns
 * sum over batches of the fp32 batch loss, accumulated in double exactly like
 * `train_loss += loss.item()` (utils.py:82).  *step_count is advanced so that the
 * momentum buffer is initialised on the very first step of the run only. */
double ure_oracle_train_epoch(float *U, float *V, float *mU, float *mV,
                              const int32_t *uid, const int32_t *iid, const float *rating,
                              const int32_t *perm, int64_t N, int64_t B,
                              int32_t n_user, int32_t n_item, int32_t d,
                              float lr, float lam, float mu, int64_t *step_count)
{
    float *gU = (float *)malloc((size_t)n_user * d * sizeof(float));
    float *gV = (float *)malloc((size_t)n_item * d * sizeof(float));
    double total = 0.0;
    for (int64_t b0 = 0; b0 < N; b0 += B) {
        const int64_t b1 = b0 + B < N ? b0 + B : N;
        total += (double)ure_oracle_train_step(U, V, mU, mV, gU, gV, uid, iid, rating, perm, b0, b1,
                                               n_user, n_item, d, lr, lam, mu, *step_count == 0);
        *step_count += 1;
    }
    free(gU);
    free(gV);
    return total;
}

/* Ensemble score (utils.py:140-145): pred = (sum_m U_m[u].V_m[i]) / S in fp32,
 * models in list order.  Us/Vs are arrays of S table pointers. */
void ure_oracle_score(const float *const *Us, const float *const *Vs, int32_t S,
                      const int32_t *uid, const int32_t *iid, int64_t n, int32_t d, float *pred)
{
    for (int64_t j = 0; j < n; ++j) {
        float acc = 0.0f;
        for (int32_t m = 0; m < S; ++m) {
            const float *u = Us[m] + (size_t)uid[j] * d;
            const float *v = Vs[m] + (size_t)iid[j] * d;
            float p = 0.0f;
            for (int32_t c = 0; c < d; ++c) p += u[c] * v[c];
            acc += p;
        }
        pred[j] = acc / (float)S;
    }
}

/* merged[rows[t]] = src[rows[t]]   (sisa.py:55-56, 110-111) */
void ure_oracle_merge_rows(float *dst, const float *src, const int64_t *rows, int64_t n_rows, int32_t d)
{
    for (int64_t t = 0; t < n_rows; ++t)
        memcpy(dst + (size_t)rows[t] * d, src + (size_t)rows[t] * d, (size_t)d * sizeof(float));
}

/* numpy's float32 pairwise sum of n contiguous values, n < 128 (PW_BLOCKSIZE):
 * eight running accumulators over blocks of 8, combined as
 * ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), leftovers added sequentially; for n < 8 a
 * plain left-to-right loop.  SURVEY section 7 ("Exact OT on GPU") checked this bit
 * for bit for d in {16,20,32,64,128}. */
static float np_pairwise_f32(const float *a, int32_t n)
{
    if (n > 128) {          /* numpy: runs longer than PW_BLOCKSIZE are split in two, the first half a multiple of 8 */
        int32_t n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise_f32(a, n2) + np_pairwise_f32(a + n2, n - n2);
    }
    if (n < 8) {
        float res = 0.0f;
        for (int32_t i = 0; i < n; ++i) res += a[i];
        return res;
    }
    float r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int32_t i;
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}

/* dist[c][i] = sum_j (X[i][j] - C[c][j])^2   (utils.py:637), d <= 4096.  bit-exact. */
void ure_oracle_ot_cost(const float *X, const float *C, int64_t n, int32_t k, int32_t d, float *dist /* [k][n] */)
{
    float tmp[4096];
    if (d > 4096) return;
    for (int32_t c = 0; c < k; ++c)
        for (int64_t i = 0; i < n; ++i) {
            for (int32_t j = 0; j < d; ++j) {
                const float t = X[(size_t)i * d + j] - C[(size_t)c * d + j];
                tmp[j] = t * t;
            }
            dist[(size_t)c * n + i] = np_pairwise_f32(tmp, d);
        }
}

/* new_centroid[c] = X[label==c].mean(axis=0)  (utils.py:648): numpy reduces over
 * axis 0 of the gathered [cnt,d] block row by row (sequential fp32 adds in
 * ascending row id), then divides by the fp32 count.  bit-exact. */
void ure_oracle_centroids(const float *X, const int64_t *label, int64_t n, int32_t k, int32_t d, float *C /* [k][d] */)
{
    int64_t *cnt = (int64_t *)calloc((size_t)k, sizeof(int64_t));
    memset(C, 0, (size_t)k * d * sizeof(float));
    for (int64_t i = 0; i < n; ++i) {
        float *c = C + (size_t)label[i] * d;
        const float *x = X + (size_t)i * d;
        if (cnt[label[i]]++ == 0)
            memcpy(c, x, (size_t)d * sizeof(float));
        else
            for (int32_t j = 0; j < d; ++j) c[j] += x[j];
    }
    for (int32_t c = 0; c < k; ++c)
        for (int32_t j = 0; j < d; ++j) C[(size_t)c * d + j] /= (float)cnt[c];
    free(cnt);
}
