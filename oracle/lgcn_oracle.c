/*
 * lgcn_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded CPU restatement of the arithmetic on the LightGCN hot path of
 * sergey-volokhin/TextGCN (reference @ /root/reference).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; textgcn_amd/ never does.
 *
 * Parity status: PINNED.  Every function here is checked against golden vectors produced by
 * running the reference itself on CPU (tests/golden/make_golden.py -> tests/golden/g*.npz);
 * see tests/test_oracle_golden.py.
 *
 * Accumulation-order facts this file encodes:
 *   - torch.sparse.mm(COO, dense) on CPU  ==  per output row, one fp32 fmaf per stored entry, in
 *     ascending column order (the coalesced COO order), starting from +0.
 *   - torch.mean(torch.stack([E0..EK]), axis=0)  ==  (((E0+E1)+E2)+...+EK) / (K+1) in fp32.
 *   - torch.matmul on CPU has no defined summation order (BLAS); the oracle uses the k-ordered fmaf
 *     chain, which is what the gfx950 fp32 MFMA computes bit-for-bit; against the reference's own
 *     scores the comparison is normwise.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* Y = A * X, A in coalesced COO order (sorted by row then column).
 * reference: TextGCN/base_model.py:141-148 (layer_aggregation -> torch.sparse.mm) */
void orc_spmm_coo_f32(int64_t nnz, const int64_t *row, const int64_t *col, const float *val,
                      const float *X, float *Y, int64_t n_rows, int32_t d)
{
    memset(Y, 0, (size_t)n_rows * (size_t)d * sizeof(float));
    for (int64_t e = 0; e < nnz; ++e) {
        const float v = val[e];
        const float *x = X + (size_t)col[e] * (size_t)d;
        float *y = Y + (size_t)row[e] * (size_t)d;
        for (int32_t j = 0; j < d; ++j)
            y[j] = fmaf(v, x[j], y[j]);
    }
}

/* Same product from CSR arrays (int32 indices), rows [0, n_rows).  Used to check the product's CSR
 * container against the COO form and as the timed "port" kernel when torch is not wanted. */
void orc_spmm_csr_f32(const int32_t *rowptr, const int32_t *colidx, const float *val,
                      const float *X, float *Y, int64_t n_rows, int32_t d)
{
    for (int64_t r = 0; r < n_rows; ++r) {
        float *y = Y + (size_t)r * (size_t)d;
        for (int32_t j = 0; j < d; ++j)
            y[j] = 0.0f;
        for (int32_t e = rowptr[r]; e < rowptr[r + 1]; ++e) {
            const float v = val[e];
            const float *x = X + (size_t)colidx[e] * (size_t)d;
            for (int32_t j = 0; j < d; ++j)
                y[j] = fmaf(v, x[j], y[j]);
        }
    }
}

/* out = mean(stack(layers), axis=0): sequential sum in layer order, then one true division.
 * reference: TextGCN/base_model.py:150-157 (layer_combination) */
void orc_layer_mean_f32(const float *const *layers, int32_t n_layers_plus_1, int64_t n_elem, float *out)
{
    const float div = (float)n_layers_plus_1;
    for (int64_t i = 0; i < n_elem; ++i) {
        float s = layers[0][i];
        for (int32_t k = 1; k < n_layers_plus_1; ++k)
            s = s + layers[k][i];
        out[i] = s / div;
    }
}

/* S[b, i] = sum_k U[b, k] * It[i, k]   as the k-ordered fmaf chain from +0.
 * reference: TextGCN/base_model.py:173-179 (score_batchwise -> torch.matmul(users_emb, items_emb.t())) */
void orc_score_dense_f32(const float *U, const float *It, int32_t B, int32_t I, int32_t d, float *S)
{
    for (int32_t b = 0; b < B; ++b) {
        const float *u = U + (size_t)b * d;
        for (int32_t i = 0; i < I; ++i) {
            const float *t = It + (size_t)i * d;
            float s = 0.0f;
            for (int32_t k = 0; k < d; ++k)
                s = fmaf(u[k], t[k], s);
            S[(size_t)b * I + i] = s;
        }
    }
}

/* rowwise dot (k-ordered fmaf chain).  reference: base_model.py:166-171 (score_pairwise) */
void orc_score_pairwise_f32(const float *U, const float *V, int64_t n, int32_t d, float *out)
{
    for (int64_t r = 0; r < n; ++r) {
        float s = 0.0f;
        for (int32_t k = 0; k < d; ++k)
            s = fmaf(U[(size_t)r * d + k], V[(size_t)r * d + k], s);
        out[r] = s;
    }
}

/* rating[b, item] = -inf for every train item of user b.
 * reference: base_model.py:256-258 (explode + advanced-index assignment of np.NINF) */
void orc_mask_train_f32(float *S, int32_t B, int32_t I, const int32_t *mask_rowptr, const int32_t *mask_items)
{
    for (int32_t b = 0; b < B; ++b)
        for (int32_t e = mask_rowptr[b]; e < mask_rowptr[b + 1]; ++e)
            S[(size_t)b * I + mask_items[e]] = -INFINITY;
}

/* top-k per row, ordered by (value descending, index ascending); NaN never produced upstream.
 * reference: base_model.py:261 (torch.topk(rating, k=max(self.k))).  torch's order among equal
 * values is implementation-defined (SURVEY.md F11 / §7 hard part 3): parity tests compare the
 * strictly ordered prefix and set-equality inside tie groups.
 * round4 != 0 additionally applies base_model.py:263 (probs.round(decimals=4)), which ATen evaluates
 * as nearbyintf(x * 1e4f) / 1e4f in fp32. */
void orc_topk_f32(const float *S, int32_t B, int32_t I, int32_t k, int32_t round4, float *out_val, int64_t *out_idx)
{
    float *bv = (float *)malloc(sizeof(float) * (size_t)k);
    int64_t *bi = (int64_t *)malloc(sizeof(int64_t) * (size_t)k);
    for (int32_t b = 0; b < B; ++b) {
        const float *s = S + (size_t)b * I;
        int32_t n = 0;
        for (int32_t i = 0; i < I; ++i) {
            const float v = s[i];
            /* position: after all entries that beat (v, i); earlier index wins ties */
            if (n == k && !(v > bv[k - 1]))
                continue;
            int32_t p = n < k ? n : k - 1;
            while (p > 0 && v > bv[p - 1]) {
                bv[p] = bv[p - 1];
                bi[p] = bi[p - 1];
                --p;
            }
            bv[p] = v;
            bi[p] = i;
            if (n < k)
                ++n;
        }
        for (int32_t j = 0; j < k; ++j) {
            float v = bv[j];
            if (round4)
                v = nearbyintf(v * 10000.0f) / 10000.0f;
            out_val[(size_t)b * k + j] = v;
            out_idx[(size_t)b * k + j] = bi[j];
        }
    }
    free(bv);
    free(bi);
}

/* LTR batchwise score, the reference's own evaluation order:
 *   f0 = e_u.e_i  f1 = r_u.r_i  f2 = d_u.d_i  f3 = r_u.d_i  f4 = d_u.r_i     (ltr_models.py:131-146)
 *   s  = Linear(5,1)(f) = ((((b + f0 w0) + f1 w1) + f2 w2) + f3 w3) + f4 w4     (ltr_models.py:181-204)
 * each dot a k-ordered fmaf chain.  BLAS/addmv order inside torch is unspecified, so parity against
 * the golden LTR scores is normwise. */
void orc_ltr_score_f32(const float *eu, const float *ru, const float *du, /* [B,d], [B,t], [B,t] */
                       const float *ei, const float *ri, const float *di, /* [I,d], [I,t], [I,t] */
                       const float *w, float bias, int32_t B, int32_t I, int32_t d, int32_t t, float *S)
{
    for (int32_t b = 0; b < B; ++b) {
        for (int32_t i = 0; i < I; ++i) {
            float f[5] = {0, 0, 0, 0, 0};
            for (int32_t k = 0; k < d; ++k)
                f[0] = fmaf(eu[(size_t)b * d + k], ei[(size_t)i * d + k], f[0]);
            for (int32_t k = 0; k < t; ++k) {
                const float r_u = ru[(size_t)b * t + k], d_u = du[(size_t)b * t + k];
                const float r_i = ri[(size_t)i * t + k], d_i = di[(size_t)i * t + k];
                f[1] = fmaf(r_u, r_i, f[1]);
                f[2] = fmaf(d_u, d_i, f[2]);
                f[3] = fmaf(r_u, d_i, f[3]);
                f[4] = fmaf(d_u, r_i, f[4]);
            }
            float s = bias;
            for (int j = 0; j < 5; ++j)
                s = fmaf(f[j], w[j], s);
            S[(size_t)b * I + i] = s;
        }
    }
}

/* ltr_models.py:148-166,206-210 (pairwise features + Linear) for n (user,item) pairs already gathered */
void orc_ltr_pairwise_f32(const float *eu, const float *ru, const float *du, const float *ei, const float *ri,
                          const float *di, const float *w, float bias, int64_t n, int32_t d, int32_t t, float *out)
{
    for (int64_t r = 0; r < n; ++r) {
        float f[5] = {0, 0, 0, 0, 0};
        for (int32_t k = 0; k < d; ++k)
            f[0] = fmaf(eu[(size_t)r * d + k], ei[(size_t)r * d + k], f[0]);
        for (int32_t k = 0; k < t; ++k) {
            const float r_u = ru[(size_t)r * t + k], d_u = du[(size_t)r * t + k];
            const float r_i = ri[(size_t)r * t + k], d_i = di[(size_t)r * t + k];
            f[1] = fmaf(r_u, r_i, f[1]);
            f[2] = fmaf(d_u, d_i, f[2]);
            f[3] = fmaf(r_u, d_i, f[3]);
            f[4] = fmaf(d_u, r_i, f[4]);
        }
        float s = bias;
        for (int j = 0; j < 5; ++j)
            s = fmaf(f[j], w[j], s);
        out[r] = s;
    }
}
