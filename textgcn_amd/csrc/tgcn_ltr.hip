// K11-K13: the LTR text-feature head folded into ONE scoring GEMM.
//
// Replaces LTRBase.get_user_vectors / get_item_vectors / get_features_batchwise and LTRLinear's nn.Linear over
// the five features (TextGCN/ltr_models.py:95-146,181-204):
//   f = [e_u.e_i, r_u.r_i, d_u.d_i, r_u.d_i, d_u.r_i],  s = sum_j w_j f_j + b
//     = [ w0 e_u | w1 r_u + w4 d_u | w2 d_u + w3 r_u | b ] . [ e_i | r_i | d_i | 1 ]          
// so the 5 GEMMs + the 2.46 GB [B, I, 5] concat + the strided Linear become a K = d + 2t (+pad) GEMM that the
// ordinary scoring kernels run (tgcn_score_dense_f32 / tgcn_score_topk_f32).  These two kernels build the folded
// operands; both are bandwidth-trivial (B x K and I x K floats).
#include <climits>

#include "tgcn_internal.h"

namespace tgcn {
namespace {

// Ua[b, :] = [ w0 e | w1 r + w4 dsc | w2 dsc + w3 r | bias, 0... ]   (row of user user_ids[b]); K_pad columns
__global__ void k_ltr_fold_users(const float *__restrict__ e, const float *__restrict__ r, const float *__restrict__ dsc,
                                 const int64_t *__restrict__ emb_ids, const int64_t *__restrict__ text_ids, int B, int d, int t,
                                 int k_pad, float w0, float w1, float w2, float w3, float w4, float bias, float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * k_pad)
        return;
    const int b = (int)(i / k_pad), k = (int)(i % k_pad);
    const int64_t ue = emb_ids ? emb_ids[b] : b;
    const int64_t ut = text_ids ? text_ids[b] : b;
    float v = 0.0f;
    if (k < d)
        v = w0 * e[ue * d + k];
    else if (k < d + t)
        v = fmaf(w4, dsc[ut * t + (k - d)], w1 * r[ut * t + (k - d)]);
    else if (k < d + 2 * t)
        v = fmaf(w3, r[ut * t + (k - d - t)], w2 * dsc[ut * t + (k - d - t)]);
    else if (k == d + 2 * t)
        v = bias;
    out[i] = v;
}

// The same rows, a wave per row and 16 bytes per lane (d, t and the row pitch multiples of 4, 16-byte aligned tables): every text
// element is read once for both of its columns, no 64-bit division per element.  8192 users x 960 columns: 150-220 us -> see
// DESIGN.md 4.2c (the element-per-thread form above was a tenth of config 5's predict).
__global__ __launch_bounds__(256) void k_ltr_fold_users_v4(const float *__restrict__ e, const float *__restrict__ r,
                                                           const float *__restrict__ dsc, const int64_t *__restrict__ emb_ids,
                                                           const int64_t *__restrict__ text_ids, int B, int d, int t, int k_pad, float w0,
                                                           float w1, float w2, float w3, float w4, float bias, float *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B)
        return;
    const int64_t ue = emb_ids ? emb_ids[b] : b;
    const int64_t ut = text_ids ? text_ids[b] : b;
    float *__restrict__ row = out + (size_t)b * k_pad;
    const float *__restrict__ pe = e + ue * d;
    const float *__restrict__ pr = r + ut * t;
    const float *__restrict__ pd = dsc + ut * t;
    for (int c = lane * 4; c < d; c += 256) {
        const float4 x = *reinterpret_cast<const float4 *>(pe + c);
        *reinterpret_cast<float4 *>(row + c) = make_float4(w0 * x.x, w0 * x.y, w0 * x.z, w0 * x.w);
    }
    for (int c = lane * 4; c < t; c += 256) {
        const float4 x = *reinterpret_cast<const float4 *>(pr + c);
        const float4 y = *reinterpret_cast<const float4 *>(pd + c);
        *reinterpret_cast<float4 *>(row + d + c) =
            make_float4(fmaf(w4, y.x, w1 * x.x), fmaf(w4, y.y, w1 * x.y), fmaf(w4, y.z, w1 * x.z), fmaf(w4, y.w, w1 * x.w));
        *reinterpret_cast<float4 *>(row + d + t + c) =
            make_float4(fmaf(w3, x.x, w2 * y.x), fmaf(w3, x.y, w2 * y.y), fmaf(w3, x.z, w2 * y.z), fmaf(w3, x.w, w2 * y.w));
    }
    for (int c = d + 2 * t + lane; c < k_pad; c += 64)
        row[c] = c == d + 2 * t ? bias : 0.0f;
}

__global__ __launch_bounds__(256) void k_ltr_pack_items_v4(const float *__restrict__ e, const float *__restrict__ r,
                                                           const float *__restrict__ dsc, int I, int d, int t, int k_pad,
                                                           float *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int it = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (it >= I)
        return;
    float *__restrict__ row = out + (size_t)it * k_pad;
    for (int c = lane * 4; c < d; c += 256)
        *reinterpret_cast<float4 *>(row + c) = *reinterpret_cast<const float4 *>(e + (size_t)it * d + c);
    for (int c = lane * 4; c < t; c += 256) {
        *reinterpret_cast<float4 *>(row + d + c) = *reinterpret_cast<const float4 *>(r + (size_t)it * t + c);
        *reinterpret_cast<float4 *>(row + d + t + c) = *reinterpret_cast<const float4 *>(dsc + (size_t)it * t + c);
    }
    for (int c = d + 2 * t + lane; c < k_pad; c += 64)
        row[c] = c == d + 2 * t ? 1.0f : 0.0f;
}

// Ia[i, :] = [ e_i | r_i | d_i | 1, 0... ]
__global__ void k_ltr_pack_items(const float *__restrict__ e, const float *__restrict__ r, const float *__restrict__ dsc, int I,
                                 int d, int t, int k_pad, float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)I * k_pad)
        return;
    const int64_t it = i / k_pad;
    const int k = (int)(i % k_pad);
    float v = 0.0f;
    if (k < d)
        v = e[it * d + k];
    else if (k < d + t)
        v = r[it * t + (k - d)];
    else if (k < d + 2 * t)
        v = dsc[it * t + (k - d - t)];
    else if (k == d + 2 * t)
        v = 1.0f;
    out[i] = v;
}

// The five pairwise features of LTRBase.get_features_pairwise (ltr_models.py:148-166) for n gathered (user, item) rows: one wave
// per pair, lane-strided partial sums (k ascending inside a lane) and a wave reduction -- the four text rows are read from their
// tables by id, so the [n, 384] gathers of get_user_vectors / get_item_vectors (ltr_models.py:116-128) never exist.
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(256) void k_ltr_pair_features(const float *__restrict__ ue, const float *__restrict__ ie,
                                                           const float *__restrict__ ur, const float *__restrict__ ud,
                                                           const float *__restrict__ ir, const float *__restrict__ idsc,
                                                           const int64_t *__restrict__ users, const int64_t *__restrict__ items, int64_t n,
                                                           int d, int t, float *__restrict__ feats)
{
    const int lane = lane_id();
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= n)
        return;
    const int64_t u = users[p], it = items[p];
    const float *__restrict__ a = ue + p * d, *__restrict__ b = ie + p * d;
    const float *__restrict__ ru = ur + u * t, *__restrict__ du = ud + u * t, *__restrict__ ri = ir + it * t, *__restrict__ di = idsc + it * t;
    float f0 = 0.f, f1 = 0.f, f2 = 0.f, f3 = 0.f, f4 = 0.f;
    for (int k = lane; k < d; k += kWave)
        f0 = fmaf(a[k], b[k], f0);
    for (int k = lane; k < t; k += kWave) {
        const float xr = ru[k], xd = du[k], yr = ri[k], yd = di[k];
        f1 = fmaf(xr, yr, f1);
        f2 = fmaf(xd, yd, f2);
        f3 = fmaf(xr, yd, f3);
        f4 = fmaf(xd, yr, f4);
    }
    f0 = wave_sum(f0), f1 = wave_sum(f1), f2 = wave_sum(f2), f3 = wave_sum(f3), f4 = wave_sum(f4);
    if (lane == 0) {
        float *o = feats + p * 5;
        o[0] = f0, o[1] = f1, o[2] = f2, o[3] = f3, o[4] = f4;
    }
}

}  // namespace
}  // namespace tgcn

using namespace tgcn;

extern "C" int tgcn_ltr_pair_features_f32(const float *users_emb_rows, const float *items_emb_rows, const float *users_reviews,
                                          const float *users_desc, const float *items_reviews, const float *items_desc,
                                          const int64_t *users, const int64_t *items, int64_t n, int32_t d, int32_t t, float *feats,
                                          tgcn_stream_t stream)
{
    TGCN_REQUIRE(n >= 0 && n < (int64_t)4 * INT32_MAX && d > 0 && t > 0, "bad sizes");
    if (n == 0)
        return TGCN_OK;
    TGCN_REQUIRE(users_emb_rows && items_emb_rows && users_reviews && users_desc && items_reviews && items_desc && users && items && feats,
                 "NULL pointer");
    hipLaunchKernelGGL(k_ltr_pair_features, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), users_emb_rows,
                       items_emb_rows, users_reviews, users_desc, items_reviews, items_desc, users, items, n, d, t, feats);
    return check_launch("k_ltr_pair_features");
}

extern "C" int32_t tgcn_ltr_folded_width(int32_t d, int32_t t)
{
    return ((d + 2 * t + 1 + 63) / 64) * 64;  // multiple of 64: the scoring kernels' full-K fast path
}

extern "C" int tgcn_ltr_fold_users_f32(const float *users_emb, const float *users_reviews, const float *users_desc,
                                       const int64_t *emb_ids, const int64_t *text_ids, int32_t B, int32_t d, int32_t t,
                                       const float *w5_host, float bias, float *out, tgcn_stream_t stream)
{
    TGCN_REQUIRE(B >= 0 && d > 0 && t > 0, "bad sizes");
    if (B == 0)
        return TGCN_OK;
    TGCN_REQUIRE(users_emb && users_reviews && users_desc && w5_host && out, "NULL pointer");
    const int k_pad = tgcn_ltr_folded_width(d, t);
    const int64_t n = (int64_t)B * k_pad;
    if (d % 4 == 0 && t % 4 == 0 && k_pad % 4 == 0 &&
        (((size_t)users_emb | (size_t)users_reviews | (size_t)users_desc | (size_t)out) & 15) == 0) {
        hipLaunchKernelGGL(k_ltr_fold_users_v4, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), users_emb,
                           users_reviews, users_desc, emb_ids, text_ids, B, d, t, k_pad, w5_host[0], w5_host[1], w5_host[2], w5_host[3],
                           w5_host[4], bias, out);
        return check_launch("k_ltr_fold_users_v4");
    }
    hipLaunchKernelGGL(k_ltr_fold_users, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       users_emb, users_reviews, users_desc, emb_ids, text_ids, B, d, t, k_pad, w5_host[0], w5_host[1], w5_host[2],
                       w5_host[3], w5_host[4], bias, out);
    return check_launch("k_ltr_fold_users");
}

extern "C" int tgcn_ltr_pack_items_f32(const float *items_emb, const float *items_reviews, const float *items_desc, int32_t I,
                                       int32_t d, int32_t t, float *out, tgcn_stream_t stream)
{
    TGCN_REQUIRE(I >= 0 && d > 0 && t > 0, "bad sizes");
    if (I == 0)
        return TGCN_OK;
    TGCN_REQUIRE(items_emb && items_reviews && items_desc && out, "NULL pointer");
    const int k_pad = tgcn_ltr_folded_width(d, t);
    const int64_t n = (int64_t)I * k_pad;
    if (d % 4 == 0 && t % 4 == 0 && k_pad % 4 == 0 &&
        (((size_t)items_emb | (size_t)items_reviews | (size_t)items_desc | (size_t)out) & 15) == 0) {
        hipLaunchKernelGGL(k_ltr_pack_items_v4, dim3((unsigned)((I + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), items_emb,
                           items_reviews, items_desc, I, d, t, k_pad, out);
        return check_launch("k_ltr_pack_items_v4");
    }
    hipLaunchKernelGGL(k_ltr_pack_items, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       items_emb, items_reviews, items_desc, I, d, t, k_pad, out);
    return check_launch("k_ltr_pack_items");
}
