// N1: the parts of one BPR training step that sit around the propagation -- edge dropout as value
// masking on the fixed CSR, the pair gather-dot + SELU loss with its gradient scatter, and the L2 term.  gfx950 only.
//
// Replaces, per mini-batch (TextGCN/base_model.py):
//   :77-86   _dropout_norm_matrix: torch.rand(nnz) on the CPU, index_select, rebuild + coalesce (a device sort), H->D copy
//   :189-198 users_emb[users], items_emb[pos|neg] gathers, score_pairwise (:171), F.selu, mean -- and their autograd backward
//   :200-210 reg_loss: three embedding gathers + norms -- and its backward
// The propagation itself (forward and transposed backward) is tgcn_spmm_* on the values these kernels write.
#include "tgcn_internal.h"

namespace tgcn {
namespace {

// ---- Philox4x32-10 (Salmon et al. 2011): counter-based, so keep(e) can be recomputed wherever entry e's fate is needed
// (the transposed value stream and the segment streams read OTHER entries' masks) without storing a mask.
__device__ __forceinline__ uint2 mulhilo(unsigned a, unsigned b)
{
    const unsigned long long p = (unsigned long long)a * b;
    return make_uint2((unsigned)(p >> 32), (unsigned)p);
}

__device__ __forceinline__ unsigned philox_first(unsigned long long seed, unsigned counter)
{
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
    unsigned c0 = counter, c1 = 0, c2 = 0, c3 = 0;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint2 a = mulhilo(0xD2511F53u, c0);
        const uint2 b = mulhilo(0xCD9E8D57u, c2);
        const unsigned n0 = b.x ^ c1 ^ k0, n1 = b.y, n2 = a.x ^ c3 ^ k1, n3 = a.y;
        c0 = n0, c1 = n1, c2 = n2, c3 = n3;
        k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
    }
    return c0;
}

struct DropArgs {
    const float *__restrict__ stored;       // [nnz] the matrix's stored values
    const float *__restrict__ stored_t;     // [nnz] stored[perm[e]], or NULL when that equals stored[e] (symmetric values)
    const float *__restrict__ rand_u;       // [nnz] uniform draws, or NULL: Philox(seed, e)
    unsigned long long seed;
    float keep_prob;
    const int *__restrict__ perm;           // [nnz] transpose permutation (A^T).vals = vals[perm], or NULL
    const float *__restrict__ ent_stored;   // [n_stream] the segment plan's own copy of the stored values
    const float *__restrict__ ent_stored_t; // [n_stream] stored_t[ent_src[s]], or NULL (symmetric values)
    const int *__restrict__ ent_src;        // [n_stream] entry each segment-stream slot copies
    const int *__restrict__ ent_src_t;      // [n_stream] perm[ent_src[s]]
    int nnz, n_stream;
    float *__restrict__ vals;               // [nnz]
    float *__restrict__ vals_t;             // [nnz] or NULL
    float *__restrict__ ev;                 // [n_stream] or NULL
    float *__restrict__ ev_t;               // [n_stream] or NULL
};

// keep entry e iff u_e < 1 - p (base_model.py:82-83)
__device__ __forceinline__ bool kept(const DropArgs &a, int e)
{
    const float u = a.rand_u ? a.rand_u[e] : (float)(philox_first(a.seed, (unsigned)e) >> 8) * (1.0f / 16777216.0f);
    return u < a.keep_prob;
}

// Every array is read and written in stream order (the only gather is rand_u[perm[e]] in the parity mode that replays a
// host-drawn mask): kept values are stored / (1 - p) (base_model.py:84; the same IEEE fp32 division the reference does).
__global__ __launch_bounds__(256) void k_dropout_values(const DropArgs a)
{
    const int stride = gridDim.x * blockDim.x;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < a.nnz; e += stride) {
        const float s = a.stored[e];
        a.vals[e] = kept(a, e) ? s / a.keep_prob : 0.0f;
        if (a.vals_t)
            a.vals_t[e] = kept(a, a.perm[e]) ? (a.stored_t ? a.stored_t[e] : s) / a.keep_prob : 0.0f;
    }
    if (a.ev) {
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n_stream; i += stride) {
            const float s = a.ent_stored[i];
            a.ev[i] = kept(a, a.ent_src[i]) ? s / a.keep_prob : 0.0f;
            if (a.ev_t)
                a.ev_t[i] = kept(a, a.ent_src_t[i]) ? (a.ent_stored_t ? a.ent_stored_t[i] : s) / a.keep_prob : 0.0f;
        }
    }
}

// ---- BPR pairs -----------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_xor(v, o);
    return v;
}

constexpr float kSeluScale = 1.0507009873554804934193349852946f;
constexpr float kSeluAlpha = 1.6732632423543772848170429916717f;
constexpr int kMaxSlabs = 8;   // d <= 512 for the training kernels

struct BprArgs {
    const float *__restrict__ users_emb;   // [U, d] propagated tables
    const float *__restrict__ items_emb;   // [I, d]
    const int64_t *__restrict__ users;     // [b]
    const int64_t *__restrict__ pos;       // [b]
    const int64_t *__restrict__ negs;      // [m, b]
    int b, m, d;
    float grad_scale;                      // multiplies every gradient written (1 / (K + 1) of the layer mean, folded in)
    const float *__restrict__ upstream;    // device scalar multiplying every gradient too (d L / d loss), or NULL (= 1)
    float *__restrict__ terms;             // [m, b] selu(s_neg - s_pos), or NULL
    float *grad_users;                     // [U, d] zero-initialised by the caller; rows are ADDED to (float atomics); or NULL
    float *grad_items;                     // [I, d]
};

// one wave per batch row; lane owns columns lane, lane + 64, ... (coalesced 256-byte slabs)
__global__ __launch_bounds__(256) void k_bpr_pairs(const BprArgs a)
{
    const int lane = lane_id();
    const int r = uniform(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (r >= a.b)
        return;
    const int n_slab = (a.d + kWave - 1) / kWave;
    if (a.users[r] < 0) {   // a padded row (ragged per-user triple lists kept dense): no term, no gradient
        if (lane == 0 && a.terms)
            for (int j = 0; j < a.m; ++j)
                a.terms[(size_t)j * a.b + r] = 0.0f;
        return;
    }
    const size_t ur = (size_t)a.users[r] * a.d, pr = (size_t)a.pos[r] * a.d;
    float u[kMaxSlabs], p[kMaxSlabs], gu[kMaxSlabs];
    float dot = 0.0f;
#pragma unroll
    for (int s = 0; s < kMaxSlabs; ++s) {
        const int c = s * kWave + lane;
        const bool on = s < n_slab && c < a.d;
        u[s] = on ? a.users_emb[ur + c] : 0.0f;
        p[s] = on ? a.items_emb[pr + c] : 0.0f;
        gu[s] = 0.0f;
        dot = fmaf(u[s], p[s], dot);
    }
    const float s_pos = wave_sum(dot);
    const float inv = a.grad_scale * (a.upstream ? *a.upstream : 1.0f) / ((float)a.b * (float)a.m);
    const bool want_grad = a.grad_users != nullptr;
    float g_pos = 0.0f;
    for (int j = 0; j < a.m; ++j) {
        const size_t nr = (size_t)a.negs[(size_t)j * a.b + r] * a.d;
        float n[kMaxSlabs];
        float dn = 0.0f;
#pragma unroll
        for (int s = 0; s < kMaxSlabs; ++s) {
            const int c = s * kWave + lane;
            n[s] = (s < n_slab && c < a.d) ? a.items_emb[nr + c] : 0.0f;
            dn = fmaf(u[s], n[s], dn);
        }
        const float x = wave_sum(dn) - s_pos;
        const float ex = expf(x);
        if (lane == 0 && a.terms)
            a.terms[(size_t)j * a.b + r] = x > 0.0f ? kSeluScale * x : kSeluScale * kSeluAlpha * (ex - 1.0f);
        if (!want_grad)
            continue;
        const float g = (x > 0.0f ? kSeluScale : kSeluScale * kSeluAlpha * ex) * inv;   // d loss / d s_neg ; d / d s_pos = -g
        g_pos -= g;
#pragma unroll
        for (int s = 0; s < kMaxSlabs; ++s) {
            const int c = s * kWave + lane;
            if (s < n_slab && c < a.d) {
                atomicAdd(a.grad_items + nr + c, g * u[s]);
                gu[s] = fmaf(g, n[s] - p[s], gu[s]);
            }
        }
    }
    if (!want_grad)
        return;
#pragma unroll
    for (int s = 0; s < kMaxSlabs; ++s) {
        const int c = s * kWave + lane;
        if (s < n_slab && c < a.d) {
            atomicAdd(a.grad_items + pr + c, g_pos * u[s]);
            atomicAdd(a.grad_users + ur + c, gu[s]);
        }
    }
}

struct RegArgs {
    const float *__restrict__ e_users;     // [U, d] layer-0 tables (the parameters)
    const float *__restrict__ e_items;     // [I, d]
    const int64_t *__restrict__ users, *__restrict__ pos, *__restrict__ negs;
    int b, m, d;
    float coef;                            // lambda / b: gradient of lambda / (2 b) * |x|^2 is coef * x
    const float *__restrict__ upstream;    // device scalar multiplying the gradient too, or NULL (= 1)
    float *__restrict__ terms;             // [b] squared norms of the 2 + m rows of batch row r, or NULL
    float *grad_users, *grad_items;        // NULL: values only
};

__global__ __launch_bounds__(256) void k_reg_rows(const RegArgs a)
{
    const int lane = lane_id();
    const int r = uniform(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (r >= a.b)
        return;
    if (a.users[r] < 0) {   // padded row
        if (lane == 0 && a.terms)
            a.terms[r] = 0.0f;
        return;
    }
    float sq = 0.0f;
    const float coef = a.coef * (a.upstream ? *a.upstream : 1.0f);
    for (int t = 0; t < 2 + a.m; ++t) {
        const bool is_user = t == 0;
        const int64_t id = is_user ? a.users[r] : t == 1 ? a.pos[r] : a.negs[(size_t)(t - 2) * a.b + r];
        const float *__restrict__ row = (is_user ? a.e_users : a.e_items) + (size_t)id * a.d;
        float *g = a.grad_users ? (is_user ? a.grad_users : a.grad_items) + (size_t)id * a.d : nullptr;
        for (int c = lane; c < a.d; c += kWave) {
            const float v = row[c];
            sq = fmaf(v, v, sq);
            if (g)
                atomicAdd(g + c, coef * v);
        }
    }
    sq = wave_sum(sq);
    if (lane == 0 && a.terms)
        a.terms[r] = sq;
}

}  // namespace
}  // namespace tgcn

using namespace tgcn;

extern "C" int tgcn_dropout_values_f32(const float *stored_vals, const float *stored_vals_t, const float *rand_u, uint64_t seed,
                                       float keep_prob, const int32_t *perm, const float *ent_stored, const float *ent_stored_t,
                                       const int32_t *ent_src, const int32_t *ent_src_t, int64_t nnz, int64_t n_stream, float *vals,
                                       float *vals_t, float *ent_val, float *ent_val_t, tgcn_stream_t stream)
{
    TGCN_REQUIRE(nnz >= 0 && nnz < INT32_MAX && n_stream >= 0 && n_stream < INT32_MAX, "nnz / n_stream out of range");
    TGCN_REQUIRE(keep_prob > 0.0f && keep_prob <= 1.0f, "keep_prob must be in (0, 1]");
    if (nnz == 0)
        return TGCN_OK;
    TGCN_REQUIRE(stored_vals && vals, "stored_vals / vals is NULL");
    TGCN_REQUIRE(!vals_t || perm, "vals_t needs the transpose permutation");
    TGCN_REQUIRE(!ent_val || (ent_src && ent_stored), "ent_val needs ent_src and ent_stored");
    TGCN_REQUIRE(!ent_val_t || (ent_val && ent_src_t), "ent_val_t needs ent_val and ent_src_t");
    DropArgs a{stored_vals, stored_vals_t, rand_u, seed, keep_prob, perm, ent_stored, ent_stored_t, ent_src, ent_src_t, (int)nnz,
               ent_val ? (int)n_stream : 0, vals, vals_t, ent_val, ent_val_t};
    const int64_t work = nnz > n_stream ? nnz : n_stream;
    const int grid = (int)((work + 255) / 256 < 8192 ? (work + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_dropout_values, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return check_launch("k_dropout_values");
}

extern "C" int tgcn_bpr_pairs_f32(const float *users_emb, const float *items_emb, const int64_t *users, const int64_t *pos,
                                  const int64_t *negs, int32_t b, int32_t m, int32_t d, float grad_scale, const float *upstream,
                                  float *terms, float *grad_users, float *grad_items, tgcn_stream_t stream)
{
    TGCN_REQUIRE(b >= 0 && m >= 1, "b / m out of range");
    TGCN_REQUIRE(d > 0 && d <= kMaxSlabs * kWave, "d out of range for the training kernels (<= 512)");
    if (b == 0)
        return TGCN_OK;
    TGCN_REQUIRE(users_emb && items_emb && users && pos && negs, "NULL pointer");
    TGCN_REQUIRE(terms || grad_users, "neither terms nor gradients requested");
    TGCN_REQUIRE((grad_users == nullptr) == (grad_items == nullptr), "give both gradient tables or neither");
    BprArgs a{users_emb, items_emb, users, pos, negs, b, m, d, grad_scale, upstream, terms, grad_users, grad_items};
    hipLaunchKernelGGL(k_bpr_pairs, dim3((b + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return check_launch("k_bpr_pairs");
}

extern "C" int tgcn_reg_rows_f32(const float *e_users, const float *e_items, const int64_t *users, const int64_t *pos,
                                 const int64_t *negs, int32_t b, int32_t m, int32_t d, float coef, const float *upstream,
                                 float *terms, float *grad_users, float *grad_items, tgcn_stream_t stream)
{
    TGCN_REQUIRE(b >= 0 && m >= 0 && d > 0, "b / m / d out of range");
    if (b == 0)
        return TGCN_OK;
    TGCN_REQUIRE(e_users && e_items && users && pos && (m == 0 || negs), "NULL pointer");
    TGCN_REQUIRE(terms || grad_users, "neither terms nor gradients requested");
    TGCN_REQUIRE((grad_users == nullptr) == (grad_items == nullptr), "give both gradient tables or neither");
    RegArgs a{e_users, e_items, users, pos, negs, b, m, d, coef, upstream, terms, grad_users, grad_items};
    hipLaunchKernelGGL(k_reg_rows, dim3((b + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return check_launch("k_reg_rows");
}
