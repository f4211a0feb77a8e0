// K5..K9: scoring.  Dense user x item scores on the fp32 matrix cores, train-item mask, per-row top-k,
// pairwise dots.  gfx950 only.
//
// Replaces torch.matmul (TextGCN/base_model.py:179), the pandas explode + -inf scatter
// (base_model.py:257-258), torch.topk (base_model.py:261), .round(decimals=4) (base_model.py:263) and
// torch.sum(u*i, dim=1) (base_model.py:171).
//
// Numerics contract: every score is the k-ordered fp32 fmaf chain from +0 over the embedding dimension
// (v_mfma_f32_32x32x2_f32 accumulates exactly that chain), which the parity tests check bit for bit.
// top-k order is (value descending, index ascending).
#include <climits>

#include "tgcn_internal.h"
#include "tgcn_topk.h"

namespace tgcn {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// ------------------------------------------------------------------------------------------------
// dense scores: S[B, I] = U[user_ids] . It^T
//
// Workgroup = 4 waves, output tile 128 users x 128 items, K walked in chunks of 64 through LDS.
// Wave w owns users [32w, 32w+32) x all 128 items of the tile: 4 accumulators of 32x32.
// MFMA operand layout (32x32x2 f32): lane l supplies A[row = l&31][k = l>>5] and B[k = l>>5][col = l&31],
// i.e. the two half-waves h = l>>5 hold k = 2t and k = 2t+1 of step t.  To feed that without a per-lane
// select, each group of four k is stored in LDS as (k0, k2, k1, k3): half h reads the float2 at 4q + 2h
// and uses .x for step 2q (k = 4q + h) and .y for step 2q+1 (k = 4q + 2 + h) -- ascending k, so every
// dot product is the k-ordered fmaf chain.  Row stride 66 floats: the ds_read_b64 of a half-wave
// (rows 0..31, fixed q) covers banks (2*row + 4q + 2h) mod 64 -- all 64 banks once, conflict-free.
constexpr int kTile = 128;
constexpr int kKC = 64;
constexpr int kLdsRow = kKC + 2;

struct DenseArgs {
    const float *__restrict__ U;
    const int64_t *__restrict__ user_ids;
    const float *__restrict__ It;
    float *__restrict__ S;
    int64_t lds;
    int B, I, d;
    int item_mul;  // item row i lives at It[i * item_mul, :] (strided item sample; 1 = every item)
};

// stage rows [row0, row0+128) x k [k0, k0+64) of a row-major [n_rows, d] table (optionally gathered
// through ids) into LDS, zero-filled outside the table
template <bool FULLK>
__device__ __forceinline__ void stage_tile(float *__restrict__ dst, const float *__restrict__ src,
                                           const int64_t *__restrict__ ids, int row0, int n_rows, int k0, int d,
                                           int row_mul = 1)
{
    const int t = threadIdx.x;
    constexpr int N = (kTile * kKC / 4) / 256;
    float4 v[N];
    // source rows first (all id loads of a gathered tile in one batch), then all the row loads, then the LDS writes: written
    // per element, the id fetch and its row fetch become N dependent round trips
    int64_t srows[N];
    if (ids) {
#pragma unroll
        for (int i = 0; i < N; ++i)
            srows[i] = ids[min(row0 + ((i * 256 + t) >> 4), n_rows - 1)];
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i)
            srows[i] = (int64_t)min(row0 + ((i * 256 + t) >> 4), n_rows - 1) * row_mul;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int f = i * 256 + t;
        const int r = f >> 4;  // 16 float4 per row chunk
        const int q = f & 15;
        const int row = row0 + r;
        const int k = k0 + q * 4;
        // FULLK (d % 64 == 0): rows past the table are clamped, not zero-filled (their products are never stored),
        // so the loads are unconditional and overlap; other widths need zeros in the K padding (predicated path)
        const int64_t srow = srows[i];
        const float *p = src + (size_t)srow * d;
        if constexpr (FULLK) {
            v[i] = *reinterpret_cast<const float4 *>(p + k);
        } else {
            const bool ok = row < n_rows;
            v[i].x = (ok && k + 0 < d) ? p[k + 0] : 0.0f;
            v[i].y = (ok && k + 1 < d) ? p[k + 1] : 0.0f;
            v[i].z = (ok && k + 2 < d) ? p[k + 2] : 0.0f;
            v[i].w = (ok && k + 3 < d) ? p[k + 3] : 0.0f;
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int f = i * 256 + t;
        float *o = dst + (f >> 4) * kLdsRow + (f & 15) * 4;
        *reinterpret_cast<float2 *>(o) = make_float2(v[i].x, v[i].z);
        *reinterpret_cast<float2 *>(o + 2) = make_float2(v[i].y, v[i].w);
    }
}

template <bool FULLK>
__global__ __launch_bounds__(256) void k_score_dense(const DenseArgs a)
{
    __shared__ __attribute__((aligned(16))) float smem[2 * kTile * kLdsRow];
    float *ldsU = smem;
    float *ldsI = smem + kTile * kLdsRow;
    const int lane = lane_id();
    const int w = threadIdx.x >> 6;
    const int r32 = lane & 31;
    const int h = lane >> 5;
    const int i0 = blockIdx.x * kTile;
    const int u0 = blockIdx.y * kTile;

    f32x16 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            acc[n][r] = 0.0f;

    for (int k0 = 0; k0 < a.d; k0 += kKC) {
        if (k0)
            __syncthreads();
        stage_tile<FULLK>(ldsU, a.U, a.user_ids, u0, a.B, k0, a.d);
        stage_tile<FULLK>(ldsI, a.It, nullptr, i0, a.I, k0, a.d, a.item_mul);
        __syncthreads();
        const float *pu = ldsU + (w * 32 + r32) * kLdsRow + 2 * h;
        const float *pi = ldsI + r32 * kLdsRow + 2 * h;
#pragma unroll 4
        for (int q = 0; q < kKC / 4; ++q) {
            const float2 a2 = *reinterpret_cast<const float2 *>(pu + q * 4);
            float2 b2[4];
#pragma unroll
            for (int n = 0; n < 4; ++n)
                b2[n] = *reinterpret_cast<const float2 *>(pi + n * 32 * kLdsRow + q * 4);
#pragma unroll
            for (int n = 0; n < 4; ++n)
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.x, b2[n].x, acc[n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < 4; ++n)
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.y, b2[n].y, acc[n], 0, 0, 0);
        }
    }

    // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int item = i0 + n * 32 + r32;
        const bool item_ok = item < a.I;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int user = u0 + w * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (item_ok && user < a.B)
                a.S[(size_t)user * a.lds + item] = acc[n][r];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dense scores with a threshold epilogue instead of the store (wide operands: the folded ltr_linear GEMM, K = 960).
// Same tile loop as k_score_dense -- 128 users x 128 items per workgroup, K in chunks of 64 through LDS -- but a workgroup
// walks ALL the tiles of one item split, and a finished tile's scores are compared with the users' thresholds in registers:
// a passing (score, item) goes to the user's log of this split, at a position drawn from an LDS counter (0.6 % of the scores
// pass, so the atomics are rare; log order is immaterial to the selection).  The [B, I] matrix (491 MB per 2048-user batch
// at config 5) is neither written nor read back.  Logs / counts have the layout k_select reads: [B][S][2][cap2], a split's
// log filling half 0 then half 1.
struct DenseFilterArgs {
    const float *__restrict__ U;
    const int64_t *__restrict__ user_ids;
    const float *__restrict__ It;
    const float *__restrict__ tau;
    int tau_stride;
    float2 *__restrict__ logs;
    int *__restrict__ counts;
    int B, I, d, S, items_per_split, cap2;
};

template <bool FULLK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_score_dense_filter(const DenseFilterArgs a)
{
    __shared__ __attribute__((aligned(16))) float smem[2 * kTile * kLdsRow];
    __shared__ int scnt[kTile];
    float *ldsU = smem;
    float *ldsI = smem + kTile * kLdsRow;
    const int lane = lane_id();
    const int w = threadIdx.x >> 6;
    const int r32 = lane & 31;
    const int h = lane >> 5;
    const int split = blockIdx.x;
    const int u0 = blockIdx.y * kTile;
    const int i_beg = split * a.items_per_split;
    const int i_end = min(a.I, i_beg + a.items_per_split);
    if (threadIdx.x < kTile)
        scnt[threadIdx.x] = 0;
    // C/D layout: col = lane & 31 (item), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (user): 16 users per lane
    float tau_r[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int user = u0 + w * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        tau_r[r] = user < a.B ? a.tau[(size_t)user * a.tau_stride] : INFINITY;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
        asm volatile("" ::"v"(tau_r[r]));   // arrived before the loops (see filter_pipelined in tgcn_score_fused.hip)
    const int cap = 2 * a.cap2;
    // threshold epilogue of one finished tile
    auto epilogue = [&](const f32x16 (&acc)[4], int i0) {
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int item = i0 + n * 32 + r32;
            if (item < i_end) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (acc[n][r] > tau_r[r]) {   // tau = +inf for rows past B
                        const int ul = w * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        const int pos = atomicAdd(&scnt[ul], 1);
                        if (pos < cap)
                            a.logs[((size_t)(u0 + ul) * a.S + split) * cap + pos] = make_float2(acc[n][r], __int_as_float(item));
                    }
                }
            }
        }
    };
    auto mfma_chunk = [&](f32x16 (&acc)[4]) {
        const float *pu = ldsU + (w * 32 + r32) * kLdsRow + 2 * h;
        const float *pi = ldsI + r32 * kLdsRow + 2 * h;
#pragma unroll 4
        for (int q = 0; q < kKC / 4; ++q) {
            const float2 a2 = *reinterpret_cast<const float2 *>(pu + q * 4);
            float2 b2[4];
#pragma unroll
            for (int n = 0; n < 4; ++n)
                b2[n] = *reinterpret_cast<const float2 *>(pi + n * 32 * kLdsRow + q * 4);
#pragma unroll
            for (int n = 0; n < 4; ++n)
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.x, b2[n].x, acc[n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < 4; ++n)
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.y, b2[n].y, acc[n], 0, 0, 0);
        }
    };
    if constexpr (FULLK) {
        // (tile, K chunk) steps flattened; the operands of step s + 1 are requested into registers before the MFMAs of step s and
        // stored to LDS after them: a step's global round trip hides under 128 MFMAs per wave instead of standing between two
        // barriers.  Every load unconditional (the step past the end repeats the last one, rows past the table clamp): with a
        // branch around the prefetch -- round 1's attempt, 320 vs 226 ms -- hipcc waits for it with vmcnt(0) where it is issued.
        constexpr int N = (kTile * kKC / 4) / 256;
        const int n_chunks = a.d / kKC;
        const int n_steps = ((i_end - i_beg + kTile - 1) / kTile) * n_chunks;
        const int t = threadIdx.x;
        size_t urow[N];          // the user rows of the tile do not change with the step: ids read once
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int r = min(u0 + ((i * 256 + t) >> 4), a.B - 1);
            urow[i] = (size_t)(a.user_ids ? a.user_ids[r] : (int64_t)r) * a.d;
        }
        float4 vu[N], vi[N];
        auto request = [&](int step_) {
            const int step = min(step_, n_steps - 1);
            const int i0 = i_beg + (step / n_chunks) * kTile, k = (step % n_chunks) * kKC + (t & 15) * 4;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                vu[i] = *reinterpret_cast<const float4 *>(a.U + urow[i] + k);
                vi[i] = *reinterpret_cast<const float4 *>(a.It + (size_t)min(i0 + ((i * 256 + t) >> 4), a.I - 1) * a.d + k);
            }
        };
        auto to_lds = [&](float *dst, const float4 (&v)[N]) {
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const int f = i * 256 + t;
                float *o = dst + (f >> 4) * kLdsRow + (f & 15) * 4;
                *reinterpret_cast<float2 *>(o) = make_float2(v[i].x, v[i].z);
                *reinterpret_cast<float2 *>(o + 2) = make_float2(v[i].y, v[i].w);
            }
        };
        f32x16 acc[4];
        request(0);
        for (int step = 0; step < n_steps; ++step) {
            const int chunk = step % n_chunks;
            if (chunk == 0) {
#pragma unroll
                for (int n = 0; n < 4; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        acc[n][r] = 0.0f;
            }
            __syncthreads();          // the previous step's LDS reads are done (and scnt is zero before the first epilogue)
            to_lds(ldsU, vu);
            to_lds(ldsI, vi);
            __syncthreads();
            request(step + 1);
            mfma_chunk(acc);
            if (chunk == n_chunks - 1)
                epilogue(acc, i_beg + (step / n_chunks) * kTile);
        }
    } else {
        bool first = true;
        for (int i0 = i_beg; i0 < i_end; i0 += kTile) {
            f32x16 acc[4];
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[n][r] = 0.0f;
            for (int k0 = 0; k0 < a.d; k0 += kKC) {
                if (!first)
                    __syncthreads();
                first = false;
                stage_tile<FULLK>(ldsU, a.U, a.user_ids, u0, a.B, k0, a.d);
                stage_tile<FULLK>(ldsI, a.It, nullptr, i0, a.I, k0, a.d);
                __syncthreads();
                mfma_chunk(acc);
            }
            epilogue(acc, i0);
        }
    }
    __syncthreads();
    if (threadIdx.x < kTile && u0 + threadIdx.x < a.B) {
        const int c = scnt[threadIdx.x];
        int *out = a.counts + ((size_t)(u0 + threadIdx.x) * a.S + split) * 2;
        out[0] = min(c, a.cap2);
        out[1] = c <= a.cap2 ? 0 : (c <= cap ? c - a.cap2 : a.cap2 + 1);   // beyond both halves: reported as an overflow
    }
}

// ------------------------------------------------------------------------------------------------
// train-item mask: one wave per user row
// item_div > 1: S holds a strided item sample (column c = item c * item_div); other items are skipped
__global__ __launch_bounds__(256) void k_mask(float *__restrict__ S, int64_t lds, int B, int I,
                                              const int *__restrict__ mask_rowptr, const int *__restrict__ mask_items,
                                              int item_div)
{
    const int b = uniform(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (b >= B)
        return;
    const int beg = mask_rowptr[b], end = mask_rowptr[b + 1];
    for (int e = beg + lane_id(); e < end; e += kWave) {
        int it = mask_items[e];
        if (item_div > 1) {
            if (it % item_div)
                continue;
            it /= item_div;
        }
        if (it >= 0 && it < I)
            S[(size_t)b * lds + it] = -INFINITY;
    }
}

// ------------------------------------------------------------------------------------------------
// one workgroup (4 waves) per row: each wave scans a quarter of the row, wave 0 merges
__global__ __launch_bounds__(256) void k_topk(const float *__restrict__ S, int64_t lds, int B, int I, int k, int do_round,
                                              float *__restrict__ out_val, int64_t *__restrict__ out_idx)
{
    __shared__ float sv[4][kWave];
    __shared__ int si[4][kWave];
    const int b = blockIdx.x;
    const int lane = lane_id();
    const int w = threadIdx.x >> 6;
    const float *__restrict__ row = S + (size_t)b * lds;
    TopList e{-INFINITY, INT_MAX};
    // quarter boundaries on multiples of 4 so that aligned float4 reads stay inside one quarter
    const int per = (((I + 3) / 4 + 3) / 4) * 4;
    const int beg = min(I, w * per), end = min(I, beg + per);
    const bool vec_ok = (lds & 3) == 0 && ((size_t)row & 15) == 0;
    for (int base = beg; base < end; base += kWave * 4) {
        const int i0 = base + lane * 4;
        float x[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        if (vec_ok && i0 + 3 < end) {
            const float4 t = *reinterpret_cast<const float4 *>(row + i0);
            x[0] = t.x, x[1] = t.y, x[2] = t.z, x[3] = t.w;
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + u < end)
                    x[u] = row[i0 + u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            list_offer(e, x[u], i0 + u, i0 + u < end, k, lane);
    }
    sv[w][lane] = e.v;
    si[w][lane] = e.i;
    __syncthreads();
    if (w != 0)
        return;
    for (int o = 1; o < 4; ++o)
        list_offer(e, sv[o][lane], si[o][lane], lane < k && si[o][lane] != INT_MAX, k, lane);
    if (lane < k) {
        float v = e.v;
        if (do_round)
            v = round4(v);
        out_val[(size_t)b * k + lane] = v;
        out_idx[(size_t)b * k + lane] = e.i;
    }
}

// narrow rows (I <= 4096): one wave per row, no LDS merge
__global__ __launch_bounds__(256) void k_topk_wave(const float *__restrict__ S, int64_t lds, int B, int I, int k, int do_round,
                                                   float *__restrict__ out_val, int64_t *__restrict__ out_idx)
{
    const int b = uniform(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (b >= B)
        return;
    const int lane = lane_id();
    const float *__restrict__ row = S + (size_t)b * lds;
    TopList e{-INFINITY, INT_MAX};
    const bool vec_ok = (lds & 3) == 0 && ((size_t)row & 15) == 0;
    // rows of <= 4096 floats: 16 sweeps of 256 at most; all loads of 8 sweeps are issued before the first offer
    constexpr int kSweeps = 8;
    for (int base = 0; base < I; base += kWave * 4 * kSweeps) {
        float x[kSweeps][4];
#pragma unroll
        for (int sw = 0; sw < kSweeps; ++sw) {
            const int i0 = base + sw * kWave * 4 + lane * 4;
            x[sw][0] = x[sw][1] = x[sw][2] = x[sw][3] = -INFINITY;
            if (vec_ok && i0 + 3 < I) {
                const float4 t = *reinterpret_cast<const float4 *>(row + i0);
                x[sw][0] = t.x, x[sw][1] = t.y, x[sw][2] = t.z, x[sw][3] = t.w;
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (i0 + u < I)
                        x[sw][u] = row[i0 + u];
            }
        }
#pragma unroll
        for (int sw = 0; sw < kSweeps; ++sw) {
            const int i0 = base + sw * kWave * 4 + lane * 4;
            if (base + sw * kWave * 4 >= I)
                break;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                list_offer(e, x[sw][u], i0 + u, i0 + u < I, k, lane);
        }
    }
    if (lane < k) {
        out_val[(size_t)b * k + lane] = do_round ? round4(e.v) : e.v;
        out_idx[(size_t)b * k + lane] = e.i;
    }
}

// ------------------------------------------------------------------------------------------------
// pairwise dots: one lane per pair, k-ordered chain (tiny workloads: a training batch)
__global__ __launch_bounds__(256) void k_score_pairwise(const float *__restrict__ U, const int64_t *__restrict__ users,
                                                        const float *__restrict__ V, const int64_t *__restrict__ items,
                                                        int64_t n, int d, float *__restrict__ out)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n)
        return;
    const float *u = U + (size_t)(users ? users[r] : r) * d;
    const float *v = V + (size_t)(items ? items[r] : r) * d;
    float s = 0.0f;
    for (int k = 0; k < d; ++k)
        s = fmaf(u[k], v[k], s);
    out[r] = s;
}

// ------------------------------------------------------------------------------------------------
// per-user candidate scores (dynamic negative sampling): out[b, j] = <U[users[b]], It[cand[b, j]]>, train items
// of the user masked to -inf.  One wave per (user, 64 candidates); lane = candidate, k-ordered chain.
__global__ __launch_bounds__(256) void k_score_candidates(const float *__restrict__ U, const int64_t *__restrict__ users,
                                                          const float *__restrict__ It, const int64_t *__restrict__ cand,
                                                          int B, int m, int d, const int *__restrict__ mask_rowptr,
                                                          const int *__restrict__ mask_items, float *__restrict__ out)
{
    extern __shared__ float su[];   // 4 user rows
    const int w = threadIdx.x >> 6;
    const int lane = lane_id();
    const int b = blockIdx.y * 4 + w;
    const bool row_ok = b < B;
    const int64_t u = row_ok ? users[b] : 0;
    float *urow = su + w * d;
    for (int k = lane; k < d; k += kWave)
        urow[k] = row_ok ? U[(size_t)u * d + k] : 0.0f;
    __syncthreads();
    const int j = blockIdx.x * kWave + lane;
    if (!row_ok || j >= m)
        return;
    const int64_t item = cand[(size_t)b * m + j];
    const float *__restrict__ p = It + (size_t)item * d;
    float s = 0.0f;
    for (int k = 0; k < d; ++k)
        s = fmaf(urow[k], p[k], s);
    if (mask_rowptr && sorted_contains(mask_items, mask_rowptr[u], mask_rowptr[u + 1], (int)item))
        s = -INFINITY;
    out[(size_t)b * m + j] = s;
}

}  // namespace
}  // namespace tgcn

using namespace tgcn;

int tgcn::launch_score_dense(const float *U, const int64_t *user_ids, int B, const float *It, int I, int d, int item_mul,
                             float *S, int64_t lds, hipStream_t stream)
{
    DenseArgs a{U, user_ids, It, S, lds, B, I, d, item_mul};
    const dim3 grid((I + kTile - 1) / kTile, (B + kTile - 1) / kTile);
    TGCN_REQUIRE(grid.y <= 65535, "B too large for one launch (max 65535*128 rows)");
    if (d % kKC == 0)
        hipLaunchKernelGGL(k_score_dense<true>, grid, dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL(k_score_dense<false>, grid, dim3(256), 0, stream, a);
    return check_launch("k_score_dense");
}

int tgcn::launch_score_dense_filter(const float *U, const int64_t *user_ids, int B, const float *It, int I, int d, const float *tau,
                                    int tau_stride, void *logs, int *counts, int S, int items_per_split, int cap2, hipStream_t stream)
{
    TGCN_REQUIRE(items_per_split > 0 && items_per_split % kTile == 0, "items_per_split must be a positive multiple of 128");
    DenseFilterArgs a{U, user_ids, It, tau, tau_stride, static_cast<float2 *>(logs), counts, B, I, d, S, items_per_split, cap2};
    const dim3 grid(S, (B + kTile - 1) / kTile);
    TGCN_REQUIRE(grid.y <= 65535, "B too large for one launch (max 65535*128 rows)");
    if (d % kKC == 0)
        hipLaunchKernelGGL(k_score_dense_filter<true>, grid, dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL(k_score_dense_filter<false>, grid, dim3(256), 0, stream, a);
    return check_launch("k_score_dense_filter");
}

int tgcn::launch_topk(const float *S, int64_t lds, int B, int I, int k, int do_round, float *out_val, int64_t *out_idx,
                      hipStream_t stream)
{
    if (I <= 4096)
        hipLaunchKernelGGL(k_topk_wave, dim3((B + 3) / 4), dim3(256), 0, stream, S, lds, B, I, k, do_round, out_val, out_idx);
    else
        hipLaunchKernelGGL(k_topk, dim3(B), dim3(256), 0, stream, S, lds, B, I, k, do_round, out_val, out_idx);
    return check_launch("k_topk");
}

int tgcn::launch_mask(float *S, int64_t lds, int B, int I, const int *mask_rowptr, const int *mask_items, int item_div,
                      hipStream_t stream)
{
    hipLaunchKernelGGL(k_mask, dim3((B + 3) / 4), dim3(256), 0, stream, S, lds, B, I, mask_rowptr, mask_items, item_div);
    return check_launch("k_mask");
}

extern "C" int tgcn_score_dense_f32(const float *U, const int64_t *user_ids, int32_t B, const float *It, int32_t I,
                                    int32_t d, float *S, int64_t lds, tgcn_stream_t stream)
{
    TGCN_REQUIRE(B >= 0 && I >= 0, "negative size");
    TGCN_REQUIRE(d > 0 && d <= 4096, "d out of range");
    if (B == 0 || I == 0)
        return TGCN_OK;
    TGCN_REQUIRE(U && It && S, "NULL pointer");
    TGCN_REQUIRE(lds >= I, "lds < I");
    return launch_score_dense(U, user_ids, B, It, I, d, 1, S, lds, static_cast<hipStream_t>(stream));
}

extern "C" int tgcn_mask_f32(float *S, int64_t lds, int32_t B, int32_t I, const int32_t *mask_rowptr,
                             const int32_t *mask_items, tgcn_stream_t stream)
{
    TGCN_REQUIRE(B >= 0 && I >= 0, "negative size");
    if (B == 0 || I == 0)
        return TGCN_OK;
    TGCN_REQUIRE(S && mask_rowptr, "NULL pointer");
    TGCN_REQUIRE(lds >= I, "lds < I");
    return launch_mask(S, lds, B, I, mask_rowptr, mask_items, 1, static_cast<hipStream_t>(stream));
}

extern "C" int tgcn_topk_f32(const float *S, int64_t lds, int32_t B, int32_t I, int32_t k, int32_t round4,
                             float *out_val, int64_t *out_idx, tgcn_stream_t stream)
{
    TGCN_REQUIRE(B >= 0, "negative B");
    TGCN_REQUIRE(k >= 1 && k <= 64, "k must be in [1, 64]");
    TGCN_REQUIRE(I >= k, "k exceeds the number of items");
    if (B == 0)
        return TGCN_OK;
    TGCN_REQUIRE(S && out_val && out_idx, "NULL pointer");
    TGCN_REQUIRE(lds >= I, "lds < I");
    return launch_topk(S, lds, B, I, k, round4, out_val, out_idx, static_cast<hipStream_t>(stream));
}

extern "C" int tgcn_score_pairwise_f32(const float *U, const int64_t *users, const float *V, const int64_t *items,
                                       int64_t n, int32_t d, float *out, tgcn_stream_t stream)
{
    TGCN_REQUIRE(n >= 0, "negative n");
    TGCN_REQUIRE(d > 0, "d must be positive");
    if (n == 0)
        return TGCN_OK;
    TGCN_REQUIRE(U && V && out, "NULL pointer");
    hipLaunchKernelGGL(k_score_pairwise, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), U, users, V, items, n, d, out);
    return check_launch("k_score_pairwise");
}

extern "C" int tgcn_score_candidates_f32(const float *U, const int64_t *users, const float *It, const int64_t *cand,
                                         int32_t B, int32_t m, int32_t d, const int32_t *mask_rowptr,
                                         const int32_t *mask_items, float *out, tgcn_stream_t stream)
{
    TGCN_REQUIRE(B >= 0 && m >= 0, "negative size");
    TGCN_REQUIRE(d > 0 && d <= 4096, "d out of range");
    if (B == 0 || m == 0)
        return TGCN_OK;
    TGCN_REQUIRE(U && users && It && cand && out, "NULL pointer");
    TGCN_REQUIRE(!mask_rowptr || mask_items, "mask_rowptr without mask_items");
    const dim3 grid((m + kWave - 1) / kWave, (B + 3) / 4);
    TGCN_REQUIRE(grid.y <= 65535, "B too large for one launch");
    hipLaunchKernelGGL(k_score_candidates, grid, dim3(256), (size_t)4 * d * sizeof(float), static_cast<hipStream_t>(stream), U,
                       users, It, cand, B, m, d, mask_rowptr, mask_items, out);
    return check_launch("k_score_candidates");
}
