// Exact top-k from a cheap first pass: the bf16 matrix pipe finds the candidates, the fp32 chains score them.
//
// tgcn_score_topk_f32 (tgcn_score_fused.hip) spends ~80 % of a call in the fp32 MFMA GEMM whose only use is the test
// `score > tau_u`.  Here that test runs on an APPROXIMATE score with a threshold lowered by a proven error bound, so the set
// it keeps is a superset of {i : score(u, i) > tau_u}; every kept pair is then rescored with the k-ordered fp32 fmaf chain
// (the chain the MFMA 32x32x2 f32 path, the dense path and the CPU restatement all compute) and dropped again unless
// score > tau_u.  What reaches k_select is therefore the SAME candidate set with the SAME fp32 scores as in the fp32-filter
// path: results are bit-identical, only the cost of finding the candidates changes (gfx950: v_mfma_f32_32x32x16_bf16 retires
// 16x the k-steps of v_mfma_f32_32x32x2_f32 per cycle).
//
// The bound.  a = bf16(x) by round-to-nearest-even (v_cvt_pk_bf16_f32): |a - x| <= 2^-8 max(|x|, 2^-50) (8 significant bits;
// the floor covers fp32 denormals flushed on conversion).  With x~ = max(|x|, 2^-50) elementwise,
//   |sum_j a_j b_j - sum_j x_j y_j| <= (2^-7 + 2^-16) sum_j x~_j y~_j <= (2^-7 + 2^-16) |x~|_2 |y~|_2       (Cauchy-Schwarz)
// the matrix pipe's fp32 accumulation of the d <= 256 exact bf16 products and the fp32 chain's own rounding add at most
// 2^-11 sum_j |x_j y_j| between them (budgeted ~30x above d 2^-23 + d 2^-24; products below 2^-126 that flush are below the
// floor's 2^-108).  So with c = 2^-7 (1 + 2^-4) and M = max_i |y~_i|_2:
//   score(u, i) > tau_u   ==>   approx(u, i) > tau_u - c |x~_u|_2 M =: tau'_u.
// Norms are fp32 sums of squares of the floored values (never underestimated by more than 2^-20 relatively: inside c's
// slack).  Non-finite data: a non-finite tau' becomes -inf (everything is logged -> log overflow -> the exact fallback), and the
// test is !(approx <= tau') so that a NaN approximation (inf - inf in bf16 only) is kept and decided by its fp32 score.
#include "tgcn_internal.h"
#include "tgcn_topk.h"

namespace tgcn {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

constexpr int kUsersPerWG = 128;   // 4 waves x 32 users (the log layout of tgcn_score_fused.hip)
constexpr int kStage = 64;         // items per LDS stage

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi)
{
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));   // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
}

__device__ __forceinline__ float wave_sum_f(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_xor(v, o);
    return v;
}

// ---- item norms: part[g] = max over workgroup g's rows of sum_j max(|y_j|, floor)^2 ----------------------------------------
// G = d / 4 lanes per row (a power of two <= 32): one 16-byte piece per lane, 64 / G rows per wave instruction, four in flight
template <int G>
__device__ __forceinline__ float norm_rows_pow2(const float *__restrict__ It, int I, int d, int wave, int n_waves, int lane)
{
    constexpr int R = kWave / G;
    const int sub = lane / G, q = lane % G;
    float best = 0.0f;
    for (int r0 = wave * 4 * R; r0 < I; r0 += n_waves * 4 * R) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            v[u] = *reinterpret_cast<const float4 *>(It + (size_t)min(r0 + u * R + sub, I - 1) * d + 4 * q);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float t = (floored_sq(v[u].x) + floored_sq(v[u].y)) + (floored_sq(v[u].z) + floored_sq(v[u].w));
#pragma unroll
            for (int o = G / 2; o > 0; o >>= 1)
                t += __shfl_xor(t, o);
            best = nan_max(best, t);
        }
    }
    return best;
}

__global__ __launch_bounds__(256) void k_item_norm_part(const float *__restrict__ It, int I, int d, float *__restrict__ part,
                                                        unsigned *__restrict__ total)
{
    __shared__ float sm[4];
    const int lane = lane_id();
    const int w = threadIdx.x >> 6;
    const int wave = blockIdx.x * 4 + w, n_waves = gridDim.x * 4;
    float best = 0.0f;
    if (d == 64)
        best = norm_rows_pow2<16>(It, I, d, wave, n_waves, lane);
    else if (d == 128)
        best = norm_rows_pow2<32>(It, I, d, wave, n_waves, lane);
    else if (d == 32)
        best = norm_rows_pow2<8>(It, I, d, wave, n_waves, lane);
    else if (d == 16)
        best = norm_rows_pow2<4>(It, I, d, wave, n_waves, lane);
    else {
        for (int r = wave; r < I; r += n_waves) {
            const float *__restrict__ p = It + (size_t)r * d;
            float s = 0.0f;
            for (int k = lane; k < d; k += kWave)
                s += floored_sq(p[k]);
            best = nan_max(best, wave_sum_f(s));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        best = nan_max(best, __shfl_xor(best, o));
    if (lane == 0)
        sm[w] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float b = nan_max(nan_max(sm[0], sm[1]), nan_max(sm[2], sm[3]));
        if (part)
            part[blockIdx.x] = b;
        if (total)   // non-negative floats (and NaNs above them) order as their bit patterns
            atomicMax(total, __float_as_uint(b));
    }
}

// ---- tau' = tau - c |u~| M: one wave per user ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tau_lo(const float *__restrict__ U, const int64_t *__restrict__ user_ids, int B, int d,
                                                const float *__restrict__ tau, int tau_stride, const float *__restrict__ part,
                                                int n_part, float *__restrict__ tau_lo)
{
    const int lane = lane_id();
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B)
        return;
    float m2 = 0.0f;
    for (int i = lane; i < n_part; i += kWave)
        m2 = nan_max(m2, part[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        m2 = nan_max(m2, __shfl_xor(m2, o));
    const float *__restrict__ p = U + (size_t)(user_ids ? user_ids[b] : b) * d;
    float s = 0.0f;
    for (int k = lane; k < d; k += kWave)
        s += floored_sq(p[k]);
    s = wave_sum_f(s);
    if (lane == 0)
        tau_lo[b] = lowered_tau(tau[(size_t)b * tau_stride], s, m2);
}

// ---- the bf16 filter ------------------------------------------------------------------------------------------------------
struct PreArgs {
    const float *__restrict__ U;
    const int64_t *__restrict__ user_ids;
    const float *__restrict__ It;
    const float *__restrict__ tau_lo;
    float2 *__restrict__ logs;      // [B][S][2][cap2]  (approximate score, item-as-float-bits)
    int *__restrict__ counts;       // [B][S][2]
    int B, I, d, S, items_per_split, cap2;
};

// KS = 16-wide k-steps per dot product (d <= 16 KS).  Items on MFMA rows (A operand, LDS stages of ST rows of bf16), the wave's
// 32 users on columns (B operand, KS x 4 registers for the whole pass): C/D col (user) = lane & 31, row (item) =
// (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5), so a lane's 16 results belong to one user and its threshold is one register.
// The kernel is bound by the tests (one compare per result register, an append for the ~1/3 that have a passing lane) and by
// the latency of the item rows, not by the matrix pipe: 2 KS MFMAs of 32 cycles per 64 items.  A stage's rows are requested
// one whole stage ahead (ST = 128: the 64-item stages of the fp32 kernel are over before their successor's rows arrive).
template <int KS, bool FULLK, int ST>
__global__ __launch_bounds__(256) void k_score_prefilter(const PreArgs a)
{
    constexpr int DQ = 4 * KS;                 // float4 pieces per source row
    constexpr int RB = 32 * KS + 16;           // LDS row stride in bytes: bf16 row + 16 (conflict-free ds_read_b128 of a column slice)
    constexpr int N = (ST * DQ) / 256;         // float4 pieces per thread per item stage
    constexpr int NU = (kStage * DQ) / 256;    // ... per 64-user half tile
    static_assert(ST % kStage == 0 && ST >= kUsersPerWG / 2, "the user tile passes through the stage buffers");
    constexpr int kBuf = ST >= kUsersPerWG ? ST : kUsersPerWG;
    __shared__ __attribute__((aligned(16))) unsigned char smem[(ST >= kUsersPerWG ? 2 * ST : kUsersPerWG) * RB];
    (void)kBuf;
    const int lane = lane_id();
    const int w = threadIdx.x >> 6;
    const int r32 = lane & 31;
    const int h = lane >> 5;
    const int u0 = blockIdx.x * kUsersPerWG;
    const int split = blockIdx.y;
    const int i_beg = split * a.items_per_split;
    const int i_end = min(a.I, i_beg + a.items_per_split);

    // M x 256 16-byte pieces of consecutive source rows -> registers (rows past the table clamped to the last one: every load
    // unconditional when FULLK), and from there, as bf16, into LDS rows
    auto load = [&](auto &v, const float *__restrict__ src, const int64_t *__restrict__ ids, int row0, int n_rows) {
        constexpr int M = sizeof(v) / sizeof(float4);
        size_t srow[M];
#pragma unroll
        for (int i = 0; i < M; ++i) {
            const int r = min(row0 + (i * 256 + (int)threadIdx.x) / DQ, n_rows - 1);
            srow[i] = ids ? (size_t)ids[r] : (size_t)r;
        }
#pragma unroll
        for (int i = 0; i < M; ++i) {
            const int k = ((i * 256 + (int)threadIdx.x) % DQ) * 4;
            const float *p = src + srow[i] * a.d;
            if constexpr (FULLK) {
                v[i] = *reinterpret_cast<const float4 *>(p + k);
            } else {
                v[i].x = k + 0 < a.d ? p[k + 0] : 0.0f;
                v[i].y = k + 1 < a.d ? p[k + 1] : 0.0f;
                v[i].z = k + 2 < a.d ? p[k + 2] : 0.0f;
                v[i].w = k + 3 < a.d ? p[k + 3] : 0.0f;
            }
        }
    };
    auto store = [&](unsigned char *dst, const auto &v) {
        constexpr int M = sizeof(v) / sizeof(float4);
#pragma unroll
        for (int i = 0; i < M; ++i) {
            const int f = i * 256 + threadIdx.x;
            const int r = f / DQ, q = f % DQ;
            *reinterpret_cast<uint2 *>(dst + r * RB + q * 8) = make_uint2(pack_bf16(v[i].x, v[i].y), pack_bf16(v[i].z, v[i].w));
        }
    };

    const int user = u0 + w * 32 + r32;
    const bool user_ok = user < a.B;
    float4 nxt[N];
    {
        float4 v[NU];
        load(v, a.U, a.user_ids, u0, a.B);
        store(smem, v);
        load(v, a.U, a.user_ids, u0 + kStage, a.B);
        if (i_beg < i_end)
            load(nxt, a.It, nullptr, i_beg, i_end);
        store(smem + kStage * RB, v);
    }
    const float tau = user_ok ? a.tau_lo[user] : INFINITY;
    __syncthreads();
    bf16x8 bfr[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s)
        bfr[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(smem + (w * 32 + r32) * RB + 32 * s + 16 * h));
    float2 *__restrict__ log = a.logs + ((size_t)(user_ok ? user : 0) * a.S + split) * 2 * a.cap2 + (size_t)h * a.cap2;
    const int cap = user_ok ? a.cap2 : 0;     // padded users never write (their tau is +inf, but a NaN approximation passes the test)
    int cnt = 0;
    __syncthreads();
    if (i_beg >= i_end) {
        if (user_ok)
            a.counts[((size_t)user * a.S + split) * 2 + h] = 0;
        return;
    }
    store(smem, nxt);
    __syncthreads();
    // tau has arrived before the loop: a first use inside it makes hipcc's wait-count pass put s_waitcnt vmcnt(0) in front of
    // EVERY test (the loop-carried state merges the pending tau load with the stage prefetch), which serialises the prefetch
    asm volatile("" ::"v"(tau));
    int buf = 0;
    auto test = [&](float val, int item) {
        if (!(val <= tau)) {
            if (cnt < cap)
                log[cnt] = make_float2(val, __int_as_float(item));
            ++cnt;
        }
    };
    for (int s0 = i_beg; s0 < i_end; s0 += ST) {
        const bool more = s0 + ST < i_end;
        if (more)
            load(nxt, a.It, nullptr, s0 + ST, i_end);
#pragma unroll
        for (int un = 0; un < ST / kStage; ++un) {
            const int t0 = s0 + un * kStage;     // first item of this 64-item unit
            if (un > 0 && t0 >= i_end)
                break;
            const unsigned char *pi = smem + ((buf * ST + un * kStage) + r32) * RB + 16 * h;
            f32x16 c0, c1;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                c0[r] = 0.0f, c1[r] = 0.0f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 a0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(pi + 32 * s));
                const bf16x8 a1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(pi + 32 * RB + 32 * s));
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bfr[s], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bfr[s], c1, 0, 0, 0);
            }
            const int lim = i_end - t0;
            if (lim >= kStage) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    test(c0[r], t0 + (r & 3) + 8 * (r >> 2) + 4 * h);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    test(c1[r], t0 + 32 + (r & 3) + 8 * (r >> 2) + 4 * h);
            } else {   // the partial last unit of the catalogue: rows past i_end are clamped copies
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (row < lim)
                        test(c0[r], t0 + row);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (row < lim)
                        test(c1[r], t0 + row);
                }
            }
        }
        if (more)
            store(smem + (buf ^ 1) * ST * RB, nxt);
        __syncthreads();
        buf ^= 1;
    }
    if (user_ok)
        a.counts[((size_t)user * a.S + split) * 2 + h] = cnt;
}

// ---- fp32 rescoring of the logged pairs -------------------------------------------------------------------------------------
// One workgroup of eight waves per user, each wave takes an eighth of the user's log segments (~40 entries: one tile).  The
// wave lists its entries and their item ids in LDS, then walks them 64 at a time: the 64 item rows are read in k-blocks of 32
// floats with coalesced 16-byte loads (8 lanes per row; the next k-block's loads are in flight while this one is chained),
// transposed through a padded LDS tile, and lane l continues candidate l's chain  s = fmaf(u_k, y_k, s), k ascending -- the
// chain of the MFMA 32x32x2 f32 path and of k_brute_part; u_k comes from scalar loads (the user is the workgroup's).  The
// entry's score is overwritten; an entry with !(s > tau) is retired (item = INT_MAX, which k_select skips).
struct RescoreArgs {
    const float *__restrict__ U;
    const int64_t *__restrict__ user_ids;
    const float *__restrict__ It;
    const float *__restrict__ tau;
    int tau_stride;
    float2 *__restrict__ logs;
    const int *__restrict__ counts;
    int B, d, S, cap2;
};

constexpr int kRescoreWaves = 8;
constexpr int kKB = 32;             // floats of a row per LDS tile
constexpr int kTileRow = kKB + 1;   // padded: lane = row reads are conflict-free

__global__ __launch_bounds__(kRescoreWaves * 64) void k_rescore(const RescoreArgs a)
{
    extern __shared__ float sh[];   // per wave: positions [8 cap2] | item ids [8 cap2] | tile [64][33]
    const int lane = lane_id();
    const int w = threadIdx.x >> 6;
    const int b = blockIdx.x;
    const int per_wave = 16 * a.cap2 + kWave * kTileRow;
    int *pos = reinterpret_cast<int *>(sh + (size_t)w * per_wave);
    int *ids = pos + 8 * a.cap2;
    float *tile = reinterpret_cast<float *>(ids + 8 * a.cap2);
    const float *__restrict__ urow = a.U + (size_t)(a.user_ids ? a.user_ids[b] : (int64_t)b) * a.d;   // wave-uniform
    const int n_seg = a.S * 2;                               // <= 64
    const int spw = (n_seg + kRescoreWaves - 1) / kRescoreWaves;   // segments per wave, <= 8
    const int seg = w * spw + lane;
    const int cnt = (lane < spw && seg < n_seg) ? min(a.counts[(size_t)b * n_seg + seg], a.cap2) : 0;
    int off = cnt;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        const int t = __shfl_up(off, o);
        if (lane >= o)
            off += t;
    }
    const int n = __builtin_amdgcn_readlane(off, 7);
    if (n == 0)
        return;
    off -= cnt;
    for (int j = 0; j < cnt; ++j)
        pos[off + j] = seg * a.cap2 + j;
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    float2 *__restrict__ lg = a.logs + (size_t)b * n_seg * a.cap2;
    for (int j = lane; j < n; j += kWave)
        ids[j] = __float_as_int(lg[pos[j]].y);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    const float tau = a.tau[(size_t)b * a.tau_stride];
    const int sub = lane >> 3, q = lane & 7;   // loader role: row sub + 8 pass, 16-byte piece q of the k-block
    const int nkb = (a.d + kKB - 1) / kKB;
    const int total = ((n + kWave - 1) / kWave) * nkb;

    auto issue = [&](int step, float4 (&v)[8]) {
        const int t0 = (step / nkb) * kWave, k = (step % nkb) * kKB + 4 * q;
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const size_t ro = (size_t)ids[min(t0 + pass * 8 + sub, n - 1)] * a.d;
            v[pass] = k < a.d ? *reinterpret_cast<const float4 *>(a.It + ro + k) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    };
    float s = 0.0f;
    auto process = [&](int step, const float4 (&v)[8]) {
        const int t0 = (step / nkb) * kWave, kb = (step % nkb) * kKB;
        __builtin_amdgcn_wave_barrier();    // the previous step's reads of the tile are done (same wave, in order)
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            float *o = tile + (pass * 8 + sub) * kTileRow + 4 * q;
            o[0] = v[pass].x, o[1] = v[pass].y, o[2] = v[pass].z, o[3] = v[pass].w;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        const float *__restrict__ row = tile + lane * kTileRow;
        if (kb == 0)
            s = 0.0f;
        if (kb + kKB <= a.d) {
#pragma unroll
            for (int kk = 0; kk < kKB; ++kk)
                s = fmaf(urow[kb + kk], row[kk], s);
        } else {
            for (int kk = 0; kk < a.d - kb; ++kk)
                s = fmaf(urow[kb + kk], row[kk], s);
        }
        if (kb + kKB >= a.d && t0 + lane < n) {
            const int p = pos[t0 + lane];
            const int item = ids[t0 + lane];
            lg[p] = make_float2(s, __int_as_float(s > tau ? item : INT_MAX));
        }
    };
    float4 va[8], vb[8];
    issue(0, va);
    for (int step = 0; step < total; step += 2) {
        if (step + 1 < total)
            issue(step + 1, vb);
        process(step, va);
        if (step + 1 < total) {
            if (step + 2 < total)
                issue(step + 2, va);
            process(step + 1, vb);
        }
    }
}

}  // namespace

int launch_item_norm_part(const float *It, int I, int d, float *part, int n_part, float *total, hipStream_t s)
{
    hipLaunchKernelGGL(k_item_norm_part, dim3(n_part), dim3(256), 0, s, It, I, d, part, reinterpret_cast<unsigned *>(total));
    return check_launch("k_item_norm_part");
}

int launch_tau_lo(const float *U, const int64_t *user_ids, int B, int d, const float *tau, int tau_stride, const float *part,
                  int n_part, float *tau_lo, hipStream_t s)
{
    hipLaunchKernelGGL(k_tau_lo, dim3((B + 3) / 4), dim3(256), 0, s, U, user_ids, B, d, tau, tau_stride, part, n_part, tau_lo);
    return check_launch("k_tau_lo");
}

bool prefilter_supports(int d) { return d <= 128 && (d & 3) == 0; }

int launch_prefilter(const float *U, const int64_t *user_ids, int B, const float *It, int I, int d, const float *tau_lo, void *logs,
                     int *counts, int S, int items_per_split, int cap2, hipStream_t s)
{
    PreArgs a{U, user_ids, It, tau_lo, static_cast<float2 *>(logs), counts, B, I, d, S, items_per_split, cap2};
    const dim3 grid((B + kUsersPerWG - 1) / kUsersPerWG, S);
    if (d == 64)
        hipLaunchKernelGGL((k_score_prefilter<4, true, 128>), grid, dim3(256), 0, s, a);
    else if (d < 64)
        hipLaunchKernelGGL((k_score_prefilter<4, false, 128>), grid, dim3(256), 0, s, a);
    else if (d == 128)
        hipLaunchKernelGGL((k_score_prefilter<8, true, 128>), grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((k_score_prefilter<8, false, 128>), grid, dim3(256), 0, s, a);
    return check_launch("k_score_prefilter");
}

int launch_rescore(const float *U, const int64_t *user_ids, int B, const float *It, int d, const float *tau, int tau_stride, void *logs,
                   const int *counts, int S, int cap2, hipStream_t s)
{
    RescoreArgs a{U, user_ids, It, tau, tau_stride, static_cast<float2 *>(logs), counts, B, d, S, cap2};
    const size_t lds = (size_t)kRescoreWaves * (16 * cap2 + kWave * kTileRow) * sizeof(float);
    hipLaunchKernelGGL(k_rescore, dim3(B), dim3(kRescoreWaves * 64), lds, s, a);
    return check_launch("k_rescore");
}

}  // namespace tgcn
