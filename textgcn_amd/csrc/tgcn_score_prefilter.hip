// Exact top-k from a cheap first pass: the bf16 matrix pipe finds the candidates, the fp32 chains score them.
//
// tgcn_score_topk_f32 (tgcn_score_fused.hip) spends ~80 % of a call in the fp32 MFMA GEMM whose only use is the test
// `score > tau_u`.  Here that test runs on an APPROXIMATE score raised by a proven bound on its error, so the set it keeps
// is a superset of {i : score(u, i) > tau_u}; every kept pair is then rescored with the k-ordered fp32 fmaf chain
// (the chain the MFMA 32x32x2 f32 path, the dense path and the CPU restatement all compute) and dropped again unless
// score > tau_u.  What reaches k_select is therefore the SAME candidate set with the SAME fp32 scores as in the fp32-filter
// path: results are bit-identical, only the cost of finding the candidates changes (gfx950: v_mfma_f32_32x32x16_bf16 retires
// 16x the k-steps of v_mfma_f32_32x32x2_f32 per cycle).
//
// The bound.  With a = bf16(x), b = bf16(y) (round-to-nearest-even, v_cvt_pk_bf16_f32) and the residuals r_x = x - a, r_y = y - b
// (exact in fp32):   x.y - a.b = x.r_y + r_x.b,   so   |x.y - a.b| <= |x|_2 |r_y|_2 + |r_x|_2 (|y|_2 + |r_y|_2)    (Cauchy-Schwarz)
// with the ACTUAL residual norms of the two rows (worst case 2^-8 of the row norm, ~0.4 of that on ordinary data).  The matrix
// pipe's fp32 accumulation of the d exact bf16 products and the fp32 chain's own rounding add at most 2^-11 |x|_2 |y|_2
// between them: (d - 1) roundings of at most 2^-23 of the running |sum| (truncation assumed for the pipe) plus d roundings of 2^-24
// in the chain, each sum bounded by |x|_2 |y|_2 -- d 2^-23 + d 2^-24 = 2^-12.4 at the widest row this path takes (d = 1024), ~30x
// below the budget at d <= 128.  So
//   score(u, i) > tau_u   ==>   approx(u, i) + n_u r_i + r_u (n_i + r_i) + 2^-11 n_u n_i > tau_u,
// n = norm of the row (elements floored at 2^-50), r = norm of its residual (floored at 2^-58), each with a 2^-12 margin.  The
// added terms are computed BY the matrix pipe: one more 16-wide k-step whose only non-zero operands are
// [n_u, r_u, 2^-11 n_u] and [r_i, n_i + r_i, n_i], each rounded UP to bf16, so the test stays one compare of the accumulator
// against tau_u.  The bound is per PAIR: one item row of enormous norm becomes a candidate for everybody but does not loosen
// anybody else's test.  Non-finite data: a non-finite factor becomes +inf, the accumulator +inf or NaN, and the test is
// !(acc <= tau) -- the pair is kept and decided by its fp32 score.
//
// In this file: k_item_pack (the item operand, once per table), the threshold sample k_tau ranks -- k_sample_bits (the call's train
// items as a bitmap over the sample) + k_sample_pack_top (rows from the pack) or k_sample_bf16<.., TOP> (fp32 rows): two values per
// user and 128-sample block; k_sample_bf16 / the SAMPLE form of k_score_prefilter_wide: every score -- the LDS-DMA helpers of
// both filters, k_score_prefilter (d <= 128: pass bits), k_score_prefilter_wide (128 < d <= 1024: the users'
// fragments in registers, the passing pairs LOGGED with their raised scores -- k_refine in tgcn_score_fused.hip turns those into a
// second threshold, the k-th largest lower bound, and only what can still reach the top k is rescored), k_rescore (the fp32
// chains, from the pass bits or from k_refine's id lists).
#include <type_traits>

#include "tgcn_internal.h"
#include "tgcn_topk.h"

namespace tgcn {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr int kPreWaves = 8;        // waves (x 32 users) per workgroup of the bf16 filter: every item row a workgroup stages is
                                   // fetched from L2 once per 256 users (with 128, the 16 user tiles of a 2048-user call pull
                                   // 205 MB through L2 for a 12.8 MB table; 16 384-user calls gained 7 %, 2 M items 15 %)
constexpr int kStage = 64;         // items per LDS stage

__device__ __forceinline__ float wave_sum_f(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_xor(v, o);
    return v;
}

// ---- item factors: norms[i] = {n_i, r_i} ---------------------------------------------------------------------------------
// G = d / 4 lanes per row (a power of two <= 32): one 16-byte piece per lane, 64 / G rows per wave instruction, four in flight
template <int G>
__device__ __forceinline__ void norm_rows_pow2(const float *__restrict__ It, int I, int d, int wave, int n_waves, int lane,
                                               float *__restrict__ norms)
{
    constexpr int R = kWave / G;
    const int sub = lane / G, q = lane % G;
    for (int r0 = wave * 4 * R; r0 < I; r0 += n_waves * 4 * R) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            v[u] = *reinterpret_cast<const float4 *>(It + (size_t)min(r0 + u * R + sub, I - 1) * d + 4 * q);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float t = (floored_sq(v[u].x) + floored_sq(v[u].y)) + (floored_sq(v[u].z) + floored_sq(v[u].w));
            float e = (residual_sq(v[u].x) + residual_sq(v[u].y)) + (residual_sq(v[u].z) + residual_sq(v[u].w));
#pragma unroll
            for (int o = G / 2; o > 0; o >>= 1) {
                t += __shfl_xor(t, o);
                e += __shfl_xor(e, o);
            }
            const int r = r0 + u * R + sub;
            if (q == 0 && r < I)
                *reinterpret_cast<float2 *>(norms + 2 * (size_t)r) = make_float2(bound_factor(t), bound_factor(e));
        }
    }
}

__global__ __launch_bounds__(256) void k_item_norms(const float *__restrict__ It, int I, int d, float *__restrict__ norms)
{
    const int lane = lane_id();
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
    if (d == 64)
        norm_rows_pow2<16>(It, I, d, wave, n_waves, lane, norms);
    else if (d == 128)
        norm_rows_pow2<32>(It, I, d, wave, n_waves, lane, norms);
    else if (d == 32)
        norm_rows_pow2<8>(It, I, d, wave, n_waves, lane, norms);
    else if (d == 16)
        norm_rows_pow2<4>(It, I, d, wave, n_waves, lane, norms);
    else {
        for (int r = wave; r < I; r += n_waves) {
            const float *__restrict__ p = It + (size_t)r * d;
            float s = 0.0f, e = 0.0f;
            for (int k = lane; k < d; k += kWave) {
                s += floored_sq(p[k]);
                e += residual_sq(p[k]);
            }
            s = wave_sum_f(s);
            e = wave_sum_f(e);
            if (lane == 0)
                *reinterpret_cast<float2 *>(norms + 2 * (size_t)r) = make_float2(bound_factor(s), bound_factor(e));
        }
    }
}

// ---- the item operand of the filter, packed once per item table --------------------------------------------------------------
// Row i of the pack is the LDS image of item i in the filter's stages: 16 KS bf16 elements (the row, round-to-nearest-even, zeros
// behind d) followed by a 16-byte chunk [r_i, n_i + r_i, n_i] rounded UP to bf16 (the A operand of the bound's k-step), zeros
// behind -- RB = 32 KS + 16 bytes, KS = 4 (d <= 64) or 8 (d <= 128).  A stage of the filter is then ST RB CONTIGUOUS bytes: 16-byte
// loads straight into 16-byte LDS stores, half the bytes of the fp32 rows and no conversion in the loop (the filter took 35 us
// with the fp32 rows converted per stage, 26 with the staging removed altogether -- profiles/r02_experiments.md).
// 16-wide k-steps of a packed row: 4 (d <= 64), 8 (d <= 128), then the widths the wide filter is built for
// (wide rows: the instantiated widths of k_score_prefilter_wide -- 256, 512, 896, 960, 1024 elements)
__host__ __device__ __forceinline__ int pack_ksteps(int d)
{
    return d <= 64 ? 4 : d <= 128 ? 8 : d <= 256 ? 16 : d <= 512 ? 32 : d <= 832 ? 64 : d <= 896 ? 56 : d <= 960 ? 60 : 64;
}

__device__ __forceinline__ uint4 bound_chunk(float n, float r)
{
    return make_uint4(bf16_up_bits(r) | (bf16_up_bits((n + r) * (1.0f + 0x1p-20f)) << 16), bf16_up_bits(n), 0u, 0u);
}

template <int G>      // d = 4 G = 16 KS exactly: G lanes per row, one float4 each
__device__ __forceinline__ void pack_rows_pow2(const float *__restrict__ It, int I, int wave, int n_waves, int lane,
                                               unsigned char *__restrict__ pack)
{
    constexpr int R = kWave / G, d = 4 * G, RB = 8 * G + 16;
    const int sub = lane / G, q = lane % G;
    for (int r0 = wave * 4 * R; r0 < I; r0 += n_waves * 4 * R) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            v[u] = *reinterpret_cast<const float4 *>(It + (size_t)min(r0 + u * R + sub, I - 1) * d + 4 * q);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float t = (floored_sq(v[u].x) + floored_sq(v[u].y)) + (floored_sq(v[u].z) + floored_sq(v[u].w));
            float e = (residual_sq(v[u].x) + residual_sq(v[u].y)) + (residual_sq(v[u].z) + residual_sq(v[u].w));
#pragma unroll
            for (int o = G / 2; o > 0; o >>= 1) {
                t += __shfl_xor(t, o);
                e += __shfl_xor(e, o);
            }
            const int r = r0 + u * R + sub;
            if (r < I) {
                unsigned char *row = pack + (size_t)r * RB;
                *reinterpret_cast<uint2 *>(row + 8 * q) = make_uint2(pack_bf16(v[u].x, v[u].y), pack_bf16(v[u].z, v[u].w));
                if (q == 0)
                    *reinterpret_cast<uint4 *>(row + 8 * G) = bound_chunk(bound_factor(t), bound_factor(e));
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_item_pack(const float *__restrict__ It, int I, int d, unsigned char *__restrict__ pack)
{
    const int lane = lane_id();
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
    if (d == 64) {
        pack_rows_pow2<16>(It, I, wave, n_waves, lane, pack);
    } else if (d == 128) {
        pack_rows_pow2<32>(It, I, wave, n_waves, lane, pack);
    } else {      // any other width: a wave per row
        const int KS = pack_ksteps(d), RB = 32 * KS + 16;
        for (int r = wave; r < I; r += n_waves) {
            const float *__restrict__ p = It + (size_t)r * d;
            unsigned char *row = pack + (size_t)r * RB;
            float s = 0.0f, e = 0.0f;
            for (int k = lane; k < 16 * KS; k += kWave) {
                const float x = k < d ? p[k] : 0.0f;
                if (k < d) {
                    s += floored_sq(x);
                    e += residual_sq(x);
                }
                reinterpret_cast<unsigned short *>(row)[k] = (unsigned short)(pack_bf16(x, 0.0f) & 0xffffu);
            }
            s = wave_sum_f(s);
            e = wave_sum_f(e);
            if (lane == 0)
                *reinterpret_cast<uint4 *>(row + 32 * KS) = bound_chunk(bound_factor(s), bound_factor(e));
        }
    }
}

// ---- user factors {n_u, r_u}: one wave per user (catalogues too large for k_tau, which writes them itself) ------------------
__global__ __launch_bounds__(256) void k_user_bound(const float *__restrict__ U, const int64_t *__restrict__ user_ids, int B, int d,
                                                    float *__restrict__ ubound)
{
    const int lane = lane_id();
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B)
        return;
    const float *__restrict__ p = U + (size_t)(user_ids ? user_ids[b] : b) * d;
    float s = 0.0f, e = 0.0f;
    for (int k = lane; k < d; k += kWave) {
        s += floored_sq(p[k]);
        e += residual_sq(p[k]);
    }
    s = wave_sum_f(s);
    e = wave_sum_f(e);
    if (lane == 0)
        *reinterpret_cast<float2 *>(ubound + 2 * (size_t)b) = make_float2(bound_factor(s), bound_factor(e));
}

// ---- LDS-DMA pieces and counted waits (both bf16 filters) ---------------------------------------------------------------
template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    static_assert(N >= 0 && N < 64, "vmcnt");
    __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | (0x7 << 4) | (0xF << 8));
}

// One LDS-DMA piece: every lane's 16 bytes at `src` land at LDS offset lds_base + 16 lane (global_load_lds_dwordx4; M0 carries the
// wave-uniform LDS offset, one wait state behind the write of M0).  As inline assembly ON PURPOSE: through
// __builtin_amdgcn_global_load_lds hipcc's wait-count pass puts s_waitcnt vmcnt(0) in front of the next ds_read of ANY LDS
// address (it cannot tell the ring's buffers apart), i.e. right behind the request -- no prefetch left.  Hidden from that pass,
// the requests are ordered by hand (wait_vmcnt<N> + the stage barrier).  An uncounted VMEM operation can only make a
// compiler-placed vmcnt wait longer, never shorter: returns are in issue order.
// THE INVARIANT OF THE COUNTED WAITS (both filters).  A wave issues PIECES requests per stage, LEAD stages ahead of the stage that
// multiplies them; at the end of stage s the pieces of stage s + 1 must be in LDS.  s_waitcnt vmcnt(N) returns once at most N
// of the wave's vector-memory operations are outstanding, and loads -- LDS-DMA requests are loads -- complete in issue order
// AMONG THEMSELVES (gfx9: one counter for loads and stores; nothing is assumed about stores against loads).  If a piece of
// stage s + 1 were still outstanding, every request issued after it would be too: the (LEAD - 1) x PIECES requests of stages
// s + 2 .. s + LEAD -- that alone is more than N = (LEAD - 1) x PIECES.  So the wait covers the awaited pieces WHEREVER the
// stage's stores (pass words, log appends, compiler spills) sit among the requests and however they retire: an outstanding store
// only uses up allowance, it makes the wait stricter, never laxer.  What must hold is
//     N  <=  number of this wave's REQUESTS that are younger than the awaited pieces
// and kAllowed below IS that number, derived from the same two constants the request sites use (tools/check_dma.py builds the
// library with -DTGCN_CHECK_DMA: every awaited piece is then compared with its source bytes before the stage is multiplied).
template <int PIECES, int LEAD>
struct DmaRingWait {
    static_assert(LEAD >= 1 && PIECES >= 1, "ring shape");
    static constexpr int kYoungerRequests = (LEAD - 1) * PIECES;   // requests of stages s + 2 .. s + LEAD, all issued after stage s + 1's
    static constexpr int kAllowed = kYoungerRequests;
    static_assert(kAllowed < 64, "vmcnt is a 6-bit field");
    __device__ static __forceinline__ void wait() { wait_vmcnt<kAllowed>(); }
};

#ifdef TGCN_CHECK_DMA
// Checked build only (never in libtgcn.so; tools/check_dma.py): counts the 16-byte pieces a consumer found stale.  Before a ring
// buffer is requested into, the requesting lanes overwrite their pieces' places with a poison pattern (and wait for those LDS
// writes), so the previous stage's bytes can never pass for the new ones; after the counted wait, before the stage barrier, every
// lane compares its pieces of the awaited stage with their source bytes, read again by an ordinary load.
__device__ unsigned long long g_dma_stale_pieces;
__device__ unsigned long long g_dma_checked_pieces;
__device__ __forceinline__ void dma_poison(unsigned lds_addr)
{
    const u32x4 poison = {0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu};
    *reinterpret_cast<__attribute__((address_space(3))) u32x4 *>(lds_addr) = poison;
}
__device__ __forceinline__ void dma_verify(const void *src, unsigned lds_addr)
{
    const u32x4 want = *reinterpret_cast<const u32x4 *>(src);
    const u32x4 got = *reinterpret_cast<const __attribute__((address_space(3))) u32x4 *>(lds_addr);
    atomicAdd(&g_dma_checked_pieces, 1ull);
    if (want.x != got.x || want.y != got.y || want.z != got.z || want.w != got.w)
        atomicAdd(&g_dma_stale_pieces, 1ull);
}
#endif

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"      // "clobber list contains reserved registers: m0" -- it is written here
__device__ __forceinline__ void lds_dma16(const void *src, unsigned lds_base)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_base), "v"(src) : "memory", "m0");
}
#pragma clang diagnostic pop

// ---- the bf16 filter ------------------------------------------------------------------------------------------------------
struct PreArgs {
    const float *__restrict__ U;
    const int64_t *__restrict__ user_ids;
    const unsigned char *__restrict__ ipack;   // [I][RB] packed item rows (k_item_pack)
    size_t pack_bytes;
    const float *__restrict__ tau;      // tau of user b at tau[b * tau_stride]
    int tau_stride;
    const float *__restrict__ ubound;   // [B][2] {n_u, r_u}  (k_tau / k_user_bound)
    unsigned *__restrict__ mask;    // [B padded to 256][2][Wh] pass bits: word (user, h, unit), register t of the unit on bit 31 - t
    int Wh;
    int B, I, d, items_per_split;
    int n_tiles, n_splits;          // user tiles of 256 x item splits; the grid is linear: 8 ceil(n_tiles n_splits / 8) workgroups
    // WIDE form, large catalogues: one bit per stage and (user, row half) -- "some word of the stage is non-zero" --
    // [B padded][2][n_splits][summ_sw] words, bit ls & 31 of word ls >> 5 for the split's stage ls; NULL: not kept.
    // k_rescore then reads 1 / (32 x words per stage) of the pass-bit area and the words of the flagged stages only (config 4: a
    // user's row is 250 KB of words holding ~200 set bits -- the scan was most of k_rescore's 175 us there).
    unsigned *__restrict__ summ;
    int summ_sw;
};

// KS = 16-wide k-steps per dot product (d <= 16 KS).  Items on MFMA rows (A operand, LDS stages of ST rows of bf16), the wave's
// 32 users on columns (B operand, KS x 4 registers for the whole pass): C/D col (user) = lane & 31, row (item) =
// (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5), so a lane's 16 results belong to one user and its threshold is one register.
// Output: one pass bit per (user, item), a 32-bit word per lane and 64-item unit (no branch, no append in the loop: with
// lane-private logs the loop spent ~2000 issue cycles per unit on compare-and-branch and appends against 256 cycles of MFMA).
// A stage's rows are requested one whole stage ahead, as the contiguous bytes of the item pack; d <= 64 walks 256-item stages.
// (Stages converted from the fp32 table inside the loop: 33 / 180 us at 2048 / 16 384 users, d = 64; from the pack 28 / 145;
// d = 128: 57 / 343 -> 45 / 300.)
// RING = 3: a stage is requested two stages ahead (the form every call size can afford).  RING = 2 with stages TWICE as long (512
// items at d <= 64, 256 at d <= 128; 144 KB of LDS), requested one stage ahead -- a stage then lasts ~3 us, more than a round trip:
// for calls with many stages per workgroup.  Cycle stamps (profiles/r04_experiments.md section 12): of a 256-item stage's 4519
// cycles per wave 1288 are its boundary -- the barrier 662, the burst of requests all eight waves issue behind it 516, the counted
// wait 110 -- during which the SIMD multiplies nothing; twice the items per boundary.
template <int KS, bool FULLK, int ST, int WAVES, bool WIDE, int RING = 3>
__global__ __launch_bounds__(WAVES * 64) void k_score_prefilter(const PreArgs a)
{
    static_assert(RING == 3 || (RING == 2 && WIDE), "ring of two: the one-store-per-stage form");
    constexpr int T = WAVES * 64;              // threads; WAVES x 32 users per workgroup share every item stage
    constexpr int UT = WAVES * 32;
    constexpr int DQ = 4 * KS;                 // float4 pieces per source row
    constexpr int RB = 32 * KS + 16;           // LDS row stride in bytes: bf16 row + 16 (conflict-free ds_read_b128 of a column slice)
    constexpr int NP = (ST * RB / 16 + T - 1) / T;   // 16-byte pieces of a packed item stage per thread (the last round reaches
    constexpr int SB = NP * T * 16;                  // past the stage: the buffers are SB bytes apart and the tail is never read)
    constexpr int NU = (kStage * DQ) / T;      // ... per 64-user piece of the user tile
    static_assert(ST % kStage == 0 && 2 * ST >= UT && (kStage * DQ) % T == 0, "the user tile passes through the stage buffers");
    static_assert(SB >= ST * RB && 2 * SB >= UT * RB, "stage buffers (the user tile passes through both)");
    static_assert(RING * SB <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(16))) unsigned char smem[RING * SB];      // the ring: one stage multiplied, RING - 1 on their way
    const int lane = lane_id();
    const int w = threadIdx.x >> 6;
    const int r32 = lane & 31;
    const int h = lane >> 5;
    // linear grid, XCD-affine: workgroups are dealt to the 8 XCDs round-robin, and XCD x takes the (split, user tile) pairs
    // [x per, (x + 1) per) in split-major order -- the user tiles of a split run side by side on ONE XCD and share its rows in that
    // L2 (2048 users x 2 M items: every XCD streamed the whole 288 MB pack, 2.3 GB per launch)
    const int per = (a.n_splits * a.n_tiles + 7) / 8;
    const int pair = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || pair >= a.n_splits * a.n_tiles)
        return;
    const int u0 = (pair % a.n_tiles) * UT;
    const int split = pair / a.n_tiles;
    const int i_beg = split * a.items_per_split;
    const int i_end = min(a.I, i_beg + a.items_per_split);

    // M x T 16-byte pieces of consecutive source rows -> registers (rows past the table clamped to the last one: every load
    // unconditional when FULLK), and from there, as bf16, into LDS rows
    auto load = [&](auto &v, const float *__restrict__ src, const int64_t *__restrict__ ids, int row0, int n_rows) {
        constexpr int M = sizeof(v) / sizeof(float4);
        size_t srow[M];
#pragma unroll
        for (int i = 0; i < M; ++i) {
            const int r = min(row0 + (i * T + (int)threadIdx.x) / DQ, n_rows - 1);
            srow[i] = ids ? (size_t)ids[r] : (size_t)r;
        }
#pragma unroll
        for (int i = 0; i < M; ++i) {
            const int k = ((i * T + (int)threadIdx.x) % DQ) * 4;
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
            const int f = i * T + threadIdx.x;
            const int r = f / DQ, q = f % DQ;
            *reinterpret_cast<uint2 *>(dst + r * RB + q * 8) = make_uint2(pack_bf16(v[i].x, v[i].y), pack_bf16(v[i].z, v[i].w));
        }
    };

    // an item stage: ST RB contiguous bytes of the pack (the rows already bf16, the factors of the bound -- [r_i, n_i + r_i, n_i]
    // rounded up to bf16 -- in each row's 16-byte pad chunk), copied piece for piece.  Every load unconditional: pieces past the
    // stage are the next rows (or, at the end of the table, its last 16 bytes) and land in the buffer's unread tail.
    auto load_stage = [&](u32x4 (&v)[NP], int row0) {
        const size_t base = (size_t)row0 * RB + (size_t)threadIdx.x * 16;
#pragma unroll
        for (int i = 0; i < NP; ++i)
            v[i] = *reinterpret_cast<const u32x4 *>(a.ipack + min(base + (size_t)i * T * 16, a.pack_bytes - 16));
    };
    // inside the loop the next stage goes from the pack straight into the other LDS buffer (LDS-DMA, 16 bytes per lane, a wave
    // instruction = 1 KB contiguous): no staging registers, no ds_write, and the buffer is free -- it was multiplied in the previous
    // stage and every wave has passed that stage's barrier.  Issued at the stage's start, waited for at its end (vmcnt(0) + barrier).
    // With the rows staged through registers the stage traffic cost a quarter of the launch (round 3's ablation build pre_nostage: 155 ->
    // 116 us at 16 384 users).
    constexpr int kLead = RING - 1;      // stages between a request and the stage that multiplies it (the ring has kLead + 1 buffers)
    using RingWait = DmaRingWait<NP, kLead>;
    auto dma_stage = [&](int nb, int row0) {
        const size_t base = (size_t)row0 * RB;
        const unsigned lds0 = (unsigned)(uintptr_t)(smem + nb * SB);
#ifdef TGCN_CHECK_DMA
#pragma unroll
        for (int i = 0; i < NP; ++i)
            dma_poison(lds0 + (unsigned)(i * T + (int)threadIdx.x) * 16u);
        __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): the poison is in LDS before the first request is issued
#endif
#pragma unroll
        for (int i = 0; i < NP; ++i)
            lds_dma16(a.ipack + min(base + (size_t)(i * T + (int)threadIdx.x) * 16, a.pack_bytes - 16),
                      uniform((int)(lds0 + (unsigned)(i * T + w * kWave) * 16u)));
    };
#ifdef TGCN_CHECK_DMA
    auto verify_stage = [&](int nb, int row0) {      // this lane's pieces of the stage in ring buffer nb against their source bytes
        const size_t base = (size_t)row0 * RB;
        const unsigned lds0 = (unsigned)(uintptr_t)(smem + nb * SB);
#pragma unroll
        for (int i = 0; i < NP; ++i)
            dma_verify(a.ipack + min(base + (size_t)(i * T + (int)threadIdx.x) * 16, a.pack_bytes - 16),
                       lds0 + (unsigned)(i * T + (int)threadIdx.x) * 16u);
    };
#endif
    auto store_stage = [&](unsigned char *dst, const u32x4 (&v)[NP]) {
#pragma unroll
        for (int i = 0; i < NP; ++i)
            *reinterpret_cast<u32x4 *>(dst + (i * T + threadIdx.x) * 16) = v[i];
    };

    const int user = u0 + w * 32 + r32;
    const bool user_ok = user < a.B;
    u32x4 nxt[NP];
    {   // the whole user tile and the first item stage are requested together: one round trip for the ids, one for the rows
        float4 v[UT / kStage][NU];
#pragma unroll
        for (int piece = 0; piece < UT / kStage; ++piece)
            load(v[piece], a.U, a.user_ids, u0 + piece * kStage, a.B);
        load_stage(nxt, min(i_beg, a.I - 1));
#pragma unroll
        for (int piece = 0; piece < UT / kStage; ++piece)
            store(smem + piece * kStage * RB, v[piece]);
    }
    const float tau = user_ok ? a.tau[(size_t)user * a.tau_stride] : INFINITY;
    // the extra k-step's B operand: [n_u, r_u, 2^-11 n_u] in the first three elements of the h = 0 half, zeros elsewhere (both
    // halves of the A operand read the row's pad chunk, so the step adds n_u r_i + r_u (n_i + r_i) + 2^-11 n_u n_i)
    const float2 ub = user_ok ? *reinterpret_cast<const float2 *>(a.ubound + 2 * (size_t)user) : make_float2(0.0f, 0.0f);
    const bf16x8 bfx = __builtin_bit_cast(bf16x8, h == 0 ? make_uint4(bf16_up_bits(ub.x) | (bf16_up_bits(ub.y) << 16),
                                                                       bf16_up_bits(ub.x * kAccumBudget), 0u, 0u)
                                                         : make_uint4(0u, 0u, 0u, 0u));
    __syncthreads();
    bf16x8 bfr[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s)
        bfr[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(smem + (w * 32 + r32) * RB + 32 * s + 16 * h));
    unsigned *__restrict__ mrow = a.mask + ((size_t)user * 2 + h) * a.Wh;   // rows of padded users exist and are never read
    __syncthreads();
    if (i_beg >= i_end)
        return;
    store_stage(smem, nxt);
    if constexpr (kLead == 2)
        dma_stage(1, i_beg + ST);      // (the user tile is out of the buffers: its fragments are in registers)
    __syncthreads();
    // tau has arrived before the loop: a first use inside it makes hipcc's wait-count pass put s_waitcnt vmcnt(0) in front of
    // EVERY test (the loop-carried state merges the pending tau load with the stage prefetch), which serialises the prefetch
    asm volatile("" ::"v"(tau), "v"(bfx));
    int buf = 0;
    // (Requesting the rows TWO stages ahead, with a second register set, was measured: no change -- 33.2 vs 32.4 us at d = 64,
    // 56.1 vs 55.4 at d = 128.  The stage registers are a native vector type: as an array of HIP's uint4 structs the compiler
    // kept them in scratch memory and waited for every load right behind its issue -- 46 us instead of 28.)
    // one 64-item unit: c0/c1 <- its accumulators; the pass bits of the PREVIOUS unit (q0/q1, a full unit: only the last unit
    // of the catalogue can be partial, and the last unit of a split is drained after the loop) are formed between the MFMAs --
    // a compare into vcc and an add-with-carry per register (bits = 2 bits + pass, register t on bit 31 - t), ~6 of them in the
    // shadow of every MFMA pair.  With the tests AFTER the unit's MFMAs the two waves of a SIMD, synchronised by the stage
    // barrier, run their matrix phases together and their vector phases together: PMC at 16 384 users -- matrix pipe 28 % busy,
    // vector instructions 37 %, and they did not overlap (203 us).
    // The words of a stage's units are adjacent in the (user, row half) row.  WIDE: they leave as ONE store per lane and stage
    // (16 bytes at d <= 64, 8 at d <= 128) -- the host then makes the splits multiples of the stage, so the stores are aligned
    // (dword-aligned 16-byte stores work, at 35 -> 50 us for the launch).  16 384 users x 50 000 items: 208 -> 181 us; with the
    // row halves padded to 16-byte multiples (Wh % 4 == 0: before, every second half sat on an 8-byte boundary) and the words of
    // the stage in registers instead of a dynamically indexed (= scratch) struct, 168 -> 145 us.  A 2048-user call has too few
    // workgroups to give up four of its 32 splits for that (35.5 -> 39 us): it keeps one 4-byte store per unit.
    constexpr int UPS = ST / kStage;          // units (= words) per stage
    struct __attribute__((packed, aligned(4))) Words { unsigned v[UPS]; };
    unsigned wv[UPS];              // (every index a constant after unrolling: a dynamically indexed copy lives in scratch memory)
#pragma unroll
    for (int i = 0; i < UPS; ++i)
        wv[i] = 0u;
    auto store_words = [&](unsigned *dst) {
        Words t;
#pragma unroll
        for (int i = 0; i < UPS; ++i)
            t.v[i] = wv[i];
        *reinterpret_cast<Words *>(dst) = t;
    };
    // the stage summary (WIDE, a.summ != NULL): the running word of 32 stages, stored when its last stage is noted (the condition
    // is wave-uniform: a scalar branch) and once more at the split's end
    unsigned sm = 0u;
    unsigned *__restrict__ srow = nullptr;
    if constexpr (WIDE) {
        if (a.summ)
            srow = a.summ + (((size_t)user * 2 + h) * a.n_splits + split) * a.summ_sw;
    }
    auto note_stage = [&](int ls, int n_words, bool last) {      // wv[0 .. n_words) are the final words of the split's stage ls
        unsigned any = 0u;
#pragma unroll
        for (int i = 0; i < UPS; ++i)
            any |= i < n_words ? wv[i] : 0u;
        const unsigned bit = any != 0u ? 1u << (ls & 31) : 0u;
        sm = (ls & 31) == 0 ? bit : (sm | bit);
        if (srow && (last || (ls & 31) == 31))
            srow[ls >> 5] = sm;
    };
    auto flush_words = [&](int t) {                        // the split ended on unit t: the words of its unfinished stage
        const int k = ((t - i_beg) >> 6) % UPS;
        if (k != UPS - 1) {
#pragma unroll
            for (int i = 0; i < UPS - 1; ++i)
                if (i <= k)
                    mrow[(t >> 6) - k + i] = wv[i];
        } else {
            store_words(mrow + ((t >> 6) - (UPS - 1)));
        }
    };
    auto unit = [&](auto prev_tag, int t0, int un, f32x16 &c0, f32x16 &c1, const f32x16 &q0, const f32x16 &q1, int t_prev) {
        constexpr bool PREV = decltype(prev_tag)::value;
        constexpr int STEPS = KS + 1;                  // k-steps incl. the bound's
        const unsigned char *pi = smem + buf * SB + (un * kStage + r32) * RB;
        unsigned bits = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            c0[r] = 0.0f, c1[r] = 0.0f;
        // all operand fragments of the unit are requested first (their LDS latency is paid once per unit, under the first tests,
        // not once per k-step), then per k-step: the previous unit's tests of that step, the MFMA pair
        bf16x8 fa0[STEPS], fa1[STEPS];
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int off = s < KS ? 32 * s + 16 * h : 32 * KS;        // the bound's step: both halves read the pad chunk
            fa0[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(pi + off));
            fa1[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(pi + 32 * RB + off));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            if constexpr (PREV) {
#pragma unroll
                for (int t = (32 * s) / STEPS; t < (32 * (s + 1)) / STEPS; ++t) {
                    const float val = t < 16 ? q0[t & 15] : q1[t & 15];
                    asm volatile("v_cmp_nle_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(bits) : "v"(val), "v"(tau) : "vcc");
                }
            }
            const bf16x8 bb = s < KS ? bfr[s < KS ? s : 0] : bfx;
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0[s], bb, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1[s], bb, c1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (PREV) {
            if constexpr (WIDE)
                wv[(un + UPS - 1) % UPS] = user_ok ? bits : 0u;     // (un is a constant of the unrolled stage loop)
            else
                mrow[t_prev >> 6] = user_ok ? bits : 0u;
        }
    };
    // the last unit of the split (possibly the partial last unit of the catalogue: rows past i_end are clamped copies)
    auto drain = [&](const f32x16 &q0, const f32x16 &q1, int t_prev) {
        const int lim = i_end - t_prev;
        unsigned bits = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            bits = (bits << 1) | (((r & 3) + 8 * (r >> 2) + 4 * h < lim && !(q0[r] <= tau)) ? 1u : 0u);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            bits = (bits << 1) | ((32 + (r & 3) + 8 * (r >> 2) + 4 * h < lim && !(q1[r] <= tau)) ? 1u : 0u);
        if constexpr (WIDE) {
            const int kk = ((t_prev - i_beg) >> 6) % UPS;
#pragma unroll
            for (int i = 0; i < UPS; ++i)
                wv[i] = i == kk ? (user_ok ? bits : 0u) : wv[i];
            flush_words(t_prev);
            note_stage((t_prev - i_beg) / ST, kk + 1, true);      // (words past kk are the previous stage's)
        } else {
            mrow[t_prev >> 6] = user_ok ? bits : 0u;
        }
    };
    using Yes = std::true_type;
    using No = std::false_type;
    f32x16 A0, A1, B0, B1;       // units alternate between the two accumulator sets (a stage holds an even number of units)
    static_assert((ST / kStage) % 2 == 0, "units per stage");
    int t_last = i_beg;
    bool last_is_a = true;
    // one stage: the rows of the next one are requested at its start and converted into the other LDS buffer at its end.  All
    // loads and stores of the loop body unconditional (a branch around them would leave the wait-count pass with vmcnt(0)):
    // rows past the split clamp to its last one, a buffer nobody reads takes the copies, and the first stage -- which has no
    // previous stage's words to store -- is its own instance of the body.
    auto stage = [&](auto first_tag, int s0) {
        constexpr bool FIRST = decltype(first_tag)::value;
        dma_stage(buf + kLead >= RING ? buf + kLead - RING : buf + kLead, s0 + kLead * ST);     // ring position (buf + kLead) % RING: multiplied last in the previous stage
#pragma unroll
        for (int un = 0; un < UPS; un += 2) {
            const int t0 = s0 + un * kStage;     // first item of this 64-item unit
            if (un > 0 && t0 >= i_end)
                break;
            if (FIRST && un == 0)
                unit(No{}, t0, un, A0, A1, B0, B1, 0);
            else
                unit(Yes{}, t0, un, A0, A1, B0, B1, t0 - kStage);
            if (WIDE && !FIRST && un == 0) {   // the previous stage's words are complete: one store per lane
                store_words(mrow + (t0 >> 6) - UPS);
                note_stage((s0 - i_beg) / ST - 1, UPS, false);
            }
            t_last = t0, last_is_a = true;
            if (t0 + kStage >= i_end)
                break;
            unit(Yes{}, t0 + kStage, un + 1, B0, B1, A0, A1, t0);
            t_last = t0 + kStage, last_is_a = false;
        }
        // The next stage's pieces have landed once at most RingWait::kAllowed (= NP: the pieces requested at THIS stage's start, two
        // stages ahead) vector-memory operations are in flight -- the invariant at DmaRingWait, which asks nothing of where the
        // stage's word stores sit among the requests (with vmcnt(0) the last store's round trip was paid every stage).  The
        // split's last stage requests nothing real and drains everything before the workgroup's LDS is released.
        if (s0 + ST >= i_end)
            wait_vmcnt<0>();
        else
            RingWait::wait();
#ifdef TGCN_CHECK_DMA
        if (s0 + ST < i_end)
            verify_stage(buf == RING - 1 ? 0 : buf + 1, s0 + ST);
#endif
        __syncthreads();
        buf = buf == RING - 1 ? 0 : buf + 1;
    };
    stage(Yes{}, i_beg);
    for (int s0 = i_beg + ST; s0 < i_end; s0 += ST)
        stage(No{}, s0);
    if (last_is_a)
        drain(A0, A1, t_last);
    else
        drain(B0, B1, t_last);
}

// ---- the bf16 filter for wide rows (128 < d <= 1024, d % 8 == 0: the folded ltr_linear operands, K = 896 / 960) -----------
// Same test, same pack (rows of 32 KS + 16 bytes).  The users' fragments stay in registers for the whole launch; a row is too wide
// for one wave to hold 32 users' fragments AND leave the SIMD a second wave (4 KS registers per lane: 240 of 512 at K = 960 --
// round 2's form, one wave per SIMD, had the matrix pipe 20 % busy: every wait, barrier and log append of the SIMD's only wave was
// exposed).  So K is SPLIT between two waves: a workgroup is 8 waves, waves w and w + 4 (one SIMD) share 32 users, each holds the
// fragments of HALF the k-steps (2 KS registers: 120 at K = 960, two waves per SIMD fit the 256-register budget) and accumulates
// its half of the dot products of the unit's 64 items x 32 users; at the end of a unit each wave hands the partial sums of ONE
// 32-item tile to its partner through LDS (16 registers each way) and finishes the other tile: adds, the bound's k-step (in wave
// 0's sums), tests and appends for 32 items.  One wave's waits, LDS traffic and appends now run under its partner's MFMAs.
// The items pass through LDS in stages of 64 rows x CK k-steps (CK = half the row where LDS allows: two barriers per unit), the
// row's factor chunk rides in the LDS row's 16-byte pad; a stage is requested one stage ahead (a stage is ~2 x 15 x 2 MFMAs of
// 32 cycles on a SIMD: more than a memory round trip).
struct PreWideArgs {
    PreArgs p;                      // (mask / Wh unused here)
    float2 *__restrict__ logs;      // [B][4 S][cap2] lane-private segments of (approx + bound, item), segment = split * 4 + tile * 2 + row half
    int *__restrict__ counts;       // [B][4 S] entries appended (may exceed cap2: the user then takes the exact fallback)
    int S, cap2;
    // SAMPLE form: "item" t is row t * row_stride of the pack, the plain approximate scores go to sample[user * sample_ld + t]
    // (the threshold sample k_tau ranks: as an fp32 GEMM over the strided rows it cost a fifth of the whole filter)
    int row_stride;
    float *__restrict__ sample;
    int64_t sample_ld;
    int n_tiles;                    // user tiles of 128; the grid is linear: 32 x 8 x ceil(n_tiles x splits / 32 / 8) workgroups
};

// workgroups of the wide filter's linear grid: the (split, tile) pairs, tile fastest, in groups of 32 (one XCD's CUs: 32 user tiles of
// one split, or the end of one split and the start of the next), dealt 8 groups per round
inline unsigned wide_grid(int n_tiles, int splits)
{
    const int groups = (n_tiles * splits + 31) / 32;
    return (unsigned)(((groups + 7) / 8) * 8 * 32);
}

// Two constants of the wide loop, each the best of a measured sweep (profiles/r03_experiments.md §3; the ablation builds that
// priced the loop's ingredients were private copies of the round-3 source and are not part of this translation unit):
constexpr int kWideFragLead = 2;    // k-steps the LDS fragment reads run ahead of their MFMAs (2 / 3 / 4 / 6 -> 8.6 k / 9.0 k / 9.7 k / 17 k cycles per unit)
constexpr int kWideStagger = 1;     // s_sleep units (64 cycles) waves 4-7 wait after every stage barrier, so that the two waves of a SIMD do
                                    // not reach their fragment waits and their MFMAs together (0 / 1 / 3 / 8: 8.44 k / 8.18 k / 8.25 k / 8.42 k)

// k-steps per LDS stage: three stage buffers of 64 rows (one multiplied, two in flight) beside the exchange buffer in 160 KB
template <int KS>
struct WideStage {
    static constexpr int CK = KS <= 16 ? 8 : KS <= 32 ? 16 : KS == 56 ? 14 : KS == 60 ? 20 : 16;
};

// s_waitcnt vmcnt(N) alone (gfx9 encoding: vmcnt in bits 3:0 and 15:14, expcnt 6:4 and lgkmcnt 11:8 left at their maxima)
template <int KS, bool SAMPLE>
__global__ __launch_bounds__(512) void k_score_prefilter_wide(const PreWideArgs wa)
{
    const PreArgs &a = wa.p;
    constexpr int UT = 128;
    constexpr int RBG = 32 * KS + 16;          // pack row
    constexpr int CK = WideStage<KS>::CK;      // k-steps per stage
    static_assert(KS % CK == 0 && CK % 2 == 0 && KS / CK >= 2, "stages per row");
    constexpr int NCH = KS / CK;
    constexpr int HK = CK / 2;                 // ... of which a wave multiplies half
    constexpr int KH = KS / 2;                 // k-steps whose user fragments a wave holds
    constexpr int RBL = 32 * CK + 16;          // LDS row: the chunk + 16 bytes (conflict-free ds_read_b128 down a column slice: 8 CK + 4
                                               // dwords = 4 x odd, so 16 rows that differ mod 16 cover the 64 banks); the pad holds the
                                               // row's factor chunk
    constexpr int SBYTES = kStage * RBL;       // a stage = SBYTES contiguous bytes of LDS
    constexpr int NPIECE = (SBYTES + 1023) / 1024;     // ... moved as 1 KB pieces, one per wave instruction (16 bytes per lane)
    constexpr int NPW = (NPIECE + 7) / 8;              // pieces per wave and stage (the same count for every wave: waits are counted)
    constexpr int SBUF = NPIECE * 1024;
    static_assert(3 * SBUF + 4 * 2 * 16 * kWave * 4 <= 160 * 1024, "LDS");
    // Stages travel by LDS-DMA (global_load_lds_dwordx4: 16 bytes per lane straight into LDS, no staging registers -- the users'
    // fragments need them) into a ring of three buffers: stage s + 2 is requested when stage s starts, so a request has two
    // stages (~2 x 2 x HK x 2 MFMAs of 32 cycles) to land.  With one stage of lead (round-3 first form, through registers) the
    // waves spent 58 % of their time in s_waitcnt / s_barrier: the pack (116 MB at config 5) comes from the Infinity Cache, a
    // loaded round trip is longer than a stage.
    __shared__ __attribute__((aligned(16))) unsigned char smem[3][SBUF];
    __shared__ __attribute__((aligned(16))) float xbuf[4][2][16 * kWave];   // [user group][sending half][register][lane]
    const int lane = lane_id();
    const int w = uniform(threadIdx.x >> 6);
    const int ug = w & 3, hk = w >> 2;                 // waves ug and ug + 4 land on one SIMD (dispatch order 0 -> 2 -> 1 -> 3)
    const int r32 = lane & 31;
    const int h = lane >> 5;
    // Which (user tile, item split) this workgroup takes.  A workgroup streams its split's slice of the pack (7 MB of config 5's
    // 116 MB) once; every user tile of the call streams all of it, so at 8192 users the launch pulls 64 x 116 MB = 7.4 GB -- at
    // ~1 ms per launch that is the Infinity Cache's whole bandwidth.  The grid is therefore linear and dealt so that the 32
    // workgroups an XCD runs at a time (one per CU: 160 KB of LDS) are 32 user tiles of ONE split: started together, working at
    // the same pace, they read the same rows within microseconds of each other and all but the first find them in that XCD's L2.
    // (Workgroups are dealt round-robin over the XCDs: linear id % 8 -- for speed only, no result depends on the placement.)
    // The pairs are numbered split-major and cut into groups of 32 WITHOUT padding a split's last group: 48 tiles x 5 splits are
    // 7.5 groups, one generation on every XCD (padded per split they were 10 groups: two XCDs ran two generations, 50.8 ms for
    // config 5 in calls of 6144 users against 33 ms at 8192).
    int tile, split;
    {
        const int L = blockIdx.x, x = L & 7, j = L >> 3;
        const int grp = (j >> 5) * 8 + x;              // group of 32 workgroups: XCD x, generation j / 32
        const int q = grp * 32 + (j & 31);
        split = q / wa.n_tiles;
        tile = q - split * wa.n_tiles;
    }
    const int u0 = tile * UT;
    const int i_beg = split * a.items_per_split;
    const int i_end = min(a.I, i_beg + a.items_per_split);
    if (tile >= wa.n_tiles || i_beg >= i_end)
        return;
    const int user = u0 + ug * 32 + r32;
    const bool user_ok = user < a.B;

    // stage (unit t0, chunk ch) -> ring buffer `buf`: LDS byte o of the stage is byte o % RBL of row o / RBL; its source is the
    // chunk's bytes of the pack row, or (the pad) the row's factor chunk.  Every request unconditional: rows past the table clamp
    // to its last row, a piece past the stage's last one repeats it (the same bytes to the same place).  What does not depend on
    // the stage is computed once per lane and piece (row, column part): a request is an add, a min, a 64-bit multiply-add.
    int prow[NPW], pcol[NPW];       // row inside the stage; byte inside the pack row for chunk 0 (the pad: the factor chunk, bit 30 set)
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int pc = min(w + 8 * i, NPIECE - 1);                 // wave-uniform
        const int o = min(pc * 1024 + lane * 16, SBYTES - 16);
        prow[i] = o / RBL;
        const int q = o - prow[i] * RBL;
        pcol[i] = q < 32 * CK ? q : (32 * KS) | (1 << 30);
    }
    const int last_row = (int)(a.pack_bytes / RBG) - 1;            // rows of the pack
    const unsigned row_pitch = (unsigned)wa.row_stride * RBG;
    const int last_t = last_row / wa.row_stride;
    constexpr int kLead = 2;      // stages between a request and the stage that multiplies it (three ring buffers)
    using RingWait = DmaRingWait<NPW, kLead>;
    auto request_piece = [&](int i, int buf, int t0, int ch) {
        const int pc = min(w + 8 * i, NPIECE - 1);
        const unsigned col = (pcol[i] >> 30) ? (unsigned)(pcol[i] & 0xFFFFFF) : (unsigned)(pcol[i] + ch * (32 * CK));
        const size_t off = (size_t)(unsigned)min(t0 + prow[i], last_t) * row_pitch + col;
#ifdef TGCN_CHECK_DMA
        if (w + 8 * i < NPIECE) {      // (a clamped duplicate of the last piece is another wave's to poison and check)
            dma_poison((unsigned)(uintptr_t)(smem[buf] + pc * 1024) + (unsigned)lane * 16u);
            __builtin_amdgcn_s_waitcnt(0xC07F);
        }
#endif
        lds_dma16(a.ipack + off, uniform((int)(unsigned)(uintptr_t)(smem[buf] + pc * 1024)));
    };
#ifdef TGCN_CHECK_DMA
    auto verify_stage = [&](int buf, int t0, int ch) {
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            if (w + 8 * i >= NPIECE)
                continue;
            const unsigned col = (pcol[i] >> 30) ? (unsigned)(pcol[i] & 0xFFFFFF) : (unsigned)(pcol[i] + ch * (32 * CK));
            const size_t off = (size_t)(unsigned)min(t0 + prow[i], last_t) * row_pitch + col;
            dma_verify(a.ipack + off, (unsigned)(uintptr_t)(smem[buf] + (w + 8 * i) * 1024) + (unsigned)lane * 16u);
        }
    };
#endif
    auto request = [&](int buf, int t0, int ch) {
#pragma unroll
        for (int i = 0; i < NPW; ++i)
            request_piece(i, buf, t0, ch);
    };
    // stage n of the launch (n = unit * NCH + chunk) lives in buffer n % 3
    request(0, i_beg, 0);
    request(1, NCH > 1 ? i_beg : i_beg + kStage, NCH > 1 ? 1 : 0);

    // the users' fragments: k-step (ch, s) of this wave is k-step ch CK + hk HK + s of the row; lane (r32, h) holds elements
    // 16 kg + 8 h .. + 7 of user r32's row, straight from global memory
    bf16x8 bfr[KH];
    {
        const float *__restrict__ urow = a.U + (size_t)(a.user_ids ? a.user_ids[min(user, a.B - 1)] : (int64_t)min(user, a.B - 1)) * a.d;
#pragma unroll
        for (int j = 0; j < KH; ++j) {
            const int kg = (j / HK) * CK + hk * HK + (j % HK);
            const int k0 = 16 * kg + 8 * h;
            const float4 x = *reinterpret_cast<const float4 *>(urow + min(k0, a.d - 8));
            const float4 y = *reinterpret_cast<const float4 *>(urow + min(k0, a.d - 8) + 4);
            const bool in = k0 < a.d && user_ok;
            bfr[j] = __builtin_bit_cast(bf16x8, make_uint4(in ? pack_bf16(x.x, x.y) : 0u, in ? pack_bf16(x.z, x.w) : 0u,
                                                          in ? pack_bf16(y.x, y.y) : 0u, in ? pack_bf16(y.z, y.w) : 0u));
            if (j % 8 == 7)
                __builtin_amdgcn_sched_barrier(0);     // (eight steps' loads in flight, not all of them: they would not fit)
        }
    }
    const float tau = user_ok && !SAMPLE ? a.tau[(size_t)user * a.tau_stride] : INFINITY;
    const float2 ub = user_ok && !SAMPLE ? *reinterpret_cast<const float2 *>(a.ubound + 2 * (size_t)user) : make_float2(0.0f, 0.0f);
    const bf16x8 bfx = __builtin_bit_cast(bf16x8, h == 0 ? make_uint4(bf16_up_bits(ub.x) | (bf16_up_bits(ub.y) << 16),
                                                                       bf16_up_bits(ub.x * kAccumBudget), 0u, 0u)
                                                         : make_uint4(0u, 0u, 0u, 0u));
    // the lane's own segment of the user's candidate log: every pair that passes is LOGGED with its raised approximate score
    // (~0.5 per lane and unit against 2 KS + 2 MFMAs per wave pair): k_refine turns them into a second, far tighter threshold
    // before any fp32 chain runs.  This wave finishes tile hk of every unit: segment (split, tile hk, row half h).
    const size_t seg = SAMPLE ? 0 : ((size_t)min(user, a.B - 1) * (4 * wa.S) + (size_t)split * 4 + hk * 2 + h);
    float2 *__restrict__ lg = wa.logs + seg * wa.cap2;
    int n_log = 0;
    asm volatile("" ::"v"(tau), "v"(bfx));
    wait_vmcnt<0>();                // everything above has arrived, the first two stages included
    __syncthreads();
    int buf = 0;                    // ring position of the stage being multiplied
    float *__restrict__ xout = &xbuf[ug][hk][0] + lane * 4;           // [register quad][lane][4]
    const float *__restrict__ xin = &xbuf[ug][hk ^ 1][0] + lane * 4;
    // A finished tile (16 sums per lane) is consumed UNDER the next unit's first stage: its tests / appends (or the sample's
    // stores) sit between that stage's MFMAs, as the stage-after-next's requests do -- behind an MFMA they issue while the
    // matrix pipe works, in front of the stage they would leave it idle for both waves of the SIMD at once.
    float fin[16];
    int fin_t = -1;                 // first item of the tile in `fin` (-1: none yet)
    constexpr int PF = kWideFragLead < HK ? kWideFragLead : HK;      // k-steps of lead of the LDS fragment reads
    constexpr int TS = HK > NPW ? HK - NPW : 0;      // k-steps of a unit's first stage that carry the tests (0: after the stage)
    auto consume = [&](int r0, int r1) {       // registers r0 .. r1 - 1 of the finished tile
        if constexpr (SAMPLE) {
            // (registers 4 g .. 4 g + 3 of a tile are four consecutive items: one 16-byte store per group, issued with its last register)
#pragma unroll
            for (int r = r0; r < r1; ++r) {
                if ((r & 3) != 3)
                    continue;
                const int g = r >> 2;
                const int item = fin_t + 8 * g + 4 * h;
                if (user_ok) {
                    float *__restrict__ srow = wa.sample + (size_t)user * wa.sample_ld;
                    if (item + 3 < i_end) {
                        *reinterpret_cast<float4 *>(srow + item) = make_float4(fin[4 * g], fin[4 * g + 1], fin[4 * g + 2], fin[4 * g + 3]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (item + e < i_end)
                                srow[item + e] = fin[4 * g + e];
                    }
                }
            }
        } else {
            const int lim = user_ok ? i_end - fin_t : 0;
#pragma unroll
            for (int r = r0; r < r1; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                const float v = fin[r];
                if (row < lim && !(v <= tau)) {
                    if (n_log < wa.cap2)
                        lg[n_log] = make_float2(v, __int_as_float(fin_t + row));
                    ++n_log;
                }
            }
        }
    };
    for (int t0 = i_beg; t0 < i_end; t0 += kStage) {
        f32x16 c0, c1;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            if (hk)
                __builtin_amdgcn_s_sleep(kWideStagger);      // (64 cycles each) take the pair out of lockstep
            const unsigned char *pi = smem[buf] + r32 * RBL + hk * (HK * 32) + 16 * h;
            // the stage after the next goes into the buffer the previous stage was read from (every wave is past that stage's
            // barrier); past the split: copies nobody reads
            const int c2 = (ch + kLead) % NCH, t2 = t0 + ((ch + kLead) / NCH) * kStage, b2 = buf == 0 ? 2 : buf - 1;
            // fragments PF k-steps ahead of their MFMAs, and no further (scheduling barriers): left to itself the compiler hoists
            // the whole chunk's LDS reads and spills the users' fragments.  (Stamps at two k-steps: ~2.5 k of a unit's 8.5 k cycles
            // per wave in s_waitcnt lgkmcnt -- eight waves' reads and the ring's DMA writes queue at the LDS.)
            bf16x8 f0[HK], f1[HK];
#pragma unroll
            for (int s = 0; s < PF && s < HK; ++s) {
                f0[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(pi + 32 * s));
                f1[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(pi + 32 * RBL + 32 * s));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < HK; ++s) {
                if (s + PF < HK) {
                    f0[s + PF] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(pi + 32 * (s + PF)));
                    f1[s + PF] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(pi + 32 * RBL + 32 * (s + PF)));
                }
                if (ch == 0 && s == 0) {       // a unit's first k-step starts its sums from zero (no register clearing)
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0[s], bfr[ch * HK + s], zero, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1[s], bfr[ch * HK + s], zero, 0, 0, 0);
                } else {
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0[s], bfr[ch * HK + s], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1[s], bfr[ch * HK + s], c1, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                // behind the pair: (first stage of a unit) a share of the previous tile's tests in the stage's first k-steps, one
                // request of stage + 2 in its last ones -- in this order, so that every log store is OLDER than every request of
                // the stage and the counted wait below never has to cover a request just issued
                if (ch == 0 && TS > 0 && s < TS && fin_t >= 0)
                    consume((16 * s) / TS, (16 * (s + 1)) / TS);
                if (s >= HK - min(NPW, HK))
                    request_piece(s - (HK - min(NPW, HK)), b2, t2, c2);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = min(NPW, HK); i < NPW; ++i)      // (stages with fewer k-steps per wave than pieces)
                request_piece(i, b2, t2, c2);
            if (ch == 0 && TS == 0 && fin_t >= 0)
                consume(0, 16);
            if (ch == NCH - 1) {
                if (!SAMPLE && hk == 0) {      // the bound's k-step, once per pair (both row halves read the row's factor chunk in the pad)
                    const unsigned char *pf = smem[buf] + r32 * RBL + 32 * CK;
                    const bf16x8 g0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(pf));
                    const bf16x8 g1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(pf + 32 * RBL));
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g0, bfx, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1, bfx, c1, 0, 0, 0);
                }
                // hand the partner the partial sums of the tile IT finishes (wave 0: tile 1, wave 1: tile 0)
                if (hk == 0) {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<float4 *>(xout + g * 4 * kWave) = make_float4(c1[4 * g], c1[4 * g + 1], c1[4 * g + 2], c1[4 * g + 3]);
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<float4 *>(xout + g * 4 * kWave) = make_float4(c0[4 * g], c0[4 * g + 1], c0[4 * g + 2], c0[4 * g + 3]);
                }
            }
            // the NEXT stage's pieces of this wave have landed once at most RingWait::kAllowed (= NPW: the requests of stage + 2 just
            // issued) operations are outstanding -- the invariant at DmaRingWait: it does not depend on where the log stores sit
            // among the requests (with TS == 0 they are issued AFTER them) nor on how stores retire against loads; the barrier then
            // covers the other waves' pieces and their reads of `buf`
            RingWait::wait();
#ifdef TGCN_CHECK_DMA
            verify_stage(buf == 2 ? 0 : buf + 1, t0 + ((ch + 1) / NCH) * kStage, (ch + 1) % NCH);
#endif
            __syncthreads();
            buf = buf == 2 ? 0 : buf + 1;
        }
        // this wave's tile: own half + the partner's (the next write of xbuf lies behind the next unit's first barrier)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 p = *reinterpret_cast<const float4 *>(xin + g * 4 * kWave);
            fin[4 * g + 0] = (hk == 0 ? c0[4 * g + 0] : c1[4 * g + 0]) + p.x;
            fin[4 * g + 1] = (hk == 0 ? c0[4 * g + 1] : c1[4 * g + 1]) + p.y;
            fin[4 * g + 2] = (hk == 0 ? c0[4 * g + 2] : c1[4 * g + 2]) + p.z;
            fin[4 * g + 3] = (hk == 0 ? c0[4 * g + 3] : c1[4 * g + 3]) + p.w;
        }
        fin_t = t0 + 32 * hk;          // first item of the tile
    }
    consume(0, 16);                    // the split's last tile
    wait_vmcnt<0>();          // requests still in flight write into this workgroup's LDS: they must land before it is released
    if (user_ok && !SAMPLE)
        wa.counts[seg] = n_log;
}

// ---- the threshold sample on the bf16 pipe ----------------------------------------------------------------------------------
// S[b][j] ~ <U[b], It[j * stride]> for the m sampled items, operands rounded to bf16, fp32 accumulation: k_tau only RANKS these
// scores (tau is a bar for the candidate search, never a result: a tau that comes out too high sends the user to the exact
// fallback, one that comes out low costs candidates), so the sample does not need the fp32 chain.  Users on MFMA rows here
// (A operand), items on columns: accumulator register t of lane l holds (user (t & 3) + 8 (t >> 2) + 4 (l >> 5), item l & 31),
// so one store instruction writes 128 contiguous bytes of two user rows.  One workgroup = 256 users x 128 sampled items, no
// loop: 16 384 users x 1563 samples took 61 us as an fp32 GEMM (k_score_dense on every 32nd item row).
//
// TOP form (round 3): the scores never leave the kernel.  k_tau needs the r-th largest of a user's m sampled scores; a workgroup
// holds 128 of them per user, so it writes only each (user, block)'s TWO largest -- S[b][2 blk], S[b][2 blk + 1] -- and k_tau ranks
// 2 ceil(m / 128) values instead of m (16 384 users x 3125 samples: 205 MB written and read back -> 3.3 MB; sample + k_tau
// 46 + 61 us -> see DESIGN.md 4.2b).  A block that holds three or more of the user's r largest makes the bar one rank lower than
// the exact one: more candidates, never fewer.  The sampled TRAIN items must not count: k_sample_bits turns every user's train
// list into a bitmap over the sample first (bit j = sampled item j is a train item; 4 words per (user, block)), and the masked
// scores enter the top-2 as -inf, as do the columns past the sample's end.
struct SampleArgs {
    const float *__restrict__ U;
    const int64_t *__restrict__ user_ids;
    const float *__restrict__ It;
    float *__restrict__ S;
    int64_t ld;
    int B, m, d, stride;
    const unsigned *__restrict__ bits;   // TOP: [B][bits_words] train-item bitmap over the sample, or NULL (no mask)
    int bits_words;
};

constexpr int kSampleBitsMaxWords = 4096;     // LDS row of k_sample_bits (4 waves x 16 KB = the 64 KB a launch may ask for): 1024 blocks = 131 072 sampled items

__global__ __launch_bounds__(256) void k_sample_bits(const int *__restrict__ mask_rowptr, const int *__restrict__ mask_items,
                                                     unsigned *__restrict__ bits, int B, int words, int stride, int m)
{
    extern __shared__ unsigned brow[];            // [4][words]: the wave's row is built here (ds_or), then stored once
    const int lane = lane_id();
    const int w = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + w;
    if (b >= B)
        return;
    unsigned *row = brow + w * words;
    const int mb = mask_rowptr[b], me = mask_rowptr[b + 1];
    for (int j = lane; j < words; j += kWave)
        row[j] = 0u;
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    for (int e = mb + lane; e < me; e += kWave) {
        const int it = mask_items[e];
        const int j = it / stride;
        if (it % stride == 0 && j < m)
            atomicOr(&row[j >> 5], 1u << (j & 31));
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    unsigned *__restrict__ out = bits + (size_t)b * words;
    for (int j = lane; j < words; j += kWave)
        out[j] = row[j];
}

// the two largest of a value pair list across the 32 lanes of a row half: DPP inside the rows of 16 (quad_perm, quad_perm,
// row_half_mirror, row_mirror: every step pairs disjoint groups), one ds_bpermute step between the two rows
template <int CTRL>
__device__ __forceinline__ void top2_dpp_step(float &a0, float &a1)
{
    const int x0 = __float_as_int(a0), x1 = __float_as_int(a1);
    const float p0 = __int_as_float(__builtin_amdgcn_update_dpp(x0, x0, CTRL, 0xf, 0xf, false));
    const float p1 = __int_as_float(__builtin_amdgcn_update_dpp(x1, x1, CTRL, 0xf, 0xf, false));
    a1 = fmaxf(fminf(a0, p0), fmaxf(a1, p1));
    a0 = fmaxf(a0, p0);
}

template <int KS, bool FULLK, bool TOP>
__global__ __launch_bounds__(512) void k_sample_bf16(const SampleArgs a)
{
    constexpr int T = 512, UT = 256, SI = 128;     // threads, users and sampled items per workgroup
    constexpr int DQ = 4 * KS, RB = 32 * KS + 16;
    constexpr int NU = UT * DQ / T, NI = SI * DQ / T;
    __shared__ __attribute__((aligned(16))) unsigned char smem[(UT + SI) * RB];
    __shared__ __attribute__((aligned(16))) unsigned mwords[TOP ? UT * 4 : 4];     // TOP: the four bitmap words of (user, this block)
    const int lane = lane_id();
    const int w = threadIdx.x >> 6;
    const int r32 = lane & 31;
    const int h = lane >> 5;
    const int u0 = blockIdx.x * UT;
    const int j0 = blockIdx.y * SI;
    if constexpr (TOP) {
        if (threadIdx.x < UT) {
            const int user = min(u0 + (int)threadIdx.x, a.B - 1);
            const u32x4 z = {0u, 0u, 0u, 0u};
            *reinterpret_cast<u32x4 *>(mwords + 4 * threadIdx.x) =
                a.bits ? *reinterpret_cast<const u32x4 *>(a.bits + (size_t)user * a.bits_words + 4 * blockIdx.y) : z;
        }
    }
    float4 vu[NU], vi[NI];
    size_t ru[NU], ri[NI];
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const int r = min(u0 + (i * T + (int)threadIdx.x) / DQ, a.B - 1);
        ru[i] = a.user_ids ? (size_t)a.user_ids[r] : (size_t)r;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i)
        ri[i] = (size_t)min(j0 + (i * T + (int)threadIdx.x) / DQ, a.m - 1) * a.stride;
    auto fetch = [&](const float *__restrict__ p, int k) {
        if constexpr (FULLK)
            return *reinterpret_cast<const float4 *>(p + k);
        else
            return make_float4(k + 0 < a.d ? p[k + 0] : 0.0f, k + 1 < a.d ? p[k + 1] : 0.0f, k + 2 < a.d ? p[k + 2] : 0.0f,
                               k + 3 < a.d ? p[k + 3] : 0.0f);
    };
#pragma unroll
    for (int i = 0; i < NI; ++i)
        vi[i] = fetch(a.It + ri[i] * a.d, ((i * T + (int)threadIdx.x) % DQ) * 4);
#pragma unroll
    for (int i = 0; i < NU; ++i)
        vu[i] = fetch(a.U + ru[i] * a.d, ((i * T + (int)threadIdx.x) % DQ) * 4);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int f = i * T + threadIdx.x;
        *reinterpret_cast<uint2 *>(smem + (UT + f / DQ) * RB + (f % DQ) * 8) = make_uint2(pack_bf16(vi[i].x, vi[i].y), pack_bf16(vi[i].z, vi[i].w));
    }
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const int f = i * T + threadIdx.x;
        *reinterpret_cast<uint2 *>(smem + (f / DQ) * RB + (f % DQ) * 8) = make_uint2(pack_bf16(vu[i].x, vu[i].y), pack_bf16(vu[i].z, vu[i].w));
    }
    __syncthreads();
    bf16x8 fu[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s)
        fu[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(smem + (w * 32 + r32) * RB + 32 * s + 16 * h));
    float t0[TOP ? 16 : 1], t1[TOP ? 16 : 1];
    if constexpr (TOP) {
#pragma unroll
        for (int t = 0; t < 16; ++t)
            t0[t] = -INFINITY, t1[t] = -INFINITY;
    }
#pragma unroll
    for (int un = 0; un < SI / 64; ++un) {
        f32x16 c0, c1;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            c0[r] = 0.0f, c1[r] = 0.0f;
        const unsigned char *pi = smem + (UT + un * 64 + r32) * RB;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bf16x8 f0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(pi + 32 * s + 16 * h));
            const bf16x8 f1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(pi + 32 * RB + 32 * s + 16 * h));
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fu[s], f0, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fu[s], f1, c1, 0, 0, 0);
        }
        const int j = j0 + un * 64 + r32;
        if constexpr (TOP) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const uint2 mw = *reinterpret_cast<const uint2 *>(mwords + 4 * (w * 32 + (t & 3) + 8 * (t >> 2) + 4 * h) + 2 * un);
                const unsigned w0 = mw.x, w1 = mw.y;
                const float x0 = (j < a.m && !((w0 >> r32) & 1u)) ? c0[t] : -INFINITY;
                const float x1 = (j + 32 < a.m && !((w1 >> r32) & 1u)) ? c1[t] : -INFINITY;
                const float hi = fmaxf(x0, x1), lo = fminf(x0, x1);      // (a NaN score drops out: the bar is built from the rest)
                t1[t] = fmaxf(fminf(t0[t], hi), fmaxf(t1[t], lo));
                t0[t] = fmaxf(t0[t], hi);
            }
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int user = u0 + w * 32 + (t & 3) + 8 * (t >> 2) + 4 * h;
                if (user < a.B) {
                    float *__restrict__ row = a.S + (size_t)user * a.ld;
                    if (j < a.m)
                        row[j] = c0[t];
                    if (j + 32 < a.m)
                        row[j + 32] = c1[t];
                }
            }
        }
    }
    if constexpr (TOP) {      // the two largest over the 32 lanes of the row half (the 128 columns of user t), then one 8-byte store
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            top2_dpp_step<0xB1>(t0[t], t1[t]);      // quad_perm [1,0,3,2]
            top2_dpp_step<0x4E>(t0[t], t1[t]);      // quad_perm [2,3,0,1]
            top2_dpp_step<0x141>(t0[t], t1[t]);     // row_half_mirror
            top2_dpp_step<0x140>(t0[t], t1[t]);     // row_mirror: every lane holds the two largest of its row of 16
            {
                const float p0 = __shfl_xor(t0[t], 16), p1 = __shfl_xor(t1[t], 16);
                t1[t] = fmaxf(fminf(t0[t], p0), fmaxf(t1[t], p1));
                t0[t] = fmaxf(t0[t], p0);
            }
            const int user = u0 + w * 32 + (t & 3) + 8 * (t >> 2) + 4 * h;
            if (r32 == 0 && user < a.B)
                *reinterpret_cast<float2 *>(a.S + (size_t)user * a.ld + 2 * blockIdx.y) = make_float2(t0[t], t1[t]);
        }
    }
}

// ---- the TOP sample from the item PACK (the prefiltered entry point: the bf16 rows exist already) ---------------------------
// Same result layout as k_sample_bf16<.., TOP> -- S[b][2 blk .. 2 blk + 1] = the two largest unmasked scores of user b among the
// sampled items of block blk -- in the filter kernel's shape instead: 8 waves x 32 users keep their fragments in registers for the
// whole launch, the sampled rows come straight from the pack (row j * stride: 16-byte pieces, no conversion) through two LDS
// stages, items on MFMA rows and the wave's users on columns, so a lane's 64 results of a block belong to ONE user and its two
// largest are a v_med3 / v_max chain in the lane -- one shuffle joins the two row halves.  (k_sample_bf16's users-on-rows form
// needs a five-step cross-lane merge per accumulator register: 40 us per 16 384 x 3125 x 64 sample against 6.5 GFLOP of MFMA.)
struct SamplePackArgs {
    const float *__restrict__ U;
    const int64_t *__restrict__ user_ids;
    const unsigned char *__restrict__ ipack;   // [I][RB] (k_item_pack)
    float *__restrict__ S;
    int64_t ld;
    int B, m, d, stride;
    const unsigned *__restrict__ bits;         // [B][bits_words] train-item bitmap over the sample (k_sample_bits), or NULL
    int bits_words;
    int bpw;                                   // blocks of 128 sampled items per workgroup
};

template <int KS, bool FULLK>
__global__ __launch_bounds__(512) void k_sample_pack_top(const SamplePackArgs a)
{
    constexpr int T = 512, UT = 256, SI = 128;
    constexpr int DQ = 4 * KS, RB = 32 * KS + 16;
    constexpr int PPR = RB / 16;                        // 16-byte pieces of a packed row
    constexpr int NP = (SI * PPR + T - 1) / T;          // ... of a block, per thread
    constexpr int SB = SI * RB;                         // bytes of a stage
    constexpr int NU = UT * DQ / T;
    static_assert(UT * RB <= 2 * SB, "the user tile passes through the two stage buffers");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * SB];
    const int lane = lane_id();
    const int w = threadIdx.x >> 6;
    const int r32 = lane & 31;
    const int h = lane >> 5;
    const int u0 = blockIdx.x * UT;
    const int n_blk = (a.m + SI - 1) / SI;
    const int blk0 = blockIdx.y * a.bpw, blk_end = min(blk0 + a.bpw, n_blk);
    if (blk0 >= blk_end)
        return;
    const int user = u0 + w * 32 + r32;                 // the lane's user (both row halves)
    const unsigned *__restrict__ brow = a.bits ? a.bits + (size_t)min(user, a.B - 1) * a.bits_words : nullptr;

    u32x4 nxt[NP];
    u32x4 mnx = {0u, 0u, 0u, 0u};
    auto request = [&](int blk) {      // every load unconditional: pieces past the block clamp to its last one and are not stored
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int idx = min(i * T + (int)threadIdx.x, SI * PPR - 1);
            const int row = idx / PPR, pc = idx % PPR;
            const size_t item = (size_t)min(blk * SI + row, a.m - 1) * a.stride;
            nxt[i] = *reinterpret_cast<const u32x4 *>(a.ipack + item * RB + pc * 16);
        }
        if (brow)
            mnx = *reinterpret_cast<const u32x4 *>(brow + 4 * blk);
    };
    auto publish = [&](unsigned char *dst) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int idx = i * T + threadIdx.x;
            if (idx < SI * PPR)
                *reinterpret_cast<u32x4 *>(dst + (idx / PPR) * RB + (idx % PPR) * 16) = nxt[i];
        }
    };
    {   // the user tile: fp32 rows -> bf16 rows in LDS (both stage buffers), requested together with the first block
        float4 vu[NU];
        size_t ru[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int r = min(u0 + (i * T + (int)threadIdx.x) / DQ, a.B - 1);
            ru[i] = a.user_ids ? (size_t)a.user_ids[r] : (size_t)r;
        }
        request(blk0);
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int k = ((i * T + (int)threadIdx.x) % DQ) * 4;
            const float *__restrict__ p = a.U + ru[i] * a.d;
            if constexpr (FULLK)
                vu[i] = *reinterpret_cast<const float4 *>(p + k);
            else
                vu[i] = make_float4(k + 0 < a.d ? p[k + 0] : 0.0f, k + 1 < a.d ? p[k + 1] : 0.0f, k + 2 < a.d ? p[k + 2] : 0.0f,
                                    k + 3 < a.d ? p[k + 3] : 0.0f);
        }
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int f = i * T + threadIdx.x;
            *reinterpret_cast<uint2 *>(smem + (f / DQ) * RB + (f % DQ) * 8) = make_uint2(pack_bf16(vu[i].x, vu[i].y), pack_bf16(vu[i].z, vu[i].w));
        }
    }
    __syncthreads();
    bf16x8 bfr[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s)
        bfr[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(smem + (w * 32 + r32) * RB + 32 * s + 16 * h));
    __syncthreads();
    publish(smem);
    u32x4 mcur = mnx;
    __syncthreads();
    int buf = 0;
    for (int blk = blk0; blk < blk_end; ++blk) {
        request(min(blk + 1, blk_end - 1));       // travels under this block's MFMAs (the last iteration's copy is never published)
        // columns past the sample's end count as masked
        const int nv = a.m - blk * SI;
        u32x4 mw = mcur;
        if (nv < SI) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned inv = nv >= 32 * (q + 1) ? 0u : nv <= 32 * q ? 0xFFFFFFFFu : 0xFFFFFFFFu << (nv - 32 * q);
                mw[q] |= inv;
            }
        }
        const bool any_masked = __any((mw.x | mw.y | mw.z | mw.w) != 0u);
        float t0 = -INFINITY, t1 = -INFINITY;      // the lane's two largest of this block (its user, its row half)
#pragma unroll
        for (int un = 0; un < SI / 64; ++un) {
            f32x16 c0, c1;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                c0[r] = 0.0f, c1[r] = 0.0f;
            const unsigned char *pi = smem + buf * SB + (un * 64 + r32) * RB;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 f0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(pi + 32 * s + 16 * h));
                const bf16x8 f1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(pi + 32 * RB + 32 * s + 16 * h));
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, bfr[s], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, bfr[s], c1, 0, 0, 0);
            }
            // register t: item row (t & 3) + 8 (t >> 2) + 4 h of the unit's first (c0) / second (c1) 32 rows
            if (any_masked) {
                const unsigned w0 = (un == 0 ? mw.x : mw.z) >> (4 * h), w1 = (un == 0 ? mw.y : mw.w) >> (4 * h);
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int bit = (t & 3) + 8 * (t >> 2);
                    const float x0 = ((w0 >> bit) & 1u) ? -INFINITY : c0[t];
                    const float x1 = ((w1 >> bit) & 1u) ? -INFINITY : c1[t];
                    t1 = __builtin_amdgcn_fmed3f(t0, t1, x0);      // (a NaN score leaves both unchanged: v_med3 / v_max return the others)
                    t0 = fmaxf(t0, x0);
                    t1 = __builtin_amdgcn_fmed3f(t0, t1, x1);
                    t0 = fmaxf(t0, x1);
                }
            } else {
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    t1 = __builtin_amdgcn_fmed3f(t0, t1, c0[t]);
                    t0 = fmaxf(t0, c0[t]);
                    t1 = __builtin_amdgcn_fmed3f(t0, t1, c1[t]);
                    t0 = fmaxf(t0, c1[t]);
                }
            }
        }
        {   // the other row half of the same user
            const float p0 = __shfl_xor(t0, 32), p1 = __shfl_xor(t1, 32);
            t1 = fmaxf(fminf(t0, p0), fmaxf(t1, p1));
            t0 = fmaxf(t0, p0);
        }
        if (h == 0 && user < a.B)
            *reinterpret_cast<float2 *>(a.S + (size_t)user * a.ld + 2 * blk) = make_float2(t0, t1);
        publish(smem + (buf ^ 1) * SB);      // (read last in the previous iteration: every wave has passed its barrier since)
        mcur = mnx;
        __syncthreads();
        buf ^= 1;
    }
}

// ---- candidates from the pass bits, fp32 scores, compact lists -----------------------------------------------------------
// One WAVE per user, no workgroup-level synchronisation.  (1) the user's 2 Wh mask words are read 32 per lane at a time (all
// loads of a chunk in flight together), counted, scanned once per chunk, and the set bits become item ids in LDS (more than
// kUserCap: the user is handed to the exact fallback); (2) 64 entries at a time: the item rows are read in k-blocks of 32
// floats with coalesced 16-byte loads (8 lanes per row; the next step's loads are in flight while this one is chained),
// transposed through a padded LDS tile, and lane l continues candidate l's chain  s = fmaf(u_k, y_k, s), k ascending -- the
// chain of the MFMA 32x32x2 f32 path, of the dense path and of k_brute_part; (3) entries with s > tau are appended to the
// user's flat (score, item) list; its length goes to totals[b] (the order of the list is irrelevant to k_select).  This is
// where the superset shrinks to {score > tau}.  Measured alternatives (config 2, 2048 users): a workgroup per user with the
// list shared by eight waves 55 us, a wave per (user, eighth of the item table) placed on its own XCD 60 us -- both bound by
// their chain of dependent round trips (mask, mask again, rows, atomic), not by bytes: without the row loads 50 us.
struct RescoreArgs {
    const float *__restrict__ U;
    const int64_t *__restrict__ user_ids;
    const float *__restrict__ It;
    const float *__restrict__ tau;
    int tau_stride;
    const unsigned *__restrict__ mask;
    int Wh, n_units;                 // words per row half (a multiple of 4), of which the first n_units = ceil(I / 64) are written
    const int *__restrict__ surv;    // FROM_LIST: [B][surv_cap] candidate ids of k_refine, surv_n[b] of them (instead of the mask)
    const int *__restrict__ surv_n;
    int surv_cap;
    float2 *__restrict__ lists;     // [B][list_cap]  (SELECT: unused -- the kept pairs never leave the kernel)
    int *__restrict__ totals;       // [B]
    int B, d, list_cap;
    // SELECT (narrow rows): the exact selection of the user's top k runs in the same wave, on the kept pairs in LDS
    const int *__restrict__ mask_rowptr;   // may be NULL
    const int *__restrict__ mask_items;
    float *__restrict__ out_val;
    int64_t *__restrict__ out_idx;
    int *__restrict__ flagged;             // [1 + B]: count, then the users left to the exact fallback
    int k, do_round;
    PassSummary summ;                      // the filter's stage summary (words == NULL: none, every mask word is scanned)
};

constexpr int kUserCap = 1536;      // candidates of a user held in LDS
constexpr int kOverflow = 1 << 20;  // a total beyond any list: k_select hands the user to the fallback
constexpr int kMaxPreD = 1024;      // widest row of the prefiltered path
constexpr int kKB = 32;             // floats of a row per LDS tile
constexpr int kTileRow = kKB + 1;   // padded: lane = row reads are conflict-free
constexpr int kChunkWords = 32;     // mask words per lane per chunk
constexpr int kMaskCacheSel = 512;  // train items per user cached in LDS by the fused selection (as k_select_flat's kMaskCache)

// SELECT (round 4; the narrow rows' call): what used to be the next launch -- k_select_flat, one wave per user as well -- is the tail
// of this one.  The kept (score, item) pairs stay in LDS: the item ids go back INTO the candidate array (kept pair number p
// overwrites ids[p]; p never runs ahead of the candidates already chained, and everything later reads ids further on), the scores
// into a 4 KB array; then the train items are dropped, the k-th largest is found by the bitwise search of select_core and the
// winners are sorted and written -- no flat list in global memory, no totals round trip, one launch less per call.  A user with
// no candidates, too many, or fewer than k unmasked ones goes to the exact fallback.  For SMALL calls only (score_topk_impl: up
// to 4096 users): this kernel's LDS holds it to 8 waves per CU, and the selection -- a latency chain of ballots and a 64-lane
// bitonic sort -- runs four times as many waves per CU as its own launch (k_select_flat): at 16 384 users the fused launch took
// 163 us where the pair took ~116 (rocprofv3, profiles/r04_experiments.md section 2); at 2048 the two are equal (28 us) and the
// fused form saves a launch.
// Waves per workgroup.  (Round 4 tried TWO for the SELECT form -- 38 KB of LDS, so that a workgroup fits on a CU beside a workgroup of
// the bf16 filter (120 KB) and one call's chains could run under the next call's filter: no gain, 2048-user calls on four streams
// 56 -> 58 us, profiles/r04_experiments.md section 2 -- the launches of a call each fill the chip by themselves.)
template <bool SELECT>
struct RescoreShape {
    static constexpr int kWaves = 4;
};

template <bool ALIGNED4, bool FROM_LIST, bool SELECT = false>     // ALIGNED4: rows are multiples of 16 bytes (d % 4 == 0): one float4 per (row, piece), else four scalars
__global__ __launch_bounds__(RescoreShape<SELECT>::kWaves * 64) void k_rescore(const RescoreArgs a)
{
    static_assert(!(SELECT && FROM_LIST), "the fused selection is the narrow rows' (d <= 128)");
    // LDS per wave decides how many waves a CU runs, and a wave here is a chain of ~10 dependent round trips: the launch's time is
    // (waves / resident waves) x that chain.  Narrow rows (from the mask; d <= 128) keep 128 floats of the user's row and 1024
    // candidates -- what k_select_flat takes anyway -- : 12.9 KB per wave, 12 waves per CU (the register limit) where round 3's
    // 18.5 KB (a 1024-float row, 1536 candidates) allowed 8.
    constexpr int kSuLen = FROM_LIST ? kMaxPreD : 128;
    constexpr int kCap = FROM_LIST ? kUserCap : kSelCap;
    constexpr int WPB = RescoreShape<SELECT>::kWaves;
    __shared__ int ids_all[WPB][kCap];
    __shared__ float tiles[WPB][kWave * kTileRow];
    __shared__ __attribute__((aligned(16))) float su_all[WPB][kSuLen];   // the user's row
    __shared__ float kscore_all[WPB][SELECT ? kSelCap : 1];
    static_assert(kWave * kTileRow * sizeof(float) >= kMaskCacheSel * sizeof(int) + kWave * sizeof(float2), "select scratch inside the tile");
    const int lane = lane_id();
    const int w = threadIdx.x >> 6;
    const int b = blockIdx.x * WPB + w;
    if (b >= a.B)
        return;
    int *ids = ids_all[w];
    float *tile = tiles[w];
    float *su = su_all[w];
    float *kscore = kscore_all[w];
    auto give_up = [&](int total) {      // the user goes to the exact fallback (k_brute_part)
        if (lane == 0) {
            a.totals[b] = total;
            if constexpr (SELECT)
                a.flagged[1 + atomicAdd(a.flagged, 1)] = b;
        }
    };
    int mb = 0, me = 0;
    if constexpr (SELECT) {
        if (a.mask_rowptr) {
            mb = a.mask_rowptr[b];
            me = a.mask_rowptr[b + 1];
        }
    }   // read back as LDS broadcasts: left to the compiler, u_k of the chains is a vector load from global
                             // memory per fmaf (the address is not proven uniform), 5 us per 64-entry step
    {
        const float *__restrict__ urow = a.U + (size_t)(a.user_ids ? a.user_ids[b] : (int64_t)b) * a.d;
        for (int k = lane; k < a.d; k += kWave)
            su[k] = urow[k];
    }
    const unsigned *__restrict__ mrow = a.mask + (size_t)b * 2 * a.Wh;
    const int n_words = 2 * a.Wh;
    // (1) extract: lane l of a chunk owns words l * 32 .. l * 32 + 31 (contiguous: 128-byte reads per lane)
    int n = 0;
    if constexpr (FROM_LIST) {
        n = a.surv_n[b];
        if (n > kCap) {
            give_up(kOverflow);
            return;
        }
        for (int j = lane; j < n; j += kWave)
            ids[j] = a.surv[(size_t)b * a.surv_cap + j];
    }
    bool summarised = false;
    if constexpr (!FROM_LIST) {
        summarised = a.summ.words != nullptr;
        if (summarised) {
            // (1a) the flagged stages of the user: lane l of a chunk reads summary word c0 + l; bit t of word q of (half hh, split sp)
            // is stage sp (items_per_split / stage_items) + 32 q + t.  Their codes (2 stage + hh) are parked in the tile.
            int *stg = reinterpret_cast<int *>(tile);
            const int per_half = a.summ.n_splits * a.summ.sw;
            const int spl_stages = a.summ.items_per_split / a.summ.stage_items;
            const int n_stages = (a.n_units * kStage + a.summ.stage_items - 1) / a.summ.stage_items;     // (n_units 64-item units)
            const unsigned *__restrict__ srow = a.summ.words + (size_t)b * 2 * per_half;
            int ns = 0;
            for (int c0 = 0; c0 < 2 * per_half; c0 += kWave) {
                const int j = c0 + lane;
                unsigned sw = 0u;
                int hh = 0, stage0 = 0;
                if (j < 2 * per_half) {
                    hh = j >= per_half ? 1 : 0;
                    const int r = j - hh * per_half, sp = r / a.summ.sw, q = r - sp * a.summ.sw;
                    stage0 = sp * spl_stages + 32 * q;
                    // words past a split's last stage, and the splits past the catalogue, were never written
                    if (32 * q < spl_stages && stage0 < n_stages)
                        sw = srow[j];
                }
                const int mine = __popc(sw);
                int incl = mine;
#pragma unroll
                for (int o = 1; o < kWave; o <<= 1) {
                    const int t = __shfl_up(incl, o);
                    if (lane >= o)
                        incl += t;
                }
                const int chunk_n = __builtin_amdgcn_readlane(incl, kWave - 1);
                if (ns + chunk_n > kCap) {       // more flagged stages than candidates a user may bring
                    give_up(kOverflow);
                    return;
                }
                int off = ns + incl - mine;
                while (sw) {
                    const int t = __ffs(sw) - 1;
                    sw &= sw - 1u;
                    stg[off++] = 2 * (stage0 + t) + hh;
                }
                ns += chunk_n;
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
            // (1b) their words -> item ids
            const int ups = a.summ.stage_items / kStage;       // words per stage (4 or 8 at d <= 64, 2 or 4 at d <= 128)
            for (int c0 = 0; c0 < ns; c0 += kWave) {
                unsigned word[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
                int hh = 0, u0w = 0;
                if (c0 + lane < ns) {
                    const int code = stg[c0 + lane];
                    hh = code & 1;
                    u0w = (code >> 1) * ups;                   // first 64-item unit (= word inside the row half) of the stage
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (i < ups && u0w + i < a.n_units)    // (the catalogue's last stage may be partial: its other words were never written)
                            word[i] = mrow[hh * a.Wh + u0w + i];
                }
                int mine = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    mine += __popc(word[i]);
                int incl = mine;
#pragma unroll
                for (int o = 1; o < kWave; o <<= 1) {
                    const int t = __shfl_up(incl, o);
                    if (lane >= o)
                        incl += t;
                }
                const int chunk_n = __builtin_amdgcn_readlane(incl, kWave - 1);
                if (n + chunk_n > kCap) {
                    give_up(kOverflow);
                    return;
                }
                int off = n + incl - mine;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    unsigned wd = word[i];
                    const int base = (u0w + i) * 64 + 4 * hh;
                    while (wd) {
                        const int t = __clz(wd);
                        wd &= ~(0x80000000u >> t);
                        ids[off++] = base + (t < 16 ? 0 : 32) + (t & 3) + 8 * ((t & 15) >> 2);
                    }
                }
                n += chunk_n;
            }
        }
    }
    for (int c0 = 0; !FROM_LIST && !summarised && c0 < n_words; c0 += kWave * kChunkWords) {
        unsigned word[kChunkWords];
        const int j0 = c0 + lane * kChunkWords;
        if (j0 + kChunkWords <= n_words && ((size_t)mrow & 15) == 0 && (n_words & 3) == 0) {
#pragma unroll
            for (int i = 0; i < kChunkWords; i += 4) {
                const uint4 t = *reinterpret_cast<const uint4 *>(mrow + j0 + i);
                word[i] = t.x, word[i + 1] = t.y, word[i + 2] = t.z, word[i + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < kChunkWords; ++i)
                word[i] = j0 + i < n_words ? mrow[j0 + i] : 0u;
        }
        if (a.Wh != a.n_units) {      // the row halves are padded to 16-byte multiples: nothing ever wrote the pad words
#pragma unroll
            for (int i = 0; i < kChunkWords; ++i) {
                const int j = j0 + i;
                word[i] = (j >= a.Wh ? j - a.Wh : j) < a.n_units ? word[i] : 0u;
            }
        }
        int mine = 0;
#pragma unroll
        for (int i = 0; i < kChunkWords; ++i)
            mine += __popc(word[i]);
        int incl = mine;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o)
                incl += t;
        }
        const int chunk_n = __builtin_amdgcn_readlane(incl, kWave - 1);
        if (n + chunk_n > kCap) {    // too many candidates (tau = -inf, non-finite norms, a degenerate threshold ...)
            give_up(kOverflow);
            return;
        }
        int off = n + incl - mine;
        if (mine) {
#pragma unroll
            for (int i = 0; i < kChunkWords; ++i) {
                unsigned wd = word[i];
                const int j = j0 + i;
                const int hh = j >= a.Wh ? 1 : 0;
                const int base = (j - hh * a.Wh) * 64 + 4 * hh;
                while (wd) {
                    const int t = __clz(wd);          // register index: bit 31 - t
                    wd &= ~(0x80000000u >> t);
                    ids[off++] = base + (t < 16 ? 0 : 32) + (t & 3) + 8 * ((t & 15) >> 2);
                }
            }
        }
        n += chunk_n;
    }
    if (n == 0) {
        give_up(0);
        return;
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    // (2) fp32 chains, (3) append
    const float tau = a.tau[(size_t)b * a.tau_stride];
    float2 *__restrict__ lg = a.lists + (size_t)b * a.list_cap;
    const int sub = lane >> 3, q = lane & 7;   // loader role: row sub + 8 pass, 16-byte piece q of the k-block
    const int nkb = (a.d + kKB - 1) / kKB;
    const int total = ((n + kWave - 1) / kWave) * nkb;
    int kept = 0;

    // every step's loads are issued unconditionally (a step past the end repeats the last one): with a branch around them, or
    // around the float4 / scalar forms, hipcc's wait-count pass cannot count the loads in flight and puts s_waitcnt vmcnt(0)
    // in front of every LDS transpose -- each 64-entry step then pays a whole memory round trip (measured: 6 us per step)
    auto issue = [&](int step_, float4 (&v)[8]) {
        const int step = min(step_, total - 1);
        const int t0 = (step / nkb) * kWave, k = (step % nkb) * kKB + 4 * q;
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const float *__restrict__ p = a.It + (size_t)ids[min(t0 + pass * 8 + sub, n - 1)] * a.d;
            if constexpr (ALIGNED4) {
                v[pass] = *reinterpret_cast<const float4 *>(p + min(k, a.d - 4));    // k past the row: a copy nobody chains
            } else {
                v[pass].x = p[min(k + 0, a.d - 1)];
                v[pass].y = p[min(k + 1, a.d - 1)];
                v[pass].z = p[min(k + 2, a.d - 1)];
                v[pass].w = p[min(k + 3, a.d - 1)];
            }
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
                s = fmaf(su[kb + kk], row[kk], s);
        } else {
            for (int kk = 0; kk < a.d - kb; ++kk)
                s = fmaf(su[kb + kk], row[kk], s);
        }
        if (kb + kKB >= a.d) {
            const bool keep = t0 + lane < n && s > tau;
            const unsigned long long m = __ballot(keep);
            const int pos = kept + __popcll(m & ((1ull << lane) - 1ull));
            if constexpr (SELECT) {
                const int id = ids[min(t0 + lane, n - 1)];     // every lane's read of the step's ids BEFORE any write below
                __builtin_amdgcn_s_waitcnt(0xC07F);
                __builtin_amdgcn_wave_barrier();
                if (keep && pos < kSelCap) {                   // pos <= t0 + lane: at or behind the candidate it came from
                    ids[pos] = id;
                    kscore[pos] = s;
                }
            } else {
                if (keep && pos < a.list_cap)
                    lg[pos] = make_float2(s, __int_as_float(ids[t0 + lane]));
            }
            kept += __popcll(m);
        }
    };
    // three steps ahead: a wave is alone with its user, so the depth of its own prefetch is what hides the row latency
    float4 v0[8], v1[8], v2[8], v3[8];
    issue(0, v0);
    issue(1, v1);
    issue(2, v2);
    for (int step = 0; step < total; step += 4) {
        issue(step + 3, v3);
        process(step, v0);
        if (step + 1 >= total)
            break;
        issue(step + 4, v0);
        process(step + 1, v1);
        if (step + 2 >= total)
            break;
        issue(step + 5, v1);
        process(step + 2, v2);
        if (step + 3 >= total)
            break;
        issue(step + 6, v2);
        process(step + 3, v3);
    }
    if constexpr (!SELECT) {
        if (lane == 0)
            a.totals[b] = kept;
        return;
    } else {
        // (4) the exact selection (k_select_flat's, on the LDS list): train items out, k-th largest, winners sorted
        if (kept == 0 || kept > kSelCap) {
            give_up(kept);
            return;
        }
        if (lane == 0)
            a.totals[b] = kept;
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();     // the last step's reads of the tile and its list writes are done
        int *smask = reinterpret_cast<int *>(tile);                              // the tile is free: mask cache + pack buffer
        float2 *spack = reinterpret_cast<float2 *>(tile + kMaskCacheSel);
        const bool cached = (me - mb) <= kMaskCacheSel;
        if (cached)
            for (int j = lane; j < me - mb; j += kWave)
                smask[j] = a.mask_items[mb + j];
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        bool ok = true;
        float out_v = -INFINITY;
        int out_i = INT_MAX;
        auto run = [&](auto vpl_tag) {
            constexpr int VPL = decltype(vpl_tag)::value;
            unsigned key[VPL];
            int idx[VPL];
            int n_valid = 0;
#pragma unroll
            for (int sgm = 0; sgm < VPL; ++sgm) {
                const int j = min(lane + kWave * sgm, kept - 1);
                const int si = ids[j];
                const float sv = kscore[j];
                bool on = lane + kWave * sgm < kept;
                if (on && (cached ? sorted_contains(smask, 0, me - mb, si) : sorted_contains(a.mask_items, mb, me, si)))
                    on = false;     // a train item (base_model.py:257-258 sets them to -inf)
                idx[sgm] = on ? si : INT_MAX;
                key[sgm] = on ? ordered_key(sv) : 0u;
                n_valid += __popcll(__ballot(on));
            }
            ok = n_valid >= a.k;
            if (ok)
                select_core<VPL>(key, idx, a.k, lane, out_v, out_i, spack);
        };
        if (kept <= 4 * kWave)
            run(std::integral_constant<int, 4>{});
        else if (kept <= 8 * kWave)
            run(std::integral_constant<int, 8>{});
        else
            run(std::integral_constant<int, kSelVPL>{});
        if (!ok && lane == 0)
            a.flagged[1 + atomicAdd(a.flagged, 1)] = b;
        if (ok && lane < a.k) {
            a.out_val[(size_t)b * a.k + lane] = a.do_round ? round4(out_v) : out_v;
            a.out_idx[(size_t)b * a.k + lane] = out_i;
        }
    }
}

}  // namespace


bool prefilter_supports(int d) { return d <= 128 || (d <= kMaxPreD && d % 8 == 0); }

int launch_item_norms(const float *It, int I, int d, float *norms, hipStream_t s)
{
    hipLaunchKernelGGL(k_item_norms, dim3(min(1024, (I + 63) / 64)), dim3(256), 0, s, It, I, d, norms);
    return check_launch("k_item_norms");
}

int launch_sample_bf16(const float *U, const int64_t *user_ids, int B, const float *It, int m, int d, int stride, float *S, int64_t ld,
                       hipStream_t s)
{
    SampleArgs a{U, user_ids, It, S, ld, B, m, d, stride, nullptr, 0};
    const dim3 grid((B + 255) / 256, (m + 127) / 128), block(512);
    if (d == 64)
        hipLaunchKernelGGL((k_sample_bf16<4, true, false>), grid, block, 0, s, a);
    else if (d < 64)
        hipLaunchKernelGGL((k_sample_bf16<4, false, false>), grid, block, 0, s, a);
    else if (d == 128)
        hipLaunchKernelGGL((k_sample_bf16<8, true, false>), grid, block, 0, s, a);
    else
        hipLaunchKernelGGL((k_sample_bf16<8, false, false>), grid, block, 0, s, a);
    return check_launch("k_sample_bf16");
}

// the TOP form: S[b][2 blk .. 2 blk + 1] = the two largest unmasked sample scores of user b in block blk (128 sampled items);
// bits: workspace of B x 4 ceil(m / 128) words for the train-item bitmap (written here when a mask is given)
bool sample_top_supports(int d, int m) { return d <= 128 && 4 * ((m + 127) / 128) <= kSampleBitsMaxWords; }

// the same sample from the packed item operand (prefiltered entry point)
int launch_sample_pack_top(const float *U, const int64_t *user_ids, int B, const void *ipack, int m, int d, int stride, const int *mask_rowptr,
                           const int *mask_items, unsigned *bits, float *S, int64_t ld, hipStream_t s)
{
    const int n_blk = (m + 127) / 128;
    if (mask_rowptr) {
        hipLaunchKernelGGL(k_sample_bits, dim3((B + 3) / 4), dim3(256), (size_t)4 * 4 * n_blk * sizeof(unsigned), s, mask_rowptr, mask_items, bits,
                           B, 4 * n_blk, stride, m);
        const int rc = check_launch("k_sample_bits");
        if (rc != TGCN_OK)
            return rc;
    }
    // as many blocks per workgroup as still leave the chip two workgroups per CU: the users' fragments are built once per workgroup
    const int tiles = (B + 255) / 256;
    const int bpw = max(1, min(n_blk, tiles * n_blk / 256));
    SamplePackArgs a{U, user_ids, static_cast<const unsigned char *>(ipack), S, ld, B, m, d, stride, mask_rowptr ? bits : nullptr, 4 * n_blk, bpw};
    const dim3 grid(tiles, (n_blk + bpw - 1) / bpw), block(512);
    if (d == 64)
        hipLaunchKernelGGL((k_sample_pack_top<4, true>), grid, block, 0, s, a);
    else if (d < 64)
        hipLaunchKernelGGL((k_sample_pack_top<4, false>), grid, block, 0, s, a);
    else if (d == 128)
        hipLaunchKernelGGL((k_sample_pack_top<8, true>), grid, block, 0, s, a);
    else
        hipLaunchKernelGGL((k_sample_pack_top<8, false>), grid, block, 0, s, a);
    return check_launch("k_sample_pack_top");
}

int launch_sample_top(const float *U, const int64_t *user_ids, int B, const float *It, int m, int d, int stride, const int *mask_rowptr,
                      const int *mask_items, unsigned *bits, float *S, int64_t ld, hipStream_t s)
{
    const int n_blk = (m + 127) / 128;
    if (mask_rowptr) {
        hipLaunchKernelGGL(k_sample_bits, dim3((B + 3) / 4), dim3(256), (size_t)4 * 4 * n_blk * sizeof(unsigned), s, mask_rowptr, mask_items, bits,
                           B, 4 * n_blk, stride, m);
        const int rc = check_launch("k_sample_bits");
        if (rc != TGCN_OK)
            return rc;
    }
    SampleArgs a{U, user_ids, It, S, ld, B, m, d, stride, mask_rowptr ? bits : nullptr, 4 * n_blk};
    const dim3 grid((B + 255) / 256, n_blk), block(512);
    if (d == 64)
        hipLaunchKernelGGL((k_sample_bf16<4, true, true>), grid, block, 0, s, a);
    else if (d < 64)
        hipLaunchKernelGGL((k_sample_bf16<4, false, true>), grid, block, 0, s, a);
    else if (d == 128)
        hipLaunchKernelGGL((k_sample_bf16<8, true, true>), grid, block, 0, s, a);
    else
        hipLaunchKernelGGL((k_sample_bf16<8, false, true>), grid, block, 0, s, a);
    return check_launch("k_sample_bf16<top>");
}

int launch_user_bound(const float *U, const int64_t *user_ids, int B, int d, float *ubound, hipStream_t s)
{
    hipLaunchKernelGGL(k_user_bound, dim3((B + 3) / 4), dim3(256), 0, s, U, user_ids, B, d, ubound);
    return check_launch("k_user_bound");
}

size_t item_pack_bytes(int I, int d) { return prefilter_supports(d) ? (size_t)I * (32 * pack_ksteps(d) + 16) : 0; }

int launch_item_pack(const float *It, int I, int d, void *pack, hipStream_t s)
{
    hipLaunchKernelGGL(k_item_pack, dim3(min(1024, (I + 63) / 64)), dim3(256), 0, s, It, I, d, static_cast<unsigned char *>(pack));
    return check_launch("k_item_pack");
}

int prefilter_stage_items(int d, bool long_stages) { return (d <= 64 ? 256 : 128) * (long_stages ? 2 : 1); }      // the ST of launch_prefilter's instantiations

int launch_prefilter(const float *U, const int64_t *user_ids, int B, const void *ipack, int I, int d, const float *tau, int tau_stride,
                     const float *ubound, unsigned *mask, int Wh, int S, int items_per_split, bool wide, bool long_stages,
                     const PassSummary &summ, hipStream_t s)
{
    constexpr int UT = kPreWaves * 32;
    const int n_tiles = (B + UT - 1) / UT;
    wide = wide && items_per_split % 256 == 0;         // (a stage's words as one aligned store: the splits must be stage multiples)
    long_stages = long_stages && wide && items_per_split % prefilter_stage_items(d, true) == 0;
    if (summ.words && (!wide || summ.n_splits != S || summ.items_per_split != items_per_split ||
                       summ.stage_items != prefilter_stage_items(d, long_stages) || summ.sw != (items_per_split / summ.stage_items + 31) / 32))
        return fail_arg("launch_prefilter: the stage summary does not describe this launch");
    PreArgs a{U, user_ids, static_cast<const unsigned char *>(ipack), item_pack_bytes(I, d), tau, tau_stride, ubound, mask, Wh, B, I, d,
              items_per_split, n_tiles, S, summ.words, summ.sw};
    const dim3 grid(8u * (unsigned)((n_tiles * S + 7) / 8)), block(kPreWaves * 64);
#define TGCN_PRE_LAUNCH(KS, FULLK, ST)                                                                                  \
    do {                                                                                                                \
        if (long_stages)                                                                                                \
            hipLaunchKernelGGL((k_score_prefilter<KS, FULLK, 2 * ST, kPreWaves, true, 2>), grid, block, 0, s, a);       \
        else if (wide)                                                                                                  \
            hipLaunchKernelGGL((k_score_prefilter<KS, FULLK, ST, kPreWaves, true>), grid, block, 0, s, a);              \
        else                                                                                                            \
            hipLaunchKernelGGL((k_score_prefilter<KS, FULLK, ST, kPreWaves, false>), grid, block, 0, s, a);             \
    } while (0)
    if (d == 64)
        TGCN_PRE_LAUNCH(4, true, 256);
    else if (d < 64)
        TGCN_PRE_LAUNCH(4, false, 256);
    else if (d == 128)
        TGCN_PRE_LAUNCH(8, true, 128);
    else
        TGCN_PRE_LAUNCH(8, false, 128);
#undef TGCN_PRE_LAUNCH
    return check_launch("k_score_prefilter");
}

int pack_row_bytes(int d) { return 32 * pack_ksteps(d) + 16; }

int launch_prefilter_wide(const float *U, const int64_t *user_ids, int B, const void *ipack, int I, int d, const float *tau, int tau_stride,
                          const float *ubound, void *logs, int *counts, int S, int items_per_split, int cap2, hipStream_t s)
{
    PreWideArgs a{PreArgs{U, user_ids, static_cast<const unsigned char *>(ipack), item_pack_bytes(I, d), tau, tau_stride, ubound, nullptr, 0, B,
                          I, d, items_per_split},
                  static_cast<float2 *>(logs), counts, S, cap2, 1, nullptr, 0, (B + 127) / 128};
    const dim3 grid(wide_grid((B + 127) / 128, S)), block(512);      // 128 users per workgroup, K split between the two waves of a SIMD
    const int ks = pack_ksteps(d);
    if (ks <= 16)
        hipLaunchKernelGGL((k_score_prefilter_wide<16, false>), grid, block, 0, s, a);
    else if (ks <= 32)
        hipLaunchKernelGGL((k_score_prefilter_wide<32, false>), grid, block, 0, s, a);
    else if (ks == 56)
        hipLaunchKernelGGL((k_score_prefilter_wide<56, false>), grid, block, 0, s, a);
    else if (ks == 60)
        hipLaunchKernelGGL((k_score_prefilter_wide<60, false>), grid, block, 0, s, a);
    else
        hipLaunchKernelGGL((k_score_prefilter_wide<64, false>), grid, block, 0, s, a);
    return check_launch("k_score_prefilter_wide");
}

// the threshold sample of wide rows from the same kernel: S[b][j] ~ <U[b], It[j * stride]> for j < m, bf16 operands from the pack
int launch_sample_wide(const float *U, const int64_t *user_ids, int B, const void *ipack, int I, int m, int d, int stride, float *S,
                       int64_t ld, hipStream_t s)
{
    const int tiles = (B + 127) / 128, units = (m + kStage - 1) / kStage;
    // as the filter's plan (tgcn_score_fused.hip make_plan): generations of 256 workgroups x (prologue of ~9 units + the split's units);
    // 8192 users: 64 tiles x 4 splits in one generation instead of 8 splits in two, each paying the prologue
    int splits = 1;
    long best = -1;
    for (int c = 1; c <= min(units, 16); ++c) {
        const long cost = (((long)tiles * c + 255) / 256) * (9 + (units + c - 1) / c);
        if (best < 0 || cost < best)
            best = cost, splits = c;
    }
    const int ips = ((units + splits - 1) / splits) * kStage;
    PreWideArgs a{PreArgs{U, user_ids, static_cast<const unsigned char *>(ipack), item_pack_bytes(I, d), nullptr, 0, nullptr, nullptr, 0, B,
                          m, d, ips},
                  nullptr, nullptr, 1, 1, stride, S, ld, tiles};
    const dim3 grid(wide_grid(tiles, (m + ips - 1) / ips)), block(512);
    const int ks = pack_ksteps(d);
    if (ks <= 16)
        hipLaunchKernelGGL((k_score_prefilter_wide<16, true>), grid, block, 0, s, a);
    else if (ks <= 32)
        hipLaunchKernelGGL((k_score_prefilter_wide<32, true>), grid, block, 0, s, a);
    else if (ks == 56)
        hipLaunchKernelGGL((k_score_prefilter_wide<56, true>), grid, block, 0, s, a);
    else if (ks == 60)
        hipLaunchKernelGGL((k_score_prefilter_wide<60, true>), grid, block, 0, s, a);
    else
        hipLaunchKernelGGL((k_score_prefilter_wide<64, true>), grid, block, 0, s, a);
    return check_launch("k_score_prefilter_wide(sample)");
}

int launch_rescore(const float *U, const int64_t *user_ids, int B, const float *It, int d, const float *tau, int tau_stride,
                   const unsigned *mask, int Wh, int n_units, const PassSummary &summ, void *lists, int *totals, int list_cap, hipStream_t s)
{
    RescoreArgs a{U, user_ids, It, tau, tau_stride, mask, Wh, n_units, nullptr, nullptr, 0, static_cast<float2 *>(lists), totals, B, d,
                  list_cap, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, summ};
    if ((d & 3) == 0)
        hipLaunchKernelGGL((k_rescore<true, false>), dim3((B + 3) / 4), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((k_rescore<false, false>), dim3((B + 3) / 4), dim3(256), 0, s, a);
    return check_launch("k_rescore");
}

int launch_rescore_select(const float *U, const int64_t *user_ids, int B, const float *It, int d, const float *tau, int tau_stride,
                          const unsigned *mask, int Wh, int n_units, const PassSummary &summ, int *totals, const int *mask_rowptr,
                          const int *mask_items, int k, int do_round, float *out_val, int64_t *out_idx, int *flagged, hipStream_t s)
{
    if (d > 128)
        return fail_arg("launch_rescore_select: narrow rows only");
    RescoreArgs a{U, user_ids, It, tau, tau_stride, mask, Wh, n_units, nullptr, nullptr, 0, nullptr, totals, B, d, 0,
                  mask_rowptr, mask_items, out_val, out_idx, flagged, k, do_round, summ};
    constexpr int WPB = RescoreShape<true>::kWaves;
    if ((d & 3) == 0)
        hipLaunchKernelGGL((k_rescore<true, false, true>), dim3((B + WPB - 1) / WPB), dim3(WPB * 64), 0, s, a);
    else
        hipLaunchKernelGGL((k_rescore<false, false, true>), dim3((B + WPB - 1) / WPB), dim3(WPB * 64), 0, s, a);
    return check_launch("k_rescore(select)");
}

int launch_rescore_list(const float *U, const int64_t *user_ids, int B, const float *It, int d, const float *tau, int tau_stride,
                        const int *surv, const int *surv_n, int surv_cap, void *lists, int *totals, int list_cap, hipStream_t s)
{
    RescoreArgs a{U, user_ids, It, tau, tau_stride, nullptr, 0, 0, surv, surv_n, surv_cap, static_cast<float2 *>(lists), totals, B, d,
                  list_cap, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, PassSummary{}};
    hipLaunchKernelGGL((k_rescore<true, true>), dim3((B + 3) / 4), dim3(256), 0, s, a);
    return check_launch("k_rescore(list)");
}

}  // namespace tgcn

#ifdef TGCN_CHECK_DMA
// checked build only: (stale, checked) 16-byte pieces since the library was loaded; synchronises the device
extern "C" int tgcn_debug_dma_counts(unsigned long long *stale_host, unsigned long long *checked_host)
{
    if (hipDeviceSynchronize() != hipSuccess)
        return TGCN_ERR_HIP;
    if (hipMemcpyFromSymbol(stale_host, HIP_SYMBOL(tgcn::g_dma_stale_pieces), sizeof(unsigned long long)) != hipSuccess ||
        hipMemcpyFromSymbol(checked_host, HIP_SYMBOL(tgcn::g_dma_checked_pieces), sizeof(unsigned long long)) != hipSuccess)
        return TGCN_ERR_HIP;
    return TGCN_OK;
}
#endif
