// K5+K6+K7+K8 fused: top-k of masked user x item scores without materialising the [B, I] matrix.
//
// Replaces the per-batch chain torch.matmul -> pandas explode + -inf scatter -> torch.topk -> round of
// TextGCN/base_model.py:254-263.  gfx950 only.
//
// Plan (all launches on one stream, no host round trip):
//   1. threshold estimate.  Score every user against a strided sample of the items (every 32nd item, every 16th for
//      narrow rows -- make_plan), mask the sampled train items, take the r-th largest sample score as tau_u.  The bar
//      must not land above the user's k-th best score (the user then pays step 4): that takes r of its best k - 1 items
//      inside the sample, P(Bin(k - 1, 1 / stride) >= r); r is the smallest rank that keeps this under 3e-6 per user
//      (k = 40: r = 10 at stride 32, ~320 candidates per user; r = 12 at stride 16, ~190.  At 7e-5, r = 8 of 32, a
//      16384-user call had a fallback launch with real work two times out of three).
//   2. k_score_filter: the fp32 MFMA GEMM over ALL items; the 32x32 accumulators are compared against tau_u in
//      registers and only scores > tau_u are written, as (score, item) pairs, to a log private to the lane that
//      owns that (user, row-half) -- no atomics, no [B, I] traffic.  Items on MFMA rows (A operand, staged
//      through LDS), users on columns (B operand, held in registers for the whole pass) so that every lane's
//      16 results belong to ONE user and the threshold is one register.
//   3. k_select: one wave per user merges its logs, drops train items (binary search in the user's sorted
//      train list) and keeps the best k by (score desc, item asc).  The result is exact whenever at least k
//      unmasked candidates were logged and no log overflowed; otherwise the user is flagged.
//   4. k_brute: flagged users (threshold too high, log overflow, fewer than k unmasked items ...) are
//      rescored against every item with the same k-ordered fmaf chain and selected exactly.  Rare by
//      construction, and it makes the whole path exact.
// Every score is the k-ordered fp32 fmaf chain (MFMA 32x32x2 f32 or v_fma), so all four steps agree bit for
// bit with the dense path and with the CPU restatement used in the parity tests.
#include <mutex>
#include <type_traits>

#include "tgcn_internal.h"
#include "tgcn_topk.h"

namespace tgcn {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kSampleStride = 32;  // items per sampled item (16 where the denser sample pays: make_plan)
constexpr double kBarRisk = 3e-6;  // per user: the chance that the bar tau_u lands above the user's k-th best score (-> exact fallback)
constexpr int kUsersPerWG = 128;   // 4 waves x 32 users
constexpr size_t kLongStagePackBytes = (size_t)64 << 20;   // item packs up to this size take the filter's ring of two (launch_prefilter)
constexpr int kSummaryMinWords = 2048;   // pass-bit words per (user, row half) from which the bf16 filter keeps its stage summary (131 072 items)
constexpr int kStage = 64;         // items per LDS stage (2 MFMA sub-tiles of 32)
constexpr int kSmallI = 8192;      // below this the dense path (score -> mask -> top-k) is used

#define TGCN_PROBE(slot) do { } while (0)

struct FilterArgs {
    const float *__restrict__ U;
    const int64_t *__restrict__ user_ids;
    const float *__restrict__ It;
    const float *__restrict__ tau;  // tau of user b at tau[b * tau_stride]
    int tau_stride;
    float2 *__restrict__ logs;      // [B][S][2][cap2]  (score, item-as-float-bits)
    int *__restrict__ counts;       // [B][S][2]
    int B, I, d;
    int S;              // item splits (gridDim.y)
    int items_per_split;  // multiple of kStage
    int cap2;
};

// stage `rows` rows x 4*DQ columns (zero-filled past n_rows / d) with the (k0,k2,k1,k3) group swizzle
// FULLK (d == 4*DQ): rows past the table are clamped to the last valid row instead of zero-filled -- their
// scores are garbage but never used (the epilogue tests item < i_end, padded users have tau = +inf) -- so every
// load is unconditional and stays in flight across the MFMA block.  (Any `cond ? load : 0` form makes hipcc
// branch around the load and wait for it inside the branch.)  Other widths need zeros in the K padding and take
// the predicated path.
template <int DQ, bool FULLK, int ROWS = kStage>
__device__ __forceinline__ void load_rows(float4 (&v)[(ROWS * DQ) / 256], const float *__restrict__ src,
                                          const int64_t *__restrict__ ids, int row0, int n_rows, int d)
{
    constexpr int N = (ROWS * DQ) / 256;
    // source rows first -- ALL the id loads of a gathered tile in one batch, then all the row loads: with the id fetch and
    // its row fetch written per element, hipcc emits N dependent (id -> wait -> row -> wait) round trips (16 of them in the
    // user-tile prologue of the filter kernels, ~10 % of a 2048-user launch)
    auto fetch = [&](int i, size_t srow) {
        const int f = i * 256 + threadIdx.x;
        const int r = f / DQ, q = f % DQ;
        const int row = row0 + r, k = q * 4;
        const float *p = src + srow * d;
        if constexpr (FULLK) {
            v[i] = *reinterpret_cast<const float4 *>(p + k);
        } else {
            const bool ok = row < n_rows;
            v[i].x = (ok && k + 0 < d) ? p[k + 0] : 0.0f;
            v[i].y = (ok && k + 1 < d) ? p[k + 1] : 0.0f;
            v[i].z = (ok && k + 2 < d) ? p[k + 2] : 0.0f;
            v[i].w = (ok && k + 3 < d) ? p[k + 3] : 0.0f;
        }
    };
    if (ids) {
        int64_t srow[N];
#pragma unroll
        for (int i = 0; i < N; ++i)
            srow[i] = ids[min(row0 + (i * 256 + (int)threadIdx.x) / DQ, n_rows - 1)];
#pragma unroll
        for (int i = 0; i < N; ++i)
            fetch(i, (size_t)srow[i]);
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i)
            fetch(i, (size_t)min(row0 + (i * 256 + (int)threadIdx.x) / DQ, n_rows - 1));
    }
}

template <int DQ, int ROWS = kStage>
__device__ __forceinline__ void store_rows(float *__restrict__ dst, const float4 (&v)[(ROWS * DQ) / 256])
{
    constexpr int N = (ROWS * DQ) / 256;
    constexpr int ROW = 4 * DQ + 2;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int f = i * 256 + threadIdx.x;
        const int r = f / DQ, q = f % DQ;
        float *o = dst + r * ROW + q * 4;
        *reinterpret_cast<float2 *>(o) = make_float2(v[i].x, v[i].z);
        *reinterpret_cast<float2 *>(o + 2) = make_float2(v[i].y, v[i].w);
    }
}

// DQ = number of 4-wide k groups held per user (d <= 4*DQ).
template <int DQ, bool FULLK>
__global__ __launch_bounds__(256) void k_score_filter(const FilterArgs a)
{
    constexpr int ROW = 4 * DQ + 2;                      // LDS row stride in floats (bank-conflict-free ds_read_b64)
    __shared__ __attribute__((aligned(16))) float smem[2 * kStage * ROW];  // two item stages == one 128-user tile
    const int lane = lane_id();
    const int w = threadIdx.x >> 6;
    const int r32 = lane & 31;
    const int h = lane >> 5;
    const int u0 = blockIdx.x * kUsersPerWG;
    const int split = blockIdx.y;
    const int i_beg = split * a.items_per_split;
    const int i_end = min(a.I, i_beg + a.items_per_split);

    // ---- this wave's 32 users as MFMA B fragments, resident in registers for the whole pass
    {
        float4 v[(kStage * DQ) / 256];
        load_rows<DQ, FULLK>(v, a.U, a.user_ids, u0, a.B, a.d);
        store_rows<DQ>(smem, v);
        load_rows<DQ, FULLK>(v, a.U, a.user_ids, u0 + kStage, a.B, a.d);
        store_rows<DQ>(smem + kStage * ROW, v);
    }
    __syncthreads();
    float2 bf[DQ];
    {
        const float *pu = smem + (w * 32 + r32) * ROW + 2 * h;
#pragma unroll
        for (int q = 0; q < DQ; ++q)
            bf[q] = *reinterpret_cast<const float2 *>(pu + q * 4);
    }
    const int user = u0 + w * 32 + r32;
    const bool user_ok = user < a.B;
    const float tau = user_ok ? a.tau[(size_t)user * a.tau_stride] : INFINITY;
    float2 *__restrict__ log = a.logs + ((size_t)(user_ok ? user : 0) * a.S + split) * 2 * a.cap2 + (size_t)h * a.cap2;
    int cnt = 0;
    __syncthreads();
    // tau has arrived before the loop: with its first use inside, hipcc's wait-count pass merges the pending tau load with the
    // stage prefetch across the back edge and puts an s_waitcnt vmcnt in front of every threshold test
    asm volatile("" ::"v"(tau));

    // ---- item stages: global -> registers (issued before the MFMA block) -> LDS (after it), double buffered
    float4 nxt[(kStage * DQ) / 256];
    load_rows<DQ, FULLK>(nxt, a.It, nullptr, i_beg, i_end, a.d);
    store_rows<DQ>(smem, nxt);
    __syncthreads();
    int buf = 0;
    for (int s0 = i_beg; s0 < i_end; s0 += kStage) {
        load_rows<DQ, FULLK>(nxt, a.It, nullptr, s0 + kStage, i_end, a.d);      // unconditional (clamped rows): see filter_pipelined
        const float *pi = smem + buf * kStage * ROW + r32 * ROW + 2 * h;
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            acc0[r] = 0.0f, acc1[r] = 0.0f;
        // A fragments are fetched QB k-groups ahead of the MFMAs that consume them, so the LDS latency sits
        // under the previous block's matrix work instead of in front of every four MFMAs
        constexpr int QB = 4;   // 16 MFMAs (~1000 cycles) per block: ample cover for the next block's LDS reads
        float2 fa0[2][QB], fa1[2][QB];
#pragma unroll
        for (int q = 0; q < QB; ++q) {
            fa0[0][q] = *reinterpret_cast<const float2 *>(pi + q * 4);
            fa1[0][q] = *reinterpret_cast<const float2 *>(pi + 32 * ROW + q * 4);
        }
#pragma unroll
        for (int g = 0; g < DQ / QB; ++g) {
            if (g + 1 < DQ / QB) {
#pragma unroll
                for (int q = 0; q < QB; ++q) {
                    fa0[(g + 1) & 1][q] = *reinterpret_cast<const float2 *>(pi + ((g + 1) * QB + q) * 4);
                    fa1[(g + 1) & 1][q] = *reinterpret_cast<const float2 *>(pi + 32 * ROW + ((g + 1) * QB + q) * 4);
                }
            }
            // pin the order: hipcc otherwise sinks the reads to just in front of their first use and the wave
            // stalls on lgkmcnt once per four MFMAs
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                const float2 b2 = bf[g * QB + q];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[g & 1][q].x, b2.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[g & 1][q].x, b2.x, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[g & 1][q].y, b2.y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[g & 1][q].y, b2.y, acc1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // C/D layout: col (user) = lane & 31, row (item) = (reg & 3) + 8 * (reg >> 2) + 4 * h.  ~0.6 % of the scores
        // pass.  Rows past i_end exist only in the last stage of the last split: handled by a uniform branch so that
        // the common path is one compare + one exec-masked block per register.
        const int lim = i_end - s0;
        if (lim >= kStage) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (acc0[r] > tau) {
                    if (cnt < a.cap2)
                        log[cnt] = make_float2(acc0[r], __int_as_float(s0 + (r & 3) + 8 * (r >> 2) + 4 * h));
                    ++cnt;
                }
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (acc1[r] > tau) {
                    if (cnt < a.cap2)
                        log[cnt] = make_float2(acc1[r], __int_as_float(s0 + 32 + (r & 3) + 8 * (r >> 2) + 4 * h));
                    ++cnt;
                }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                if (acc0[r] > tau && row < lim) {
                    if (cnt < a.cap2)
                        log[cnt] = make_float2(acc0[r], __int_as_float(s0 + row));
                    ++cnt;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (acc1[r] > tau && row < lim) {
                    if (cnt < a.cap2)
                        log[cnt] = make_float2(acc1[r], __int_as_float(s0 + row));
                    ++cnt;
                }
            }
        }
        store_rows<DQ>(smem + (buf ^ 1) * kStage * ROW, nxt);
        __syncthreads();
        buf ^= 1;
    }
    if (user_ok)
        a.counts[((size_t)user * a.S + split) * 2 + h] = cnt;
}

// d <= 128 form of k_score_filter with the threshold test of stage s-1 folded into the MFMA block of stage s.  A
// 32x32x2 fp32 MFMA keeps the matrix pipe busy for 64 cycles after a 4-cycle issue, so one compare-and-branch per two
// MFMAs runs in the shadow of the wave's own matrix work; with the test after the block (k_score_filter) the two or
// three waves of a SIMD drift into lockstep -- they share the pipe, finish their blocks together and test together --
// and the pipe idles through every epilogue (PMC: 58 % MFMA-busy, 63 % of the wave cycles waiting).  Costs a second
// accumulator set (stages alternate between them).
template <int DQ, bool FULLK, int ST>
__device__ __forceinline__ void filter_pipelined(const FilterArgs &a)
{
    static_assert(DQ == 16 || DQ == 32, "one test slot per MFMA pair (d <= 64) or per two pairs (d <= 128)");
    static_assert(ST == kStage || ST == 2 * kStage, "an LDS stage holds one or two 64-item units");
    constexpr int H = ST / kStage;            // 64-item units per LDS stage: one barrier per H units
    constexpr int kSlotsPerReg = (2 * DQ) / 32;
    constexpr int ROW = 4 * DQ + 2;
    __shared__ __attribute__((aligned(16))) float smem[2 * ST * ROW];
    const int lane = lane_id();
    const int w = threadIdx.x >> 6;
    const int r32 = lane & 31;
    const int h = lane >> 5;
    const int u0 = blockIdx.x * kUsersPerWG;
    const int split = blockIdx.y;
    const int i_beg = split * a.items_per_split;
    const int i_end = min(a.I, i_beg + a.items_per_split);
    // prologue: the 128-user tile (gathered through user_ids) and the first item stage are requested together -- one
    // global round trip for the ids, one for all the rows -- then the users pass through LDS into registers
    TGCN_PROBE(0);
    const int user = u0 + w * 32 + r32;
    const bool user_ok = user < a.B;
    float4 nxt[(ST * DQ) / 256];
    float2 bf[DQ];
    {   // the tile in two 64-row halves (one batch of id loads and one of row loads each): a 128-row staging array costs the
        // d <= 64 kernel its third wave per SIMD (spills at the 168-register budget)
        float4 v[(kStage * DQ) / 256];
        load_rows<DQ, FULLK>(v, a.U, a.user_ids, u0, a.B, a.d);
        store_rows<DQ>(smem, v);
        load_rows<DQ, FULLK>(v, a.U, a.user_ids, u0 + kStage, a.B, a.d);
        if (i_beg < i_end)
            load_rows<DQ, FULLK, ST>(nxt, a.It, nullptr, i_beg, i_end, a.d);
        store_rows<DQ>(smem + kStage * ROW, v);
    }
    const float tau = user_ok ? a.tau[(size_t)user * a.tau_stride] : INFINITY;
    __syncthreads();
    // tau has arrived before the loop: with its first use inside, hipcc's wait-count pass merges the pending tau load with the
    // stage prefetch across the back edge and puts an s_waitcnt vmcnt in front of every threshold test
    asm volatile("" ::"v"(tau));
    {
        const float *pu = smem + (w * 32 + r32) * ROW + 2 * h;
#pragma unroll
        for (int q = 0; q < DQ; ++q)
            bf[q] = *reinterpret_cast<const float2 *>(pu + q * 4);
    }
    float2 *__restrict__ log = a.logs + ((size_t)(user_ok ? user : 0) * a.S + split) * 2 * a.cap2 + (size_t)h * a.cap2;
    int cnt = 0;
    __syncthreads();
    if (i_beg >= i_end) {
        if (user_ok)
            a.counts[((size_t)user * a.S + split) * 2 + h] = 0;
        return;
    }
    store_rows<DQ, ST>(smem, nxt);
    __syncthreads();
    TGCN_PROBE(1);
    int buf = 0;
    const int cap2 = a.cap2;

    // C/D layout: col (user) = lane & 31, row (item) = (reg & 3) + 8 * (reg >> 2) + 4 * h
    auto test = [&](float val, int item) {
        if (val > tau) {
            if (cnt < cap2)
                log[cnt] = make_float2(val, __int_as_float(item));
            ++cnt;
        }
    };
    // one 64-item unit: c0/c1 <- scores of items [s0, s0 + 64), read from row `half * 64` of the current LDS stage; p0/p1
    // (items [s_prev, s_prev + 64), a full unit) are tested between the MFMAs when PREV.  FIRST: the unit opens an LDS
    // stage (the next stage's rows are requested from global memory); LAST: it closes one (those rows go to the other LDS
    // buffer, barrier, swap).
    auto unit = [&](auto prev_tag, auto first_tag, auto last_tag, f32x16 &c0, f32x16 &c1, const f32x16 &p0, const f32x16 &p1,
                    int s_prev, int s0, int half) {
        constexpr bool PREV = decltype(prev_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr bool LAST = decltype(last_tag)::value;
        const int stage0 = s0 - half * kStage;          // first item of this LDS stage
        // the next stage's rows: requested unconditionally (rows past the split clamp to its last one; a buffer nobody reads takes
        // the copies) -- a branch around the prefetch leaves hipcc's wait-count pass unable to count the loads in flight
        if (FIRST)
            load_rows<DQ, FULLK, ST>(nxt, a.It, nullptr, stage0 + ST, i_end, a.d);
        const float *pi = smem + (buf * ST + half * kStage) * ROW + r32 * ROW + 2 * h;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            c0[r] = 0.0f, c1[r] = 0.0f;
        constexpr int QB = 4;
        float2 fa0[2][QB], fa1[2][QB];
#pragma unroll
        for (int q = 0; q < QB; ++q) {
            fa0[0][q] = *reinterpret_cast<const float2 *>(pi + q * 4);
            fa1[0][q] = *reinterpret_cast<const float2 *>(pi + 32 * ROW + q * 4);
        }
#pragma unroll
        for (int g = 0; g < DQ / QB; ++g) {
            if (g + 1 < DQ / QB) {
#pragma unroll
                for (int q = 0; q < QB; ++q) {
                    fa0[(g + 1) & 1][q] = *reinterpret_cast<const float2 *>(pi + ((g + 1) * QB + q) * 4);
                    fa1[(g + 1) & 1][q] = *reinterpret_cast<const float2 *>(pi + 32 * ROW + ((g + 1) * QB + q) * 4);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                const float2 b2 = bf[g * QB + q];
                const int t0 = (g * QB + q) * 2;  // 2*DQ test slots per unit; one of every kSlotsPerReg is used
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[g & 1][q].x, b2.x, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[g & 1][q].x, b2.x, c1, 0, 0, 0);
                if constexpr (PREV) {
                    if (t0 % kSlotsPerReg == 0) {  // folded: the loops are fully unrolled
                        const int reg = t0 / kSlotsPerReg, r = reg & 15;
                        test(reg < 16 ? p0[r] : p1[r], s_prev + (reg < 16 ? 0 : 32) + (r & 3) + 8 * (r >> 2) + 4 * h);
                    }
                }
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[g & 1][q].y, b2.y, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[g & 1][q].y, b2.y, c1, 0, 0, 0);
                if constexpr (PREV) {
                    if ((t0 + 1) % kSlotsPerReg == 0) {
                        const int reg = (t0 + 1) / kSlotsPerReg, r = reg & 15;
                        test(reg < 16 ? p0[r] : p1[r], s_prev + (reg < 16 ? 0 : 32) + (r & 3) + 8 * (r >> 2) + 4 * h);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (LAST) {
            store_rows<DQ, ST>(smem + (buf ^ 1) * ST * ROW, nxt);
            __syncthreads();
            buf ^= 1;
        }
    };
    // the last unit of a split (possibly partial: rows past i_end) is tested after the loop
    auto test_last = [&](const f32x16 &p0, const f32x16 &p1, int s_prev) {
        const int lim = i_end - s_prev;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < lim)
                test(p0[r], s_prev + row);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < lim)
                test(p1[r], s_prev + row);
        }
    };

    // units alternate between the accumulator sets A and B; with H = 2 the A units open an LDS stage and the B units close it
    using T = std::true_type;
    using F = std::false_type;
    using OddFirst = std::integral_constant<bool, H == 1>;    // is a B unit the first of its stage?
    using EvenLast = std::integral_constant<bool, H == 1>;    // is an A unit the last of its stage?
    f32x16 A0, A1, B0, B1;
    int s0 = i_beg;
    unit(F{}, T{}, EvenLast{}, A0, A1, B0, B1, 0, s0, 0);
    for (;;) {
        int s_prev = s0;
        s0 += kStage;
        if (s0 >= i_end) {
            test_last(A0, A1, s_prev);
            break;
        }
        unit(T{}, OddFirst{}, T{}, B0, B1, A0, A1, s_prev, s0, H - 1);
        s_prev = s0;
        s0 += kStage;
        if (s0 >= i_end) {
            test_last(B0, B1, s_prev);
            break;
        }
        unit(T{}, T{}, EvenLast{}, A0, A1, B0, B1, s_prev, s0, 0);
    }
    TGCN_PROBE(2);
    if (user_ok)
        a.counts[((size_t)user * a.S + split) * 2 + h] = cnt;
}

// register budgets: d <= 64 fits 3 waves per SIMD (168 VGPRs), d <= 128 two.  A 2048-user call alone is 512 workgroups = two
// per CU (measured placement: 256 CUs x 2); the third slot is what lets the filter launches of consecutive calls on different
// streams overlap, and a SIMD with three waves keeps its matrix pipe busier than one with two (16 384-user calls: 0.60).
template <bool FULLK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_score_filter16(const FilterArgs a)
{
    filter_pipelined<16, FULLK, kStage>(a);
}

// the same loop with the registers of two waves per SIMD (no spills): a call of a few thousand users has two workgroups per CU
// whatever the budget, and without the 168-register squeeze it runs 0.62 instead of 0.60 T pairs/s end to end (2048 users,
// config 2); calls that fill the chip several times over keep the third wave (16 384 users: 0.69 against 0.67)
template <bool FULLK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_score_filter16_small(const FilterArgs a)
{
    filter_pipelined<16, FULLK, kStage>(a);
}

template <bool FULLK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_score_filter32(const FilterArgs a)
{
    filter_pipelined<32, FULLK, kStage>(a);
}

struct SelectArgs {
    const float2 *__restrict__ logs;
    const int *__restrict__ counts;
    const int *__restrict__ mask_rowptr;  // may be NULL
    const int *__restrict__ mask_items;
    float *__restrict__ out_val;
    int64_t *__restrict__ out_idx;
    int *__restrict__ flagged;  // [1 + B]: count, then the flagged user rows (count zeroed by the caller each call)
    int B, S, cap2, k, do_round;
    const int *__restrict__ totals;   // k_select_flat (k_rescore's output): user b's entries are logs[b * list_cap + 0 .. totals[b])
    int list_cap;                     // ... entries of log area per user
};

constexpr int kMaskCache = 512;  // train items per user cached in LDS for the membership test

// ---- threshold from the strided sample: one wave per user ---------------------------------------------------------------
// tau_u = the rank-th largest of the user's sampled scores after its sampled train items are dropped.  The row is
// staged in LDS (coalesced load, mask applied by the lanes that own the train items), read back VPL values per lane, and
// the maximum is extracted `rank` times (per-lane max, wave max, the owning lane retires one copy).  Replaces the
// k_mask + k_topk pair on the sample (28 us for 2048 x 1563 -> ~4 us) and zeroes the fallback counter for this call.
// Prefilter mode (ubound != NULL): the user's factors {n_u, r_u} of tgcn_score_prefilter.hip's bound are written too -- the row
// is fetched before the rounds and costs no extra launch.
template <int VPL>
__global__ __launch_bounds__(256) void k_tau(const float *__restrict__ Ss, int m_ld, int B, int m, const int *__restrict__ mask_rowptr,
                                             const int *__restrict__ mask_items, float *__restrict__ tau, int *__restrict__ flagged,
                                             int *__restrict__ done, const float *__restrict__ U, const int64_t *__restrict__ user_ids,
                                             int d, float *__restrict__ ubound, int *__restrict__ totals, int stride, int rank)
{
    extern __shared__ float srow[];   // [4][64 * VPL]
    const int lane = lane_id();
    const int w = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + w;
    // this call's fallback bookkeeping starts from zero: the flag count and one arrival counter per user (the workspace is
    // the caller's and arrives uninitialised); saves the memset node in front of every call
    if (blockIdx.x == 0 && threadIdx.x == 0)
        flagged[0] = 0;
    if (b < B && lane == 0) {
        done[b] = 0;
        if (totals)
            totals[b] = 0;      // k_rescore's per-user list lengths
    }
    float *row = srow + w * (kWave * VPL);     // private to the wave: wave-level barriers below
    const bool ok = b < B;
    int mb = 0, me = 0;
    if (ok && mask_rowptr) {                   // requested with the row, not after it
        mb = mask_rowptr[b];
        me = mask_rowptr[b + 1];
    }
    {   // the row, every load unconditional (clamped index): a `cond ? load : const` form makes hipcc branch around each load
        // and wait for it inside the branch -- VPL dependent round trips instead of one
        const float *__restrict__ src = Ss + (size_t)min(b, B - 1) * m_ld;
        float x[VPL];
#pragma unroll
        for (int i = 0; i < VPL; ++i)
            x[i] = src[min(lane + kWave * i, m - 1)];
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int j = lane + kWave * i;
            row[j] = (ok && j < m) ? x[i] : -INFINITY;
        }
    }
    float usq = 0.0f, ursq = 0.0f;
    if (ubound && ok) {
        const float *__restrict__ p = U + (size_t)(user_ids ? user_ids[b] : b) * d;
        for (int k = lane; k < d; k += kWave) {
            usq += floored_sq(p[k]);
            ursq += residual_sq(p[k]);
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    {
        for (int e = mb + lane; e < me; e += kWave) {
            const int it = mask_items[e];
            if (it % stride == 0 && it / stride < m)
                row[it / stride] = -INFINITY;
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    // the lane's two largest values, then `rank` rounds of (wave maximum, its first owner retires one copy).  A lane that
    // holds three or more of the row's `rank` largest gives a slightly lower tau than the exact rank: tau is only a bar that
    // at least k items must clear (k_select checks that), not a result.
    float a0 = -INFINITY, a1 = -INFINITY;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const float x = row[lane + kWave * i];
        a1 = fmaxf(a1, fminf(a0, x));
        a0 = fmaxf(a0, x);
    }
    float t = -INFINITY;
    for (int r = 0; r < rank; ++r) {
        t = wave_max_all(a0);
        const int owner = __ffsll((long long)__ballot(a0 == t)) - 1;
        if (lane == owner) {
            a0 = a1;
            a1 = -INFINITY;
        }
    }
    if (ubound) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            usq += __shfl_xor(usq, o);
            ursq += __shfl_xor(ursq, o);
        }
    }
    if (ok && lane == 0) {
        tau[b] = t;
        if (ubound)
            *reinterpret_cast<float2 *>(ubound + 2 * (size_t)b) = make_float2(bound_factor(usq), bound_factor(ursq));
    }
}

// ---- exact selection from the logs ----------------------------------------------------------------------------------------
// One wave per user.  (1) lane = log segment: counts -> exclusive prefix -> the segments' entries are copied, compacted, into
// an LDS array (n ~ 320 of them); (2) train items are dropped (binary search in the user's sorted list); (3) the k-th
// largest value is found by a bitwise binary search over order-preserving uint keys (32 steps of VPL ballots -- no serial
// insertion chain); (4) the k winners ((value desc, item asc); ties at the k-th value are taken by ascending item) are packed
// into lanes 0..k-1 and sorted by a 64-lane bitonic network.  A user with fewer than k unmasked candidates, an overflowed
// log or more than kSelCap candidates is flagged for the exact fallback.
template <int VPL>
__device__ __forceinline__ void select_from_lds(const float2 *__restrict__ cand, int n, int k, int lane, float &out_v, int &out_i,
                                                float2 *__restrict__ pack)
{
    unsigned key[VPL];
    int idx[VPL];
#pragma unroll
    for (int s = 0; s < VPL; ++s) {
        const int j = lane + kWave * s;
        const float2 t = j < n ? cand[j] : make_float2(-INFINITY, __int_as_float(INT_MAX));
        idx[s] = __float_as_int(t.y);
        key[s] = idx[s] == INT_MAX ? 0u : ordered_key(t.x);      // dropped / padding: below every real score (key >= 1)
    }
    select_core<VPL>(key, idx, k, lane, out_v, out_i, pack);
}

// ---- a second, tight threshold from the LOGGED approximate scores (wide rows: tgcn_score_prefilter.hip logs them) ----------
// The wide bf16 filter leaves, per user, the pairs with approx + bound > tau_u and their raised scores R_i = approx_i + bound_i.
// bound_i is recomputed here from the same bf16 factors, so L_i = R_i - 2 bound_i (less a rounding margin) <= score_i <= R_i.
// tau2 = the k-th largest L_i among the user's unmasked candidates: at least k unmasked items score >= tau2, so an item with
// R_i < tau2 has k items strictly above it and cannot be in the top k -- it is dropped WITHOUT an fp32 chain.  What survives
// (the top k and whatever lies within the error band of the k-th score: ~70 of ~350 at K = 960) goes to k_rescore as an id list;
// from there the path is the narrow rows' (scores > tau_u kept, k_select_flat, fallback).  Non-finite R or bound: survives,
// never counted towards k.  One wave per user; train items are dropped here already.
struct RefineArgs {
    const float2 *__restrict__ logs;   // [B][2 S][cap2]  (S = twice the wide filter's splits: 4 segments per split)
    const int *__restrict__ counts;    // [B][2 S]
    const int *__restrict__ mask_rowptr;
    const int *__restrict__ mask_items;
    const float *__restrict__ ubound;  // [B][2] {n_u, r_u}
    const unsigned char *__restrict__ ipack;
    int row_bytes;                     // packed item row; the factor chunk [r_i, n_i + r_i, n_i] (bf16) is its last 16 bytes
    int *__restrict__ surv;            // [B][surv_cap]
    int *__restrict__ surv_n;          // [B]; kRefineOverflow: the user takes the exact fallback
    int surv_cap;
    int B, S, cap2, k;
};
constexpr int kRefineOverflow = 1 << 20;
constexpr int kRefineCap = 2048;      // logged candidates per user k_refine takes (more: exact fallback)

__global__ __launch_bounds__(256) void k_refine(const RefineArgs a)
{
    __shared__ float2 scand[4][kRefineCap];
    __shared__ int smask[4][kMaskCache];
    const int lane = lane_id();
    const int w = threadIdx.x >> 6;
    const int b = uniform(blockIdx.x * 4 + w);
    if (b >= a.B)
        return;
    int mb = 0, me = 0;
    if (a.mask_rowptr) {
        mb = a.mask_rowptr[b];
        me = a.mask_rowptr[b + 1];
    }
    const bool cached = (me - mb) <= kMaskCache;
    const int n_seg = a.S * 2;   // <= 64: one segment per lane
    int cnt = lane < n_seg ? a.counts[(size_t)b * n_seg + lane] : 0;
    bool overflow = __any(cnt > a.cap2);
    cnt = min(cnt, a.cap2);
    int off = cnt;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const int t = __shfl_up(off, o);
        if (lane >= o)
            off += t;
    }
    const int n = __builtin_amdgcn_readlane(off, kWave - 1);
    off -= cnt;
    if (overflow || n > min(kRefineCap, a.surv_cap)) {
        if (lane == 0)
            a.surv_n[b] = kRefineOverflow;
        return;
    }
    float2 *cand = scand[w];
    {
        const float2 *__restrict__ lg = a.logs + ((size_t)b * n_seg + min(lane, n_seg - 1)) * a.cap2;
        int longest = cnt;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            longest = max(longest, __shfl_xor(longest, o));
        for (int j0 = 0; j0 < longest; j0 += 8) {
            float2 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                t[u] = lg[min(j0 + u, a.cap2 - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (j0 + u < cnt)
                    cand[off + j0 + u] = t[u];
        }
    }
    if (cached)
        for (int j = lane; j < me - mb; j += kWave)
            smask[w][j] = a.mask_items[mb + j];
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    // the user's factors as the filter's bound step saw them (rounded up to bf16)
    const float2 ub = *reinterpret_cast<const float2 *>(a.ubound + 2 * (size_t)b);
    const float nu = __uint_as_float(bf16_up_bits(ub.x) << 16), ru = __uint_as_float(bf16_up_bits(ub.y) << 16);
    const float nb = __uint_as_float(bf16_up_bits(ub.x * kAccumBudget) << 16);
    auto run = [&](auto vpl_tag) {
        constexpr int VPL = decltype(vpl_tag)::value;
        unsigned keyL[VPL], keyR[VPL];
        int idx[VPL];
        bool live[VPL];
        uint2 fac[VPL];
#pragma unroll
        for (int t = 0; t < VPL; ++t) {
            const float2 c = cand[min(lane + kWave * t, n - 1)];
            idx[t] = __float_as_int(c.y);
            keyR[t] = __float_as_uint(c.x);      // (the raw score for now)
            fac[t] = *reinterpret_cast<const uint2 *>(a.ipack + (size_t)idx[t] * a.row_bytes + (a.row_bytes - 16));
        }
        int n_valid = 0;
#pragma unroll
        for (int t = 0; t < VPL; ++t) {
            live[t] = lane + kWave * t < n &&
                      !(cached ? sorted_contains(smask[w], 0, me - mb, idx[t]) : sorted_contains(a.mask_items, mb, me, idx[t]));
            const float R = __uint_as_float(keyR[t]);
            const float ri = __uint_as_float(fac[t].x << 16), nri = __uint_as_float(fac[t].x & 0xFFFF0000u), ni = __uint_as_float(fac[t].y << 16);
            const float bound = (nu * ri + ru * nri + nb * ni) * (1.0f + 0x1p-20f);
            const float L = R - 2.0f * bound - 0x1p-21f * (fabsf(R) + bound);
            const bool fin = fabsf(R) < INFINITY && bound < INFINITY;      // (false for NaN too)
            keyL[t] = live[t] && fin ? ordered_key(L) : 0u;                 // 0: does not count towards k
            keyR[t] = fin ? ordered_key(R) : 0xFFFFFFFFu;                   // non-finite: survives whatever tau2 is
            n_valid += __popcll(__ballot(live[t] && fin));
        }
        // T = the k-th largest keyL (0 when fewer than k count: everything survives and the later stages decide)
        unsigned T = 0;
        if (n_valid >= a.k) {
            for (int bit = 31; bit >= 0; --bit) {
                const unsigned c = T | (1u << bit);
                int cn = 0;
#pragma unroll
                for (int t = 0; t < VPL; ++t)
                    cn += __popcll(__ballot(keyL[t] >= c));
                if (cn >= a.k)
                    T = c;
            }
        }
        int base = 0;
        const unsigned long long lt = (1ull << lane) - 1ull;
        int *__restrict__ out = a.surv + (size_t)b * a.surv_cap;
#pragma unroll
        for (int t = 0; t < VPL; ++t) {
            const bool take = live[t] && keyR[t] >= T;
            const unsigned long long m = __ballot(take);
            if (take)
                out[base + __popcll(m & lt)] = idx[t];
            base += __popcll(m);
        }
        if (lane == 0)
            a.surv_n[b] = base;
    };
    if (n == 0) {
        if (lane == 0)
            a.surv_n[b] = 0;
    } else if (n <= 4 * kWave)
        run(std::integral_constant<int, 4>{});
    else if (n <= 8 * kWave)
        run(std::integral_constant<int, 8>{});
    else if (n <= 16 * kWave)
        run(std::integral_constant<int, 16>{});
    else
        run(std::integral_constant<int, kRefineCap / kWave>{});
}

__global__ __launch_bounds__(256) void k_select(const SelectArgs a)
{
    __shared__ float2 scand[4][kSelCap];
    __shared__ float2 spack[4][kWave];
    __shared__ int smask[4][kMaskCache];
    const int lane = lane_id();
    const int w = threadIdx.x >> 6;
    const int b = uniform(blockIdx.x * 4 + w);
    if (b >= a.B)
        return;
    int mb = 0, me = 0;
    if (a.mask_rowptr) {
        mb = a.mask_rowptr[b];
        me = a.mask_rowptr[b + 1];
    }
    const bool cached = (me - mb) <= kMaskCache;
    // the first 64 train items are requested here and stored after the candidate loads have been issued too: one round trip
    // for both instead of two
    const int m_first = (cached && lane < me - mb) ? a.mask_items[mb + lane] : 0;
    const int n_seg = a.S * 2;   // <= 64: one segment per lane
    int cnt = lane < n_seg ? a.counts[(size_t)b * n_seg + lane] : 0;
    bool overflow = __any(cnt > a.cap2);
    cnt = min(cnt, a.cap2);
    // exclusive prefix of the counts over lanes
    int off = cnt;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const int t = __shfl_up(off, o);
        if (lane >= o)
            off += t;
    }
    const int n = __builtin_amdgcn_readlane(off, kWave - 1);
    off -= cnt;
    if (n > kSelCap)
        overflow = true;
    bool ok = !overflow;
    float out_v = -INFINITY;
    int out_i = INT_MAX;
    if (ok) {
        float2 *cand = scand[w];
        {
            const float2 *__restrict__ lg = a.logs + ((size_t)b * n_seg + min(lane, n_seg - 1)) * a.cap2;
            int longest = cnt;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1)
                longest = max(longest, __shfl_xor(longest, o));
            for (int j0 = 0; j0 < longest; j0 += 8) {
                float2 t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    t[u] = lg[min(j0 + u, a.cap2 - 1)];   // in bounds; entries past cnt are ignored
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (j0 + u < cnt)
                        cand[off + j0 + u] = t[u];
            }
        }
        if (cached) {
            if (lane < me - mb)
                smask[w][lane] = m_first;
            for (int j = kWave + lane; j < me - mb; j += kWave)
                smask[w][j] = a.mask_items[mb + j];
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        // drop train items (base_model.py:257-258 sets them to -inf); count what is left
        int n_valid = 0;
        for (int j0 = 0; j0 < n; j0 += kWave) {
            const int j = j0 + lane;
            bool on = j < n;
            if (on) {
                const int si = __float_as_int(cand[j].y);
                if (cached ? sorted_contains(smask[w], 0, me - mb, si) : sorted_contains(a.mask_items, mb, me, si)) {
                    cand[j].y = __int_as_float(INT_MAX);
                    on = false;
                }
            }
            n_valid += __popcll(__ballot(on));
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        ok = n_valid >= a.k;
        if (ok) {
            if (n <= 4 * kWave)
                select_from_lds<4>(cand, n, a.k, lane, out_v, out_i, spack[w]);
            else if (n <= 8 * kWave)
                select_from_lds<8>(cand, n, a.k, lane, out_v, out_i, spack[w]);
            else
                select_from_lds<kSelVPL>(cand, n, a.k, lane, out_v, out_i, spack[w]);
        }
    }
    if (!ok && lane == 0)
        a.flagged[1 + atomicAdd(a.flagged, 1)] = b;
    if (ok && lane < a.k) {
        a.out_val[(size_t)b * a.k + lane] = a.do_round ? round4(out_v) : out_v;
        a.out_idx[(size_t)b * a.k + lane] = out_i;
    }
}

// k_select for k_rescore's flat lists (totals[b] entries at logs[b * S * 2 * cap2 ...]): the candidates go from global memory
// straight into registers (16 per lane at most) -- no 8 KB LDS staging array per wave, so the launch is not held to three
// workgroups per CU by it -- and the train items are dropped there.
__global__ __launch_bounds__(256) void k_select_flat(const SelectArgs a)
{
    __shared__ float2 spack[4][kWave];
    __shared__ int smask[4][kMaskCache];
    const int lane = lane_id();
    const int w = threadIdx.x >> 6;
    const int b = uniform(blockIdx.x * 4 + w);
    if (b >= a.B)
        return;
    int mb = 0, me = 0;
    if (a.mask_rowptr) {
        mb = a.mask_rowptr[b];
        me = a.mask_rowptr[b + 1];
    }
    const bool cached = (me - mb) <= kMaskCache;
    const int m_first = (cached && lane < me - mb) ? a.mask_items[mb + lane] : 0;
    const int list_cap = a.list_cap;
    const int n = a.totals[b];
    bool ok = n <= min(kSelCap, list_cap) && n > 0;
    float out_v = -INFINITY;
    int out_i = INT_MAX;
    if (ok) {
        const float2 *__restrict__ lg = a.logs + (size_t)b * list_cap;
        auto run = [&](auto vpl_tag) {
            constexpr int VPL = decltype(vpl_tag)::value;
            float2 t[VPL];
#pragma unroll
            for (int s = 0; s < VPL; ++s)
                t[s] = lg[min(lane + kWave * s, n - 1)];       // unconditional (clamped) loads, all in flight
            if (cached) {
                if (lane < me - mb)
                    smask[w][lane] = m_first;
                for (int j = kWave + lane; j < me - mb; j += kWave)
                    smask[w][j] = a.mask_items[mb + j];
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
            unsigned key[VPL];
            int idx[VPL];
            int n_valid = 0;
#pragma unroll
            for (int s = 0; s < VPL; ++s) {
                const int si = __float_as_int(t[s].y);
                bool on = lane + kWave * s < n;
                if (on && (cached ? sorted_contains(smask[w], 0, me - mb, si) : sorted_contains(a.mask_items, mb, me, si)))
                    on = false;     // a train item (base_model.py:257-258 sets them to -inf)
                idx[s] = on ? si : INT_MAX;
                key[s] = on ? ordered_key(t[s].x) : 0u;
                n_valid += __popcll(__ballot(on));
            }
            ok = n_valid >= a.k;
            if (ok)
                select_core<VPL>(key, idx, a.k, lane, out_v, out_i, spack[w]);
        };
        if (n <= 4 * kWave)
            run(std::integral_constant<int, 4>{});
        else if (n <= 8 * kWave)
            run(std::integral_constant<int, 8>{});
        else
            run(std::integral_constant<int, kSelVPL>{});
    }
    if (!ok && lane == 0)
        a.flagged[1 + atomicAdd(a.flagged, 1)] = b;
    if (ok && lane < a.k) {
        a.out_val[(size_t)b * a.k + lane] = a.do_round ? round4(out_v) : out_v;
        a.out_idx[(size_t)b * a.k + lane] = out_i;
    }
}

struct BruteArgs {
    const float *__restrict__ U;
    const int64_t *__restrict__ user_ids;
    const float *__restrict__ It;
    const int *__restrict__ mask_rowptr;
    const int *__restrict__ mask_items;
    const int *__restrict__ flagged;  // [1 + B]: count, rows
    float2 *parts;                   // [flag_cap][kBruteSplits][64] partial lists
    int *done;                       // [flag_cap] arrival counters (zero on entry; the merging workgroup re-zeroes its own)
    int flag_cap;
    float *__restrict__ out_val;
    int64_t *__restrict__ out_idx;
    int B, I, d, k, do_round;
};

// Exact fallback for flagged users, spread over the chip: kBruteSplits workgroups per user, each scanning a
// contiguous 1/kBruteSplits of the items.  A wave walks its share in tiles of 64 rows: the tile is loaded with
// coalesced 16-byte reads into a padded LDS buffer (row stride d+1 floats: lane = row reads are conflict-free),
// then lane l chains row l in ascending k -- the same fmaf chain as the MFMA.  The workgroup's waves merge their
// lists through LDS and write one 64-entry partial list per (user, split); k_brute_merge finishes the job.
// A flagged user costs ~25 us of latency instead of ~1 ms in a single workgroup.
constexpr int kBruteSplits = 32;
constexpr int kBruteWaves = 4;
constexpr int kBruteTileMaxD = 32;    // floats of a row per pass through the LDS tile (4 waves x 64 rows x 33 floats = 34 KB).  With
                                      // whole rows in the tile (132 KB at d = 128) every workgroup of this launch -- 256 of them,
                                      // every call, almost always with nothing to do -- needed a CU's whole LDS to itself and
                                      // waited for the other streams' workgroups to drain from one.

__global__ __launch_bounds__(kBruteWaves * 64) void k_brute_part(const BruteArgs a)
{
    extern __shared__ float su[];  // user row [d_pad] | lists 2 x [waves x 64] | tiles [waves][64][d+1]
    const int lane = lane_id();
    const int w = threadIdx.x >> 6;
    const int d_pad = (a.d + 63) & ~63;
    float *lv = su + d_pad;
    int *li = reinterpret_cast<int *>(lv + kBruteWaves * kWave);
    const bool tiled = (a.d & 3) == 0;
    const int trow = min(a.d, kBruteTileMaxD) + 1;
    float *tile = reinterpret_cast<float *>(li + kBruteWaves * kWave) + (size_t)w * kWave * trow;
    const int split = blockIdx.x;
    const int n_flagged = min(a.flagged[0], a.flag_cap);
    // this workgroup's item range, then this wave's share of it (multiples of 64)
    const int per_split = (((a.I + kBruteSplits - 1) / kBruteSplits + kWave - 1) / kWave) * kWave;
    const int s_beg = min(a.I, split * per_split), s_end = min(a.I, s_beg + per_split);
    const int per = ((((s_end - s_beg) + kBruteWaves - 1) / kBruteWaves + kWave - 1) / kWave) * kWave;
    const int beg = min(s_end, s_beg + w * per), end = min(s_end, beg + per);
    for (int f = blockIdx.y; f < n_flagged; f += gridDim.y) {
        const int b = a.flagged[1 + f];
        __syncthreads();  // su / lists reused across iterations
        const float *__restrict__ urow = a.U + (size_t)(a.user_ids ? a.user_ids[b] : b) * a.d;
        for (int k = threadIdx.x; k < a.d; k += blockDim.x)
            su[k] = urow[k];
        __syncthreads();
        int mb = 0, me = 0;
        if (a.mask_rowptr) {
            mb = a.mask_rowptr[b];
            me = a.mask_rowptr[b + 1];
        }
        TopList e{-INFINITY, INT_MAX};
        for (int i0 = beg; i0 < end; i0 += kWave) {
            const int item = i0 + lane;
            const bool on = item < end;
            float s = 0.0f;
            if (tiled) {
                // coalesced: consecutive lanes read consecutive 16-byte pieces of the 64 x kw block of the tile's rows
                const int n_rows = min(kWave, end - i0);
                const float *__restrict__ src = a.It + (size_t)i0 * a.d;
                for (int kb = 0; kb < a.d; kb += kBruteTileMaxD) {
                    const int kw = min(kBruteTileMaxD, a.d - kb), qb = kw >> 2;
                    __builtin_amdgcn_wave_barrier();   // the previous block's reads of the tile are done (same wave, in order)
                    for (int p = lane; p < n_rows * qb; p += kWave) {
                        const int r = p / qb, k = (p % qb) * 4;
                        const float4 t = *reinterpret_cast<const float4 *>(src + (size_t)r * a.d + kb + k);
                        float *o = tile + r * trow + k;
                        o[0] = t.x, o[1] = t.y, o[2] = t.z, o[3] = t.w;
                    }
                    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's LDS writes before its reads
                    __builtin_amdgcn_wave_barrier();
                    const float *__restrict__ row = tile + lane * trow;
                    if (on)
                        for (int k = 0; k < kw; ++k)
                            s = fmaf(su[kb + k], row[k], s);
                }
            } else if (on) {
                const float *__restrict__ p = a.It + (size_t)item * a.d;
                for (int k = 0; k < a.d; ++k)
                    s = fmaf(su[k], p[k], s);
            }
            if (on && sorted_contains(a.mask_items, mb, me, item))
                s = -INFINITY;
            list_offer(e, s, item, on, a.k, lane);
        }
        lv[w * kWave + lane] = e.v;
        li[w * kWave + lane] = e.i;
        __syncthreads();
        if (w == 0) {
            for (int o = 1; o < kBruteWaves; ++o)
                list_offer(e, lv[o * kWave + lane], li[o * kWave + lane], lane < a.k && li[o * kWave + lane] != INT_MAX, a.k, lane);
            a.parts[((size_t)f * kBruteSplits + split) * kWave + lane] = make_float2(e.v, __int_as_float(e.i));
            // The workgroup whose arrival is the last of this user's kBruteSplits merges the partial lists (one launch
            // instead of part + merge).  Hand-off: plain stores by this one wave -> agent release -> drained -> agent atomic
            // add; the last arriver (told by the value its add returned) -> agent acquire -> plain loads
            // (MI355X_MICROARCH.md, inter-workgroup visibility).  Placement-independent; no spin anywhere.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            int ticket = 0;
            if (lane == 0)
                ticket = atomicAdd(a.done + f, 1);
            ticket = __builtin_amdgcn_readfirstlane(ticket);
            if (ticket == kBruteSplits - 1) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                TopList t{-INFINITY, INT_MAX};
                for (int sidx = 0; sidx < kBruteSplits; ++sidx) {
                    const float2 q = a.parts[((size_t)f * kBruteSplits + sidx) * kWave + lane];
                    const int qi = __float_as_int(q.y);
                    list_offer(t, q.x, qi, lane < a.k && qi != INT_MAX, a.k, lane);
                }
                if (lane < a.k) {
                    a.out_val[(size_t)b * a.k + lane] = a.do_round ? round4(t.v) : t.v;
                    a.out_idx[(size_t)b * a.k + lane] = t.i;
                }
                if (lane == 0)
                    a.done[f] = 0;
            }
        }
    }
}

struct Plan {
    int S, items_per_split, cap2, m, m_ld;  // m = sampled items
    int stride, rank;                       // the sample: every stride-th item; tau = the rank-th largest sampled score
    int m_rank;                             // values per user k_tau ranks: m, or (top form) the two largest of every 128-sample block
    bool top;                               // d <= 128: the sample's scores stay inside the sample kernel (launch_sample_top)
    size_t off_bits;                        // top form: train-item bitmap over the sample, [B][4 ceil(m / 128)] words
    int S_w, ips_w, cap2_w;                 // wide bf16 filter: its own (at most 16) item splits -- 4 log segments of cap2_w per split and user
    int Wh;                                 // mask words per (user, row half): the 64-item units of the catalogue, padded to 4
    size_t off_mask;
    size_t off_summ;                        // the narrow bf16 filter's stage summary (PassSummary), summ_words per (user, row half); 0: none
    int summ_words;
    size_t off_sample, off_tauv, off_taui, off_tau, off_ubound, off_ipack, off_logs, off_counts, off_parts, off_flags, off_done, off_totals, off_surv, off_surv_n, off_stats, total;
    int n_seg;                              // log segments per user (2 S; the wide bf16 filter: 4 S_w)
    int flag_cap;
    bool small;
};

size_t align256(size_t x) { return (x + 255) & ~size_t(255); }

// The bar tau_u must not land above the user's k-th best score, or the user pays the exact fallback: that takes `rank` of its best
// k - 1 items inside the sample, P = P(Bin(k - 1, 1 / stride) >= rank).  The smallest rank with P <= kBarRisk; the candidates
// a user then brings to the filter's logs / the rescoring number ~rank x stride.
int bar_rank(int k, int stride)
{
    const int n = max(k - 1, 0);
    const double q = 1.0 / stride;
    for (int r = 4; r < 64; ++r) {
        double pmf = 1.0, tail = 0.0;       // pmf(j) = C(n, j) q^j (1 - q)^(n - j), built up from j = 0
        for (int j = 0; j < n; ++j)
            pmf *= 1.0 - q;
        for (int j = 0; j <= n; ++j) {
            if (j >= r)
                tail += pmf;
            pmf *= (double)(n - j) / (j + 1) * q / (1.0 - q);
        }
        if (tail <= kBarRisk)
            return r;
    }
    return 64;
}

// The threshold sample.  A sample twice as dense halves the distance between the bar and the k-th score: k = 40 -> rank 10 of every
// 32nd item = ~320 candidates per user, rank 12 of every 16th = ~190, rank 17 of every 8th = ~140 -- the fp32 rescoring of the bf16
// path (a third of a narrow call, at the random-row rate) and the filters' appends shrink with them.  Narrow rows only: their sample
// runs on the bf16 pipe and keeps its scores to itself (the TOP forms: two values per user and 128-sample block reach k_tau), so a
// denser sample costs MFMAs, not traffic.  The prefiltered entry point reads the sampled rows from the pack (k_sample_pack_top,
// 19 us per 16 384 x 6250 x 64): every 8th item; the fp32 entry point converts them itself (k_sample_bf16, twice the time): every
// 16th.  Catalogues whose bitmap row would not fit k_sample_bits' LDS at that density (above 1.0 / 2.1 M items) take every 16th /
// 32nd / 64th item, still in the TOP form: with ALL sample scores written and ranked by k_mask + k_topk (round 3's form for them) a
// 2048-user call on config 4's 2 M items spent 118 + 6 + 178 us of its 1063 on the bar -- the sample pass was a 512 MB store
// (profiles/r04_experiments.md section 9).  At K = 960 the sample is a tenth of the filter and k_refine's second bar decides what
// is rescored: every 32nd item, all scores to k_tau, as before; so do catalogues too small for the blocks to resolve the rank
// (below ~35 000 items at k = 40) and beyond 8 M items.
void set_sample(Plan &p, int I, int d, int k, bool from_pack)
{
    // the densest TOP form whose blocks still resolve the rank: k_tau sees two values per block of 128 sampled items, and with fewer
    // than ~2 rank blocks several of a user's `rank` best share a block (the bar then sits ranks lower than asked for -- at 9 blocks
    // and rank 22 it would fall off the end: tau = -inf, every user to the exact fallback)
    for (int dense = from_pack ? 8 : 16; dense <= 64; dense *= 2) {
        const int m = (I + dense - 1) / dense, n_blk = (m + 127) / 128, rank = bar_rank(k, dense);
        if (sample_top_supports(d, m) && n_blk >= 2 * rank) {
            p.top = true, p.stride = dense, p.rank = rank, p.m = m, p.m_rank = 2 * n_blk;
            p.m_ld = (p.m_rank + 3) & ~3;  // row stride of the sample score matrix (16-byte aligned rows)
            return;
        }
    }
    p.top = false, p.stride = kSampleStride, p.rank = bar_rank(k, p.stride);
    p.m = (I + p.stride - 1) / p.stride, p.m_rank = p.m;
    p.m_ld = (p.m_rank + 3) & ~3;
}

Plan make_plan(int B, int I, int d, int k)
{
    Plan p{};
    p.small = I <= kSmallI;
    if (p.small) {
        p.off_sample = 0;  // the [B, I] score matrix
        p.total = align256((size_t)B * I * sizeof(float));
        return p;
    }
    // 32 item splits (512 workgroups for one 2048-user call, 2 per CU).  Larger calls keep 32: measured on config 2
    // with 16 384 users per call, 8 / 12 / 24 / 32 / 48 splits -> 0.60 / 0.58 / 0.65 / 0.66 / 0.65 T pairs/s (many short
    // workgroups keep the chip evenly loaded); 2048 users: 32 and 48 splits tie.
    const int max_S = max(1, I / (8 * kStage));   // at least 8 stages per split
    const int S = min(32, max_S);
    const int gran = d > 256 ? 128 : kStage;   // wide operands walk 128-item tiles (k_score_dense_filter)
    p.items_per_split = (((I + S - 1) / S + gran - 1) / gran) * gran;
    p.S = (I + p.items_per_split - 1) / p.items_per_split;
    p.cap2 = max(32, 1024 / (2 * p.S));
    if (d > 128 && prefilter_supports(d))
        p.cap2 = max(p.cap2, 64);      // the wide bf16 filter logs its (more numerous) raised candidates in these segments
    // the threshold sample of either entry point (set_sample); the regions hold the larger of the two
    Plan q = p;
    set_sample(p, I, d, k, true);
    set_sample(q, I, d, k, false);
    size_t o = 0;
    p.off_sample = o, o += align256((size_t)B * max(p.m_ld, q.m_ld) * sizeof(float));
    p.off_bits = o;
    if (p.top || q.top)
        o += align256((size_t)B * 4 * ((max(p.top ? p.m : 0, q.top ? q.m : 0) + 127) / 128) * sizeof(unsigned));
    p.off_tauv = o, o += align256((size_t)B * max(p.rank, q.rank) * sizeof(float));
    p.off_taui = o, o += align256((size_t)B * max(p.rank, q.rank) * sizeof(int64_t));
    p.off_tau = o, o += align256((size_t)B * sizeof(float));
    p.off_ubound = o, o += align256((size_t)B * 2 * sizeof(float));         // prefilter mode: the users' factors of the bound
    p.off_ipack = o;                                                       // ... and the packed item operand (unless the caller
    o += align256(item_pack_bytes(I, d));                                  //     holds it: tgcn_item_pack_bf16)
    p.Wh = ((I + kStage - 1) / kStage + 3) & ~3;   // words per row half, padded: every half starts on a 16-byte boundary
    p.off_mask = o;
    if (prefilter_supports(d))
        o += align256((size_t)((B + 255) / 256) * 256 * 2 * p.Wh * sizeof(unsigned));   // rows of the bf16 filter's padded users
    // ... and, for large catalogues, their stage summary (PassSummary): at most ceil(stages / 32) + one word per split, per row half
    p.off_summ = o, p.summ_words = 0;
    if (d <= 128 && prefilter_supports(d) && p.Wh >= kSummaryMinWords) {
        p.summ_words = (p.Wh * kStage / prefilter_stage_items(d, false) + 31) / 32 + 40;
        o += align256((size_t)((B + 255) / 256) * 256 * 2 * p.summ_words * sizeof(unsigned));
    }
    // the wide bf16 filter (K split between two waves per SIMD) logs into 4 segments per (user, split) -- (tile, row half) of the
    // lane that finished the pair -- and k_refine reads one segment per lane: at most 16 splits.  One 512-thread workgroup per CU
    // (256 at a time); a workgroup pays a prologue (its users' fragments, ~35 us by the cycle stamps = ~9 units of 64 items) and
    // then its split's units, so a call costs generations x (prologue + units / splits).  8192 users: 64 tiles x 4 splits fill the
    // chip once; 12 288 users: 96 x 2 would leave a quarter of the CUs idle and 96 x 3 a second generation of 32 -- 96 x 8 = three
    // full generations is 20 % shorter (profiles/r04_c5_call_sweep.jsonl).
    {
        const int tiles = (B + 127) / 128;
        const int units = (I + kStage - 1) / kStage;
        constexpr int kPrologueUnits = 9;
        int sw = 1;
        long best = -1;
        for (int c = 1; c <= 16 && c <= max(1, units / 8); ++c) {         // at least 8 units per split
            const long gens = ((long)tiles * c + 255) / 256;
            const long cost = gens * (kPrologueUnits + (units + c - 1) / c);
            if (best < 0 || cost < best)
                best = cost, sw = c;
        }
        p.ips_w = ((units + sw - 1) / sw) * kStage;
        p.S_w = (I + p.ips_w - 1) / p.ips_w;
        p.cap2_w = max(p.cap2, (64 * 64) / (4 * p.S_w));       // 4096 log entries per user in all (~550 are used at K = 960)
    }
    p.n_seg = 2 * p.S;
    if (d > 128 && prefilter_supports(d))      // the log area in units of cap2: both layouts fit
        p.n_seg = max(p.n_seg, (4 * p.S_w * p.cap2_w + p.cap2 - 1) / p.cap2);
    p.off_logs = o, o += align256((size_t)B * p.n_seg * p.cap2 * sizeof(float2));
    p.off_counts = o, o += align256((size_t)B * p.n_seg * sizeof(int));
    p.flag_cap = B;   // every user may need the fallback (e.g. fully tied scores): 16 KB of partial lists each
    p.off_parts = o, o += align256((size_t)p.flag_cap * kBruteSplits * kWave * sizeof(float2));
    p.off_flags = o, o += align256((size_t)(B + 1) * sizeof(int));
    p.off_done = o, o += align256((size_t)p.flag_cap * sizeof(int));   // contiguous with the flags: one memset covers both
    p.off_totals = o, o += align256((size_t)B * sizeof(int));          // ... and k_rescore's list lengths
    p.off_surv = p.off_surv_n = o;                                     // wide rows: k_refine's surviving candidates
    if (d > 128 && prefilter_supports(d)) {
        o += align256((size_t)B * kRefineCap * sizeof(int));
        p.off_surv_n = o, o += align256((size_t)B * sizeof(int));
    }
    p.off_stats = o, o += 256;                                         // tgcn_score_topk_stats' four counters
    p.total = o;
    return p;
}

// tgcn_score_topk_stats: sums over the users of the last call.  One wave per 64 users; wave-reduced adds into four counters.
struct StatsArgs {
    const int *__restrict__ flagged;   // [0] = users sent to the exact fallback
    const int *__restrict__ totals;    // [B] pairs kept by k_rescore (score > tau), or NULL
    const unsigned *__restrict__ mask; // narrow bf16 filter: pass bits [B padded][2][Wh], or NULL
    int Wh, n_units;
    const int *__restrict__ surv_n;    // wide bf16 filter: candidates k_refine kept [B], or NULL
    const int *__restrict__ counts;    // logged candidates [B][n_seg] (fp32 filters and the wide bf16 filter), or NULL
    int n_seg, cap2, B;
    unsigned long long *__restrict__ out;   // [4]: fallback users, kept pairs, candidates rescored in fp32, logged pairs
};

__global__ __launch_bounds__(256) void k_stats(const StatsArgs a)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    unsigned long long kept = 0, cand = 0, logged = 0;
    if (b < a.B) {
        if (a.totals)
            kept = (unsigned long long)min(a.totals[b], 1 << 20);
        if (a.mask) {
            const unsigned *__restrict__ row = a.mask + (size_t)b * 2 * a.Wh;
            for (int hh = 0; hh < 2; ++hh)
                for (int j = 0; j < a.n_units; ++j)
                    cand += __popc(row[hh * a.Wh + j]);
        }
        if (a.surv_n)
            cand = (unsigned long long)min(a.surv_n[b], 1 << 20);
        if (a.counts)
            for (int j = 0; j < a.n_seg; ++j)
                logged += (unsigned long long)min(a.counts[(size_t)b * a.n_seg + j], a.cap2);
    }
    for (int o = 32; o > 0; o >>= 1) {
        kept += __shfl_xor(kept, o);
        cand += __shfl_xor(cand, o);
        logged += __shfl_xor(logged, o);
    }
    if (lane_id() == 0) {
        atomicAdd(a.out + 1, kept);
        atomicAdd(a.out + 2, cand);
        atomicAdd(a.out + 3, logged);
        if (b == 0)
            a.out[0] = (unsigned long long)a.flagged[0];
    }
}

// > 64 KB of dynamic LDS needs hipFuncAttributeMaxDynamicSharedMemorySize on the kernel, per device (the attribute lives
// with the device's copy of the code object).  Set once per device, thread-safe, and the outcome is kept: a device where
// it failed reports the error on every call instead of faulting in the launch.
int brute_lds_opt_in()
{
    constexpr int kMaxDevices = 64;
    static std::once_flag once[kMaxDevices];
    static hipError_t result[kMaxDevices];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) {
        set_error("hipGetDevice failed or device index >= %d", kMaxDevices);
        return TGCN_ERR_HIP;
    }
    std::call_once(once[dev], [dev] {
        result[dev] = hipFuncSetAttribute(reinterpret_cast<const void *>(k_brute_part), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          160 << 10);
    });
    if (result[dev] != hipSuccess) {
        set_error("hipFuncSetAttribute(k_brute_part, MaxDynamicSharedMemorySize) on device %d: %s", dev, hipGetErrorString(result[dev]));
        return TGCN_ERR_HIP;
    }
    return TGCN_OK;
}

template <int DQ>
int launch_filter(const FilterArgs &a, hipStream_t s)
{
    const dim3 grid((a.B + kUsersPerWG - 1) / kUsersPerWG, a.S);
    if (a.d == 4 * DQ)
        hipLaunchKernelGGL((k_score_filter<DQ, true>), grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((k_score_filter<DQ, false>), grid, dim3(256), 0, s, a);
    return check_launch("k_score_filter");
}

}  // namespace
}  // namespace tgcn

using namespace tgcn;


extern "C" int64_t tgcn_score_topk_workspace_bytes(int32_t B, int32_t I, int32_t d, int32_t k)
{
    if (B <= 0 || I <= 0 || d <= 0)
        return 0;
    return (int64_t)make_plan(B, I, d, k).total;
}

namespace {
// prefilter: candidates from the bf16 pass of tgcn_score_prefilter.hip, rescored in fp32 (same results; d <= 128, or d <= 1024 with d % 8 == 0;
// otherwise the fp32 filter runs).  item_pack: device pointer to the packed item operand (tgcn_item_pack_bf16), or NULL: packed by
// this call.
constexpr int kFuseSelectMaxUsers = 4096;     // calls up to this size select inside k_rescore (see its SELECT note)

int score_topk_impl(const float *U, const int64_t *user_ids, int32_t B, const float *It, int32_t I, int32_t d,
                    const int32_t *mask_rowptr, const int32_t *mask_items, int32_t k, int32_t round4, float *out_val,
                    int64_t *out_idx, void *workspace, int64_t workspace_bytes, tgcn_stream_t stream, bool prefilter,
                    const void *item_pack)
{
    TGCN_REQUIRE(B >= 0 && I >= 0, "negative size");
    TGCN_REQUIRE(d > 0 && d <= 4096, "d out of range");
    TGCN_REQUIRE(k >= 1 && k <= 64, "k must be in [1, 64]");
    TGCN_REQUIRE(I >= k, "k exceeds the number of items");
    if (B == 0)
        return TGCN_OK;
    TGCN_REQUIRE(U && It && out_val && out_idx, "NULL pointer");
    TGCN_REQUIRE(!mask_rowptr || mask_items, "mask_rowptr without mask_items");
    TGCN_REQUIRE(B <= 65535 * kUsersPerWG, "B too large for one launch");
    TGCN_REQUIRE(((size_t)item_pack & 15) == 0, "item_pack must be 16-byte aligned");
    Plan p = make_plan(B, I, d, k);
    TGCN_REQUIRE(workspace && workspace_bytes >= (int64_t)p.total, "workspace too small (tgcn_score_topk_workspace_bytes)");
    TGCN_REQUIRE(((size_t)workspace & 255) == 0, "workspace must be 256-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    char *ws = static_cast<char *>(workspace);
    int rc;

    if (p.small) {  // tiny catalogues: the three-kernel dense path on a workspace-resident [B, I] matrix
        float *S = reinterpret_cast<float *>(ws + p.off_sample);
        if ((rc = launch_score_dense(U, user_ids, B, It, I, d, 1, S, I, s)) != TGCN_OK)
            return rc;
        if (mask_rowptr && (rc = launch_mask(S, I, B, I, mask_rowptr, mask_items, 1, s)) != TGCN_OK)
            return rc;
        return launch_topk(S, I, B, I, k, round4, out_val, out_idx, s);
    }

    // prefilter mode: the item-side factor of the bound, unless the caller holds it
    prefilter = prefilter && prefilter_supports(d);
    float *ubound = prefilter ? reinterpret_cast<float *>(ws + p.off_ubound) : nullptr;
    const void *ipack = item_pack;
    if (prefilter && !ipack) {
        if ((rc = launch_item_pack(It, I, d, ws + p.off_ipack, s)) != TGCN_OK)
            return rc;
        ipack = ws + p.off_ipack;
    }
    // 1. tau from a strided item sample
    if (!prefilter)
        set_sample(p, I, d, k, false);       // (make_plan holds the prefiltered entry point's; the regions fit both)
    float *Ss = reinterpret_cast<float *>(ws + p.off_sample);
    float *tauv = reinterpret_cast<float *>(ws + p.off_tauv);
    int64_t *taui = reinterpret_cast<int64_t *>(ws + p.off_taui);
    float *tau1 = reinterpret_cast<float *>(ws + p.off_tau);
    int *flagged = reinterpret_cast<int *>(ws + p.off_flags);
    int *done = reinterpret_cast<int *>(ws + p.off_done);
    int *totals = prefilter ? reinterpret_cast<int *>(ws + p.off_totals) : nullptr;
    // (the sample is only ranked -- tau is a bar, never a result: up to d = 128 it comes from the bf16 pipe in both entry points)
    if ((rc = p.top && prefilter ? launch_sample_pack_top(U, user_ids, B, ipack, p.m, d, p.stride, mask_rowptr, mask_items,
                                                          reinterpret_cast<unsigned *>(ws + p.off_bits), Ss, p.m_ld, s)     // (rows from the pack)
              : p.top     ? launch_sample_top(U, user_ids, B, It, p.m, d, p.stride, mask_rowptr, mask_items,
                                              reinterpret_cast<unsigned *>(ws + p.off_bits), Ss, p.m_ld, s)     // (train items masked inside)
              : d <= 128  ? launch_sample_bf16(U, user_ids, B, It, p.m, d, p.stride, Ss, p.m_ld, s)
              : prefilter ? launch_sample_wide(U, user_ids, B, ipack, I, p.m, d, p.stride, Ss, p.m_ld, s)      // (from the pack)
                          : launch_score_dense(U, user_ids, B, It, p.m, d, p.stride, Ss, p.m_ld, s)) != TGCN_OK)
        return rc;
    const float *tau_ptr;
    int tau_stride;
    const int *mask_rowptr_tau = mask_rowptr;      // (the top form's values are masked already)
    if (p.m_rank <= 64 * kWave) {   // one wave per user: mask + rank selection in one launch
        const dim3 grid((B + 3) / 4);
        const int vpl = p.m_rank <= 2 * kWave ? 2 : p.m_rank <= 8 * kWave ? 8 : p.m_rank <= 16 * kWave ? 16 : p.m_rank <= 32 * kWave ? 32 : 64;
        const size_t lds = (size_t)4 * kWave * vpl * sizeof(float);
        if (p.top)
            mask_rowptr_tau = nullptr;
        switch (vpl) {
            case 2: hipLaunchKernelGGL((k_tau<2>), grid, dim3(256), lds, s, Ss, p.m_ld, B, p.m_rank, mask_rowptr_tau, mask_items, tau1, flagged, done, U, user_ids, d, ubound, totals, p.stride, p.rank); break;      // (the in-kernel sample's 2 x blocks values)
            case 8: hipLaunchKernelGGL((k_tau<8>), grid, dim3(256), lds, s, Ss, p.m_ld, B, p.m_rank, mask_rowptr_tau, mask_items, tau1, flagged, done, U, user_ids, d, ubound, totals, p.stride, p.rank); break;
            case 16: hipLaunchKernelGGL((k_tau<16>), grid, dim3(256), lds, s, Ss, p.m_ld, B, p.m_rank, mask_rowptr_tau, mask_items, tau1, flagged, done, U, user_ids, d, ubound, totals, p.stride, p.rank); break;
            case 32: hipLaunchKernelGGL((k_tau<32>), grid, dim3(256), lds, s, Ss, p.m_ld, B, p.m_rank, mask_rowptr_tau, mask_items, tau1, flagged, done, U, user_ids, d, ubound, totals, p.stride, p.rank); break;
            default: hipLaunchKernelGGL((k_tau<64>), grid, dim3(256), lds, s, Ss, p.m_ld, B, p.m_rank, mask_rowptr_tau, mask_items, tau1, flagged, done, U, user_ids, d, ubound, totals, p.stride, p.rank); break;
        }
        if ((rc = check_launch("k_tau")) != TGCN_OK)
            return rc;
        tau_ptr = tau1, tau_stride = 1;
    } else {                   // very large catalogues (> 131 072 items): mask + workgroup-per-row top-k on the sample
        // the fallback's flag count and arrival counters start from zero (the workspace arrives uninitialised)
        if (hipMemsetAsync(flagged, 0, (size_t)(ws + p.off_totals - reinterpret_cast<char *>(flagged)) + (size_t)B * sizeof(int), s) != hipSuccess)
            return check_launch("hipMemsetAsync(flagged, done)");
        if (mask_rowptr && (rc = launch_mask(Ss, p.m_ld, B, p.m, mask_rowptr, mask_items, p.stride, s)) != TGCN_OK)
            return rc;
        if ((rc = launch_topk(Ss, p.m_ld, B, p.m, p.rank, 0, tauv, taui, s)) != TGCN_OK)
            return rc;
        tau_ptr = tauv + (p.rank - 1), tau_stride = p.rank;
    }
    // 2. filtered GEMM over all items
    FilterArgs fa;
    fa.U = U, fa.user_ids = user_ids, fa.It = It, fa.tau = tau_ptr, fa.tau_stride = tau_stride;
    fa.logs = reinterpret_cast<float2 *>(ws + p.off_logs);
    fa.counts = reinterpret_cast<int *>(ws + p.off_counts);
    fa.B = B, fa.I = I, fa.d = d, fa.S = p.S, fa.items_per_split = p.items_per_split, fa.cap2 = p.cap2;
    bool selected = false;
    if (prefilter) {   // pass bits from the bf16 GEMM, then candidates -> fp32 chains -> flat (score, item) lists
        unsigned *mask = reinterpret_cast<unsigned *>(ws + p.off_mask);
        if (tau_stride != 1 && (rc = launch_user_bound(U, user_ids, B, d, ubound, s)) != TGCN_OK)
            return rc;       // (the k_tau launch wrote the users' factors itself)
        // its own item splits: pass bits are indexed by unit, not by split, so the grid need not be k_select's 32 segments.
        // Measured (rocprofv3): d = 128, 60 k items, 2048 users 68.6 -> 62.2 us with 1024-item splits.  d = 64: calls with
        // plenty of workgroups take splits of 256-item multiples (one aligned 16-byte store of mask words per lane and stage:
        // 16 384 users 208 -> 181 us); a 2048-user call keeps its 32 splits (28 would cost it 35.5 -> 39 us).
        if (d > 128) {
            // wide rows: the filter logs its candidates (k_select's segment layout: the plan's splits), k_refine keeps those that
            // can still reach the top k, the fp32 chains run on what is left
            int *surv = reinterpret_cast<int *>(ws + p.off_surv), *surv_n = reinterpret_cast<int *>(ws + p.off_surv_n);
            // at most 16 splits (make_plan): 4 lane-private log segments per split and user, 64 segments = one per lane of k_refine
            if ((rc = launch_prefilter_wide(U, user_ids, B, ipack, I, d, tau_ptr, tau_stride, ubound, fa.logs, fa.counts, p.S_w, p.ips_w,
                                            p.cap2_w, s)) != TGCN_OK)
                return rc;
            RefineArgs ra{fa.logs, fa.counts, mask_rowptr, mask_items, ubound, static_cast<const unsigned char *>(ipack), pack_row_bytes(d),
                          surv, surv_n, kRefineCap, B, 2 * p.S_w, p.cap2_w, k};
            hipLaunchKernelGGL(k_refine, dim3((B + 3) / 4), dim3(256), 0, s, ra);
            if ((rc = check_launch("k_refine")) != TGCN_OK)
                return rc;
            // (the flat lists of the kept pairs go where the filter's logs were: k_refine is done with them)
            if ((rc = launch_rescore_list(U, user_ids, B, It, d, tau_ptr, tau_stride, surv, surv_n, kRefineCap, fa.logs, totals,
                                          p.n_seg * p.cap2, s)) != TGCN_OK)
                return rc;
        } else {
        // as few item splits as fill the chip ONCE (a 512-thread workgroup per CU: 256 workgroups).  A workgroup's prologue -- its
        // 256 users' rows by two dependent round trips, their bf16 image through LDS into fragments -- hides under nothing, and
        // with 28 splits a 16 384-user launch paid it in seven generations of workgroups: with the MFMAs alone left in the loop
        // (round 3's ablation build pre_onlymfma) the launch still took 109 us for 53 us of matrix work.  The one-store-per-stage form
        // needs splits of whole stages: for large calls, and wherever a split is long enough that the rounding costs nothing.
        const int tiles = (B + 255) / 256;
        const int s_target = max(1, min(32, 256 / tiles));
        const bool wide = B > 4096 || I / s_target >= 16 * 256;
        int ips_pre = p.items_per_split;
        // splits of at least 16 double-length stages over a pack the caches hold: the ring of two (half the stage boundaries per item;
        // one stage of lead covers a round trip to the L2 / Infinity Cache, not to HBM -- A/B on one box, four streams: 16 384 users x
        // 60 k x 128 537 -> 512 us, x 50 k x 64 311 -> 301; config 4's 288 MB pack 845 -> 892, so it keeps the ring of three)
        const int st_long = prefilter_stage_items(d, true);
        const bool long_stages = wide && (I + s_target - 1) / s_target >= 16 * st_long && item_pack_bytes(I, d) <= kLongStagePackBytes;
        if (long_stages) {
            ips_pre = (((I + s_target - 1) / s_target + st_long - 1) / st_long) * st_long;
        } else if (wide) {
            ips_pre = (((I + s_target - 1) / s_target + 255) / 256) * 256;
        } else if (d > 64 && I <= (1 << 18)) {
            ips_pre = 1024;      // (768 / 1280 / 2048 with the packed operand: 54 / 53 / 56 us against 46)
        }
        // large catalogues (one-store-per-stage form): the filter also keeps one bit per stage, and the chains' kernel reads the flagged
        // stages' words instead of scanning the user's whole row (2 M items: 250 KB per user for ~200 set bits)
        PassSummary summ;
        const int n_splits_pre = (I + ips_pre - 1) / ips_pre;
        if (wide && p.summ_words > 0) {
            summ.stage_items = prefilter_stage_items(d, long_stages);
            summ.items_per_split = ips_pre, summ.n_splits = n_splits_pre;
            summ.sw = (ips_pre / summ.stage_items + 31) / 32;
            if (summ.n_splits * summ.sw <= p.summ_words)        // (always: make_plan's bound is stages / 32 + one word per split)
                summ.words = reinterpret_cast<unsigned *>(ws + p.off_summ);
        }
        if ((rc = launch_prefilter(U, user_ids, B, ipack, I, d, tau_ptr, tau_stride, ubound, mask, p.Wh, n_splits_pre, ips_pre, wide, long_stages,
                                   summ, s)) != TGCN_OK)
            return rc;
        if (B <= kFuseSelectMaxUsers) {
            // small calls: the fp32 chains and the exact selection in ONE launch, the kept pairs never leave the wave's LDS (round 4)
            rc = launch_rescore_select(U, user_ids, B, It, d, tau_ptr, tau_stride, mask, p.Wh, (I + kStage - 1) / kStage, summ, totals,
                                       mask_rowptr, mask_items, k, round4, out_val, out_idx, flagged, s);
            selected = true;
        } else {
            // large calls: k_select_flat as its own launch runs 32 waves per CU where the fused kernel's LDS allows 8
            rc = launch_rescore(U, user_ids, B, It, d, tau_ptr, tau_stride, mask, p.Wh, (I + kStage - 1) / kStage, summ, fa.logs, totals,
                                p.S * 2 * p.cap2, s);
        }
        }
    } else if (d <= 128) {
        const dim3 grid((B + kUsersPerWG - 1) / kUsersPerWG, p.S);
        if (d == 64 && B <= 4096)
            hipLaunchKernelGGL((k_score_filter16_small<true>), grid, dim3(256), 0, s, fa);
        else if (d == 64)
            hipLaunchKernelGGL((k_score_filter16<true>), grid, dim3(256), 0, s, fa);
        else if (d < 64 && B <= 4096)
            hipLaunchKernelGGL((k_score_filter16_small<false>), grid, dim3(256), 0, s, fa);
        else if (d < 64)
            hipLaunchKernelGGL((k_score_filter16<false>), grid, dim3(256), 0, s, fa);
        else if (d == 128)
            hipLaunchKernelGGL((k_score_filter32<true>), grid, dim3(256), 0, s, fa);
        else
            hipLaunchKernelGGL((k_score_filter32<false>), grid, dim3(256), 0, s, fa);
        rc = check_launch("k_score_filter16/32");
    } else {
        rc = d <= 256 ? launch_filter<64>(fa, s)
            : launch_score_dense_filter(U, user_ids, B, It, I, d, tau_ptr, tau_stride, fa.logs, fa.counts, p.S, p.items_per_split, p.cap2, s);
    }
    if (rc != TGCN_OK)
        return rc;

    // 3. exact selection from the logs; 4. exact rescoring of flagged users
    SelectArgs sa{fa.logs, fa.counts, mask_rowptr, mask_items, out_val, out_idx, flagged, B, p.S, p.cap2, k, round4, totals,
                  (prefilter && d > 128 ? p.n_seg : 2 * p.S) * p.cap2};
    if (selected)
        ;      // (narrow prefiltered rows: k_rescore selected already)
    else if (sa.totals)
        hipLaunchKernelGGL(k_select_flat, dim3((B + 3) / 4), dim3(256), 0, s, sa);
    else
        hipLaunchKernelGGL(k_select, dim3((B + 3) / 4), dim3(256), 0, s, sa);
    if ((rc = check_launch("k_select")) != TGCN_OK)
        return rc;
    BruteArgs ba{U, user_ids, It, mask_rowptr, mask_items, flagged, reinterpret_cast<float2 *>(ws + p.off_parts), done, p.flag_cap,
                 out_val, out_idx, B, I, d, k, round4};
    size_t brute_lds = ((size_t)((d + 63) & ~63) + 2 * kBruteWaves * kWave) * sizeof(float);
    if ((d & 3) == 0)
        brute_lds += (size_t)kBruteWaves * kWave * (min(d, kBruteTileMaxD) + 1) * sizeof(float);
    if ((rc = brute_lds_opt_in()) != TGCN_OK)
        return rc;
    hipLaunchKernelGGL(k_brute_part, dim3(kBruteSplits, 8), dim3(kBruteWaves * 64), brute_lds, s, ba);
    return check_launch("k_brute_part");
}
}  // namespace

extern "C" int tgcn_score_topk_f32(const float *U, const int64_t *user_ids, int32_t B, const float *It, int32_t I,
                                   int32_t d, const int32_t *mask_rowptr, const int32_t *mask_items, int32_t k,
                                   int32_t round4, float *out_val, int64_t *out_idx, void *workspace,
                                   int64_t workspace_bytes, tgcn_stream_t stream)
{
    return score_topk_impl(U, user_ids, B, It, I, d, mask_rowptr, mask_items, k, round4, out_val, out_idx, workspace,
                           workspace_bytes, stream, false, nullptr);
}

extern "C" int tgcn_score_topk_prefilter_f32(const float *U, const int64_t *user_ids, int32_t B, const float *It, int32_t I,
                                             int32_t d, const int32_t *mask_rowptr, const int32_t *mask_items, int32_t k,
                                             int32_t round4, const void *item_pack, float *out_val, int64_t *out_idx,
                                             void *workspace, int64_t workspace_bytes, tgcn_stream_t stream)
{
    return score_topk_impl(U, user_ids, B, It, I, d, mask_rowptr, mask_items, k, round4, out_val, out_idx, workspace,
                           workspace_bytes, stream, true, item_pack);
}

extern "C" int tgcn_score_topk_fallback_count(const void *workspace, int32_t B, int32_t I, int32_t d, int32_t k, int32_t *out_host,
                                              tgcn_stream_t stream)
{
    TGCN_REQUIRE(workspace && out_host, "NULL pointer");
    TGCN_REQUIRE(B > 0 && I > 0 && d > 0, "empty call");
    Plan p = make_plan(B, I, d, k);
    *out_host = 0;
    if (p.small)
        return TGCN_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemcpyAsync(out_host, static_cast<const char *>(workspace) + p.off_flags, sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess)
        return check_launch("tgcn_score_topk_fallback_count");
    return TGCN_OK;
}

extern "C" int tgcn_score_topk_stats(void *workspace, int32_t B, int32_t I, int32_t d, int32_t k, int32_t prefilter, int64_t *out_host,
                                     tgcn_stream_t stream)
{
    TGCN_REQUIRE(workspace && out_host, "NULL pointer");
    TGCN_REQUIRE(B > 0 && I > 0 && d > 0, "empty call");
    Plan p = make_plan(B, I, d, k);
    out_host[0] = out_host[1] = out_host[2] = out_host[3] = 0;
    if (p.small)
        return TGCN_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    char *ws = static_cast<char *>(workspace);
    const bool pre = prefilter && prefilter_supports(d);
    const bool wide = pre && d > 128;
    StatsArgs a{};
    a.flagged = reinterpret_cast<const int *>(ws + p.off_flags);
    a.totals = pre ? reinterpret_cast<const int *>(ws + p.off_totals) : nullptr;
    a.mask = pre && !wide ? reinterpret_cast<const unsigned *>(ws + p.off_mask) : nullptr;
    a.Wh = p.Wh, a.n_units = (I + kStage - 1) / kStage;
    a.surv_n = wide ? reinterpret_cast<const int *>(ws + p.off_surv_n) : nullptr;
    // (the narrow bf16 path rewrites the log area with k_rescore's flat lists: its counts are not logs)
    a.counts = (!pre || wide) ? reinterpret_cast<const int *>(ws + p.off_counts) : nullptr;
    a.n_seg = wide ? 4 * p.S_w : 2 * p.S, a.cap2 = wide ? p.cap2_w : p.cap2, a.B = B;
    a.out = reinterpret_cast<unsigned long long *>(ws + p.off_stats);
    if (hipMemsetAsync(a.out, 0, 32, s) != hipSuccess)
        return check_launch("hipMemsetAsync(stats)");
    hipLaunchKernelGGL(k_stats, dim3((B + 255) / 256), dim3(256), 0, s, a);
    int rc = check_launch("k_stats");
    if (rc != TGCN_OK)
        return rc;
    if (hipMemcpyAsync(out_host, a.out, 32, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        return check_launch("tgcn_score_topk_stats");
    return TGCN_OK;
}

extern "C" int64_t tgcn_item_pack_bytes(int32_t I, int32_t d)
{
    return (I < 0 || d <= 0) ? -1 : (int64_t)item_pack_bytes(I, d);
}

extern "C" int tgcn_item_pack_bf16(const float *It, int32_t I, int32_t d, void *out, tgcn_stream_t stream)
{
    TGCN_REQUIRE(I >= 0, "negative size");
    TGCN_REQUIRE(d > 0 && d <= 4096, "d out of range");
    TGCN_REQUIRE(prefilter_supports(d), "no bf16 candidate pass for this width (tgcn_item_pack_bytes() == 0)");
    if (I == 0)
        return TGCN_OK;
    TGCN_REQUIRE(It && out, "NULL pointer");
    TGCN_REQUIRE(((size_t)out & 15) == 0, "out must be 16-byte aligned");
    return launch_item_pack(It, I, d, out, static_cast<hipStream_t>(stream));
}

extern "C" int tgcn_item_norms_f32(const float *It, int32_t I, int32_t d, float *out, tgcn_stream_t stream)
{
    TGCN_REQUIRE(I >= 0, "negative size");
    TGCN_REQUIRE(d > 0 && d <= 4096, "d out of range");
    if (I == 0)
        return TGCN_OK;
    TGCN_REQUIRE(It && out, "NULL pointer");
    return launch_item_norms(It, I, d, out, static_cast<hipStream_t>(stream));
}
