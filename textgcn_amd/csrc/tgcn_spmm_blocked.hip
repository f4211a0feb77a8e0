// K1 + K3, cache-blocked: the same product as tgcn_spmm_csr_f32 with the row loads served from L2.
//
// Why.  At d = 64 every stored entry gathers a 256-byte row of X.  On BASELINE config 2 the table is 38 MB:
// it lives in the 256 MB Infinity Cache, not in an XCD's 4 MB L2, so the plain kernel moves ~1.7 GB of L2-miss
// traffic per layer for 0.16 GB of compulsory bytes (profiles/r01a_pmc.json: L2 hit rate 37 %), and runs at the
// Infinity-Cache gather rate.  A 4 MB table gathers 2x faster on the same kernel (profiles/r01_spmm_variants.md).
//
// How.  The columns a row block reads (item rows for users, user rows for items) are cut into blocks of
// `block_width` rows of X (~3 MB).  All waves walk the column blocks in the same order, so at any moment the
// whole chip gathers from one ~3 MB slice that every XCD's L2 can hold.  A row's accumulator stays in
// REGISTERS across the blocks: each 16-lane group (d = 64) owns NSET rows for the whole launch, so nothing is
// spilled to or re-read from memory between blocks and the fp32 fmaf chain of a row is exactly the sequential
// one (ascending columns) -- results are bit-identical to k_spmm_wave / the reference CPU kernel.
// No inter-workgroup synchronisation: workgroups start together and do statistically equal work per block;
// drifting apart costs hit rate, never correctness.
//
// STATUS (round 1): correct and bit-identical, but NOT faster on MI355X -- config 2 runs 127 us (user rows, L2 hit
// rate 70 %) + 297 us (item rows, Zipf lengths keep the groups out of step: hit rate 28 %) + 58 us (long rows) per
// layer against 263 us for the plain kernel (profiles/r01_blocked_experiment.md).  Kept opt-in.
//
// Rows longer than the split threshold are not touched here (their block segments are empty in the plan); the
// chunk waves + k_spmm_long_reduce of tgcn_spmm.hip handle them as before.
#include <climits>

#include "tgcn_internal.h"

namespace tgcn {
namespace {

struct BlockedArgs {
    const int *__restrict__ blkptr;  // [(n_blocks+1), ld]
    const int *__restrict__ rowptr;  // of the whole local CSR (long-row test)
    const int *__restrict__ colidx;
    const float *__restrict__ vals;
    const float *__restrict__ X;
    float *__restrict__ Y;
    const float *acc_in;
    float *acc_out;
    float acc_div;
    int n_blocks, ld, row_begin, n_rows, threshold;
};

__device__ __forceinline__ void fma4(float4 &acc, float v, const float4 &x)
{
    acc.x = fmaf(v, x.x, acc.x);
    acc.y = fmaf(v, x.y, acc.y);
    acc.z = fmaf(v, x.z, acc.z);
    acc.w = fmaf(v, x.w, acc.w);
}

// A group of G lanes owns NSET rows.  Inside one column block the NSET segments advance TOGETHER: every round
// issues up to UNROLL row loads for each of them before any is consumed, so a group keeps NSET*UNROLL gathers
// in flight instead of walking its segments one after the other (which left the kernel latency-bound).
template <int G, int NSET, int UNROLL>
__global__ __launch_bounds__(256) void k_spmm_blocked(const BlockedArgs a)
{
    constexpr int R = kWave / G;
    constexpr int D = 4 * G;
    const int lane = lane_id();
    const int gl = lane & (G - 1);
    const int group = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R + lane / G;
    const int n_groups = gridDim.x * 4 * R;
    const float *__restrict__ Xl = a.X + gl * 4;

    float4 acc[NSET];
    int pos[NSET], pe[NSET];  // next entry / end of the current block's segment, per owned row
#pragma unroll
    for (int s = 0; s < NSET; ++s) {
        acc[s] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int lr = group + s * n_groups;
        pos[s] = pe[s] = 0;
        if (lr < a.n_rows) {
            pos[s] = a.blkptr[lr];
            pe[s] = a.blkptr[a.ld + lr];
        }
    }
    for (int cb = 0; cb < a.n_blocks; ++cb) {
        // bounds of the next block: loaded now, needed only after this block is done
        int ne[NSET];
#pragma unroll
        for (int s = 0; s < NSET; ++s) {
            const int lr = group + s * n_groups;
            ne[s] = 0;
            if (cb + 1 < a.n_blocks && lr < a.n_rows)
                ne[s] = a.blkptr[(size_t)(cb + 2) * a.ld + lr];
        }
        // (col, val) windows: lane gl of the group holds entry pos[s] + gl of segment s
        int c[NSET], wbase[NSET];
        float v[NSET];
#pragma unroll
        for (int s = 0; s < NSET; ++s) {
            wbase[s] = pos[s];
            c[s] = 0, v[s] = 0.0f;
            if (pos[s] + gl < pe[s]) {
                c[s] = a.colidx[pos[s] + gl];
                v[s] = a.vals[pos[s] + gl];
            }
        }
        bool any = false;
#pragma unroll
        for (int s = 0; s < NSET; ++s)
            any |= pos[s] < pe[s];
        while (any) {
            float4 x[NSET][UNROLL];
            float vv[NSET][UNROLL];
#pragma unroll
            for (int s = 0; s < NSET; ++s) {
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    const int e = min(pos[s] + u, pe[s] - 1);       // clamped: a finished segment re-reads a valid row
                    const int j = max(e - wbase[s], 0);
                    const int cj = __shfl(c[s], j, G);
                    vv[s][u] = __shfl(v[s], j, G);
                    x[s][u] = *reinterpret_cast<const float4 *>(Xl + (size_t)cj * D);
                }
            }
            any = false;
#pragma unroll
            for (int s = 0; s < NSET; ++s) {
#pragma unroll
                for (int u = 0; u < UNROLL; ++u)
                    if (pos[s] + u < pe[s])
                        fma4(acc[s], vv[s][u], x[s][u]);
                pos[s] = min(pos[s] + UNROLL, pe[s]);
                if (pos[s] < pe[s] && pos[s] + UNROLL > wbase[s] + G) {   // window exhausted: slide it
                    wbase[s] = pos[s];
                    c[s] = 0, v[s] = 0.0f;
                    if (pos[s] + gl < pe[s]) {
                        c[s] = a.colidx[pos[s] + gl];
                        v[s] = a.vals[pos[s] + gl];
                    }
                }
                any |= pos[s] < pe[s];
            }
        }
#pragma unroll
        for (int s = 0; s < NSET; ++s)
            pos[s] = pe[s], pe[s] = ne[s];
    }
#pragma unroll
    for (int s = 0; s < NSET; ++s) {
        const int lr = group + s * n_groups;
        if (lr >= a.n_rows)
            continue;
        const int row = a.row_begin + lr;
        if (a.rowptr[row + 1] - a.rowptr[row] > a.threshold)
            continue;  // long row: chunk waves + k_spmm_long_reduce
        const size_t off = (size_t)row * D + gl * 4;
        const float4 y = acc[s];
        if (a.Y)
            *reinterpret_cast<float4 *>(a.Y + off) = y;
        if (a.acc_out) {
            float4 t = *reinterpret_cast<const float4 *>(a.acc_in + off);
            t.x += y.x, t.y += y.y, t.z += y.z, t.w += y.w;
            if (a.acc_div != 1.0f)
                t.x /= a.acc_div, t.y /= a.acc_div, t.z /= a.acc_div, t.w /= a.acc_div;
            *reinterpret_cast<float4 *>(a.acc_out + off) = t;
        }
    }
}

template <int G, int NSET>
void launch_nset(const BlockedArgs &a, int grid, hipStream_t s)
{
    hipLaunchKernelGGL((k_spmm_blocked<G, NSET, (NSET >= 4 ? 2 : 4)>), dim3(grid), dim3(256), 0, s, a);
}

template <int G>
int launch_blocked(const BlockedArgs &a, hipStream_t s)
{
    constexpr int R = kWave / G;
    constexpr int kMaxGrid = 2048;  // 8 workgroups of 256 threads per CU: everything resident, blocks walked in step
    const int groups_per_wg = 4 * R;
    int nset = (a.n_rows + kMaxGrid * groups_per_wg - 1) / (kMaxGrid * groups_per_wg);
    nset = nset <= 1 ? 1 : nset <= 2 ? 2 : 4;
    const int grid = (a.n_rows + nset * groups_per_wg - 1) / (nset * groups_per_wg);
    if (nset == 1)
        launch_nset<G, 1>(a, grid, s);
    else if (nset == 2)
        launch_nset<G, 2>(a, grid, s);
    else
        launch_nset<G, 4>(a, grid, s);
    return check_launch("k_spmm_blocked");
}

}  // namespace
}  // namespace tgcn

using namespace tgcn;

extern "C" int tgcn_spmm_blocked_f32(const tgcn_block_plan_t *plans, int32_t n_plans, const int32_t *rowptr,
                                     const int32_t *colidx, const float *vals, int64_t n_rows, const float *X,
                                     int64_t n_src_rows, int32_t d, float *Y, const float *acc_in, float *acc_out,
                                     float acc_div, const tgcn_split_plan_t *split, tgcn_stream_t stream)
{
    TGCN_REQUIRE(plans && n_plans > 0, "no block plans");
    TGCN_REQUIRE(d == 64 || d == 128 || d == 256, "blocked SpMM supports d in {64, 128, 256}");
    TGCN_REQUIRE(n_rows > 0 && n_rows < INT_MAX - 256, "n_rows out of range");
    TGCN_REQUIRE(n_src_rows > 0 && n_src_rows < INT_MAX, "n_src_rows out of range");
    TGCN_REQUIRE(rowptr && colidx && vals && X, "NULL pointer");
    TGCN_REQUIRE(Y || acc_out, "both Y and acc_out are NULL: nothing to compute");
    TGCN_REQUIRE(!acc_out || acc_in, "acc_out given without acc_in");
    TGCN_REQUIRE(acc_div != 0.0f, "acc_div must be non-zero");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int threshold = (split && split->n_chunks > 0) ? split->threshold : INT_MAX;
    int64_t covered = 0;
    for (int p = 0; p < n_plans; ++p) {
        const tgcn_block_plan_t &pl = plans[p];
        TGCN_REQUIRE(pl.blkptr && pl.n_blocks >= 1 && pl.n_rows >= 0 && pl.ld >= pl.n_rows, "malformed block plan");
        TGCN_REQUIRE(pl.row_begin >= 0 && (int64_t)pl.row_begin + pl.n_rows <= n_rows, "block plan rows out of range");
        covered += pl.n_rows;
    }
    TGCN_REQUIRE(covered == n_rows, "block plans must cover every row exactly once");
    for (int p = 0; p < n_plans; ++p) {
        const tgcn_block_plan_t &pl = plans[p];
        if (pl.n_rows == 0)
            continue;
        BlockedArgs a{pl.blkptr, rowptr, colidx, vals, X, Y, acc_in, acc_out, acc_div, pl.n_blocks, pl.ld, pl.row_begin, pl.n_rows,
                      threshold};
        const int rc = d == 64 ? launch_blocked<16>(a, s) : d == 128 ? launch_blocked<32>(a, s) : launch_blocked<64>(a, s);
        if (rc != TGCN_OK)
            return rc;
    }
    if (threshold == INT_MAX)
        return TGCN_OK;
    // long rows: chunk waves only (n_rows = 0 row waves) + ordered reduce, exactly as the unblocked kernel does
    return launch_long_rows(rowptr, colidx, vals, (int)n_rows, X, d, Y, acc_in, acc_out, acc_div, split, s);
}
