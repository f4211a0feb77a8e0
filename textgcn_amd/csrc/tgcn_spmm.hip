// K1 + K3: CSR SpMM  Y = A.X  (fp32) with the LightGCN layer combination fused into the epilogue.
//
// Replaces torch.sparse.mm (TextGCN/base_model.py:148) and the running sum of
// torch.mean(torch.stack(...)) (base_model.py:157).  gfx950 only.
//
// Numerics contract: every output element is the fp32 fmaf chain over the row's stored entries in
// ascending column order, starting from +0 -- the order torch's CPU COO kernel uses --
// so the result is bit-identical to the reference's CPU forward.  Rows longer than the split plan's
// threshold are the one exception: their chunks are chained independently and the chunk sums are
// added in chunk order by a second launch (deterministic; differs from the reference in rounding only).
// The segmented form (tgcn_spmm_segmented_f32) has the same contract for the rows it cuts at column-block
// boundaries; its direct rows are exact chains.
//
// Roofline: HBM-bound gather.  Algorithmic bytes per layer:
//   nnz*(4+4) + (rows+1)*4 + n_src*d*4 (X read once) + rows*d*4 (Y) [+ acc read/write]
//
// Kernels
//   k_spmm_wave<VEC,UNROLL>   one wave64 per row; lane owns VEC consecutive columns (d = 64*VEC).
//                             The row's (col, val) pairs are loaded 64 at a time, coalesced, and
//                             broadcast with v_readlane into SGPRs: per entry one scalar address
//                             computation, one 256*VEC-byte coalesced row load, VEC v_fma.
//   k_spmm_generic            any d: wave per row, 64-column slabs.
//   k_spmm_long_reduce<VEC>   adds the chunk sums of split rows and applies the epilogue.
//   k_spmm_seg<VEC,UNROLL>    XCD-affine column blocking: tile waves walk per-class entry streams (the workgroups of
//                             one XCD only gather from 1/8 of the table, which then lives in that XCD's L2) and
//                             write piece sums; the other waves of the launch own one direct row each.
//   k_spmm_seg_reduce<VEC>    adds a row's piece sums in column order and applies the epilogue.
#include <climits>

#include "tgcn_internal.h"

namespace tgcn {
namespace {

struct SpmmArgs {
    const int *__restrict__ rowptr;
    const int *__restrict__ colidx;
    const float *__restrict__ vals;
    const float *__restrict__ X;
    float *__restrict__ Y;
    const float *acc_in;  // may alias acc_out
    float *acc_out;
    float acc_div;
    int n_rows;
    int d;
    int row_waves;  // waves [0, n_chunks) own chunks of long rows -- dispatched FIRST, they are the longest work items
                    // (up to `threshold` entries each) -- and waves [n_chunks, n_chunks + row_waves) own rows
    const int *__restrict__ row_order;  // optional: wave w (after the chunk waves) owns row row_order[w]
    int threshold;  // rows with more entries than this are left to the chunk waves
    int n_chunks;
    const int *__restrict__ chunk_beg;
    const int *__restrict__ chunk_end;
    float *__restrict__ ws;
};

// Streams touched once per launch -- the CSR entries, acc_in, Y, acc_out -- carry the non-temporal hint: what the L2s and the
// Infinity Cache should keep is the GATHERED table (config 4: the Zipf-popular item rows), not 7 GB of bytes nobody reads twice.
// A/B on one box (profiles/r04_experiments.md section 6): config 4 layer 7.49-7.56 -> 7.32-7.41 ms, config 2 198 -> 194, config 3 232 -> 228 us.
using f32x2_t = __attribute__((ext_vector_type(2))) float;
using f32x4_t = __attribute__((ext_vector_type(4))) float;
template <typename T>
__device__ __forceinline__ T ld_stream(const T *p)
{
    return __builtin_nontemporal_load(p);
}
template <int VEC>
__device__ __forceinline__ void load_vec_stream(const float *p, float (&x)[VEC])
{
    if constexpr (VEC == 1) {
        x[0] = ld_stream(p);
    } else if constexpr (VEC == 2) {
        const f32x2_t t = ld_stream(reinterpret_cast<const f32x2_t *>(p));
        x[0] = t.x, x[1] = t.y;
    } else {
        const f32x4_t t = ld_stream(reinterpret_cast<const f32x4_t *>(p));
        x[0] = t.x, x[1] = t.y, x[2] = t.z, x[3] = t.w;
    }
}
template <int VEC>
__device__ __forceinline__ void store_vec_stream(float *p, const float (&x)[VEC])
{
    if constexpr (VEC == 1) {
        __builtin_nontemporal_store(x[0], p);
    } else if constexpr (VEC == 2) {
        const f32x2_t t = {x[0], x[1]};
        __builtin_nontemporal_store(t, reinterpret_cast<f32x2_t *>(p));
    } else {
        const f32x4_t t = {x[0], x[1], x[2], x[3]};
        __builtin_nontemporal_store(t, reinterpret_cast<f32x4_t *>(p));
    }
}

template <int VEC>
__device__ __forceinline__ void load_vec(const float *p, float (&x)[VEC])
{
    if constexpr (VEC == 1) {
        x[0] = *p;
    } else if constexpr (VEC == 2) {
        const float2 t = *reinterpret_cast<const float2 *>(p);
        x[0] = t.x, x[1] = t.y;
    } else {
        static_assert(VEC == 4, "VEC must be 1, 2 or 4");
        const float4 t = *reinterpret_cast<const float4 *>(p);
        x[0] = t.x, x[1] = t.y, x[2] = t.z, x[3] = t.w;
    }
}

template <int VEC>
__device__ __forceinline__ void store_vec(float *p, const float (&x)[VEC])
{
    if constexpr (VEC == 1) {
        *p = x[0];
    } else if constexpr (VEC == 2) {
        *reinterpret_cast<float2 *>(p) = make_float2(x[0], x[1]);
    } else {
        *reinterpret_cast<float4 *>(p) = make_float4(x[0], x[1], x[2], x[3]);
    }
}

// Y / acc epilogue for VEC consecutive columns starting at `off` (element offset of row r, column c)
template <int VEC>
__device__ __forceinline__ void epilogue(const SpmmArgs &a, size_t off, const float (&y)[VEC])
{
    if (a.Y)
        store_vec<VEC>(a.Y + off, y);
    if (a.acc_out) {
        float t[VEC];
        load_vec<VEC>(a.acc_in + off, t);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            t[k] = t[k] + y[k];
            if (a.acc_div != 1.0f)
                t[k] = t[k] / a.acc_div;  // IEEE division: the reference divides (torch.mean), base_model.py:157
        }
        store_vec<VEC>(a.acc_out + off, t);
    }
}

// The same epilogue with the row's acc_in already in registers: the load is issued at the START of the row's work (only
// this wave writes the row, so acc_in aliasing acc_out is harmless) and has long landed when the chain ends -- one
// dependent round trip less per row than loading it here.
template <int VEC>
__device__ __forceinline__ void preload_acc(const SpmmArgs &a, size_t off, float (&t)[VEC])
{
#pragma unroll
    for (int k = 0; k < VEC; ++k)
        t[k] = 0.0f;
    if (a.acc_out)
        load_vec_stream<VEC>(a.acc_in + off, t);
}

template <int VEC>
__device__ __forceinline__ void epilogue_pre(const SpmmArgs &a, size_t off, const float (&y)[VEC], float (&t)[VEC])
{
    if (a.Y)
        store_vec_stream<VEC>(a.Y + off, y);
    if (a.acc_out) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            t[k] = t[k] + y[k];
            if (a.acc_div != 1.0f)
                t[k] = t[k] / a.acc_div;  // IEEE division: the reference divides (torch.mean), base_model.py:157
        }
        store_vec_stream<VEC>(a.acc_out + off, t);
    }
}

// One wave walks entries [beg, end) (wave-uniform bounds) of one row; lane owns columns
// [lane*VEC, lane*VEC+VEC).  acc continues the fmaf chain.
template <int VEC, int UNROLL>
__device__ __forceinline__ void accumulate_wave(const SpmmArgs &a, int beg, int end, int lane, float (&acc)[VEC])
{
    const float *__restrict__ Xl = a.X + lane * VEC;
    const size_t d = (size_t)a.d;
    for (int base = beg; base < end; base += kWave) {
        const int n = min(kWave, end - base);  // uniform
        int c = 0;
        float v = 0.0f;
        if (lane < n) {
            c = ld_stream(a.colidx + base + lane);
            v = ld_stream(a.vals + base + lane);
        }
        int j = 0;
        for (; j + UNROLL <= n; j += UNROLL) {
            float x[UNROLL][VEC];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int cj = __builtin_amdgcn_readlane(c, j + u);
                load_vec<VEC>(Xl + (size_t)cj * d, x[u]);
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const float vj = readlane_f(v, j + u);
#pragma unroll
                for (int k = 0; k < VEC; ++k)
                    acc[k] = fmaf(vj, x[u][k], acc[k]);
            }
        }
        if (j < n) {  // tail: same shape, clamped index + uniform predicate, so the loads still overlap
            float x[UNROLL][VEC];
#pragma unroll
            for (int u = 0; u < UNROLL - 1; ++u) {
                const int cj = __builtin_amdgcn_readlane(c, min(j + u, n - 1));
                load_vec<VEC>(Xl + (size_t)cj * d, x[u]);
            }
#pragma unroll
            for (int u = 0; u < UNROLL - 1; ++u) {
                if (j + u < n) {
                    const float vj = readlane_f(v, j + u);
#pragma unroll
                    for (int k = 0; k < VEC; ++k)
                        acc[k] = fmaf(vj, x[u][k], acc[k]);
                }
            }
        }
    }
}

// chunk wave: partial chain of one chunk of a long row -> workspace
template <int VEC, int UNROLL>
__device__ __forceinline__ void chunk_wave(const SpmmArgs &a, int chunk, int lane)
{
    const int beg = a.chunk_beg[chunk];
    const int end = a.chunk_end[chunk];
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k)
        acc[k] = 0.0f;
    accumulate_wave<VEC, UNROLL>(a, beg, end, lane, acc);
    store_vec<VEC>(a.ws + (size_t)chunk * a.d + lane * VEC, acc);
}

template <int VEC, int UNROLL>
__global__ __launch_bounds__(256) void k_spmm_wave(const SpmmArgs a)
{
    const int lane = lane_id();
    const int wave = uniform(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (wave < a.n_chunks) {
        chunk_wave<VEC, UNROLL>(a, wave, lane);
        return;
    }
    if (wave - a.n_chunks >= a.row_waves)
        return;
    const int row = a.row_order ? a.row_order[wave - a.n_chunks] : wave - a.n_chunks;
    const size_t off = (size_t)row * a.d + lane * VEC;
    float t[VEC];
    preload_acc<VEC>(a, off, t);  // before the rowptr -> (col, val) -> gather chain
    const int beg = a.rowptr[row];
    const int end = a.rowptr[row + 1];
    if (end - beg > a.threshold)
        return;  // long row: chunk waves + k_spmm_long_reduce
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k)
        acc[k] = 0.0f;
    accumulate_wave<VEC, UNROLL>(a, beg, end, lane, acc);
    epilogue_pre<VEC>(a, off, acc, t);
}

// ---- narrow tables (d in {8, 16, 32}) ---------------------------------------------------------------------------------------
// The feature (column) partition of the multi-GPU path (dist.ColumnShardedPropagator) leaves a rank d / P columns of every
// table.  G = d / 4 lanes own one work item -- a row, or one chunk of a long row -- with a float4 per lane, 64 / G items per
// wave; the item's (col, val) pairs are loaded G at a time by its lanes and broadcast with width-G shuffles.  Every output
// element is still the sequential fmaf chain over the row's entries in column order: bit-identical to the wide kernels on
// the same columns.  Row gathers are 4d bytes (128 B at d = 32: one cache line; below that the fabric over-fetches).
template <int G, int UNROLL>
__global__ __launch_bounds__(256) void k_spmm_narrow(const SpmmArgs a)
{
    static_assert(UNROLL <= G, "a group holds G (col, val) pairs at a time");
    constexpr int D = 4 * G;
    const int lane = lane_id();
    const int gl = lane & (G - 1);
    const long long item = ((long long)blockIdx.x * 256 + threadIdx.x) / G;
    const long long n_items = (long long)a.n_chunks + a.n_rows;
    const bool valid = item < n_items;
    int beg = 0, end = 0, row = -1;
    if (valid) {
        if (item < a.n_chunks) {
            beg = a.chunk_beg[item];
            end = a.chunk_end[item];
        } else {
            row = (int)(item - a.n_chunks);
            beg = a.rowptr[row];
            end = a.rowptr[row + 1];
            if (end - beg > a.threshold) {   // long row: its chunks are other items of this launch, then k_spmm_narrow_reduce
                end = beg;
                row = -2;
            }
        }
    }
    const float *__restrict__ Xl = a.X + gl * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = beg; base < end; base += G) {
        const int n = min(G, end - base);  // uniform inside a group
        int c = 0;
        float v = 0.0f;
        if (gl < n) {
            c = a.colidx[base + gl];
            v = a.vals[base + gl];
        }
        for (int j = 0; j < n; j += UNROLL) {
            float4 x[UNROLL];
            float vv[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int jj = min(j + u, n - 1);
                const int cj = __shfl(c, jj, G);
                vv[u] = __shfl(v, jj, G);
                x[u] = *reinterpret_cast<const float4 *>(Xl + (size_t)cj * D);
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                if (j + u < n) {
                    acc.x = fmaf(vv[u], x[u].x, acc.x);
                    acc.y = fmaf(vv[u], x[u].y, acc.y);
                    acc.z = fmaf(vv[u], x[u].z, acc.z);
                    acc.w = fmaf(vv[u], x[u].w, acc.w);
                }
            }
        }
    }
    if (!valid)
        return;
    const float y[4] = {acc.x, acc.y, acc.z, acc.w};
    if (item < a.n_chunks)
        store_vec<4>(a.ws + (size_t)item * D + gl * 4, y);
    else if (row >= 0)
        epilogue<4>(a, (size_t)row * D + gl * 4, y);
}

// one thread per (split row, column): y = ((p0 + p1) + p2) + ... in chunk order, then the usual epilogue
__global__ __launch_bounds__(256) void k_spmm_narrow_reduce(const SpmmArgs a, const int *__restrict__ long_rows,
                                                            const int *__restrict__ long_chunk_ptr, int n_long)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int l = (int)(t / a.d), col = (int)(t % a.d);
    if (l >= n_long)
        return;
    const int c0 = long_chunk_ptr[l], c1 = long_chunk_ptr[l + 1];
    float yv = a.ws[(size_t)c0 * a.d + col];
    for (int c = c0 + 1; c < c1; ++c)
        yv = yv + a.ws[(size_t)c * a.d + col];
    const float y[1] = {yv};
    epilogue<1>(a, (size_t)long_rows[l] * a.d + col, y);
}

// XCD-affine segmented launch (tgcn_spmm_segmented_f32).  Waves [0, n_tiles) walk one tile each of the plan's
// own (column, value) streams: 64 pairs are loaded at a time (the next 64 while the current ones are consumed),
// rows of X are gathered 16 or 32 at a time, and every entry whose bit is set in the stream's flag words
// closes a piece: the running sum goes to the next workspace slot and restarts from +0.  The branch on the
// flag is scalar (the flag word is wave-uniform).  The remaining waves own one direct row each.
struct SegArgs {
    const int2 *__restrict__ tile_meta;  // {first slot, entries}
    const int *__restrict__ ent_col;
    const float *__restrict__ ent_val;
    const unsigned long long *__restrict__ ent_flags;  // bit i of word w: entry 64 w + i closes a piece
    const int *__restrict__ direct_rows;
    int n_tiles;
    int tile_entries;
    int n_direct;
};

// One batch of a tile wave: UNROLL row gathers in flight, then the fmaf chain; an entry whose bit is set in `fm`
// (bit u = entry j + u closes a piece) stores the running sum to the next workspace slot and restarts the chain.
// The gathered table is < 4 GB (checked by the entry point), so a row's byte offset is one s_lshl of the column
// id and the load is SGPR base + 32-bit VGPR offset: v_readlane, s_lshl, v_add, global_load per entry.  Flags are
// tested four entries at a time: most groups of four close no piece and run four fmaf back to back.
// FULL: all UNROLL entries exist (every 64-entry slab but a stream's last one), so no per-entry bounds test.
template <int VEC, int UNROLL, bool FULL>
__device__ __forceinline__ void tile_batch(const char *__restrict__ Xb, unsigned lane_off, float *__restrict__ wsl, int c, float v,
                                           int j, int n, unsigned fm, int &slot, float (&acc)[VEC])
{
    constexpr unsigned kRowShift = VEC == 1 ? 8 : VEC == 2 ? 9 : 10;  // log2(4 * d)
    constexpr size_t d = 64 * VEC;
    float x[UNROLL][VEC];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const unsigned cj = (unsigned)__builtin_amdgcn_readlane(c, FULL ? j + u : min(j + u, n - 1));
        load_vec<VEC>(reinterpret_cast<const float *>(Xb + ((cj << kRowShift) + lane_off)), x[u]);
    }
#pragma unroll
    for (int u4 = 0; u4 < UNROLL; u4 += 4) {
        const bool any = (fm >> u4) & 0xfu;  // scalar
#pragma unroll
        for (int u = u4; u < u4 + 4; ++u) {
            if (FULL || j + u < n) {
                const float vj = readlane_f(v, j + u);
#pragma unroll
                for (int k = 0; k < VEC; ++k)
                    acc[k] = fmaf(vj, x[u][k], acc[k]);
                if (any && (fm & (1u << u))) {  // last entry of a piece
                    store_vec<VEC>(wsl + (size_t)slot * d, acc);
                    ++slot;
#pragma unroll
                    for (int k = 0; k < VEC; ++k)
                        acc[k] = 0.0f;
                }
            }
        }
    }
}

template <int VEC, int UNROLL>
__global__ __launch_bounds__(256) void k_spmm_seg(const SpmmArgs a, const SegArgs g)
{
    static_assert(UNROLL <= 32 && UNROLL % 4 == 0 && kWave % UNROLL == 0, "a batch's flag bits must fit one 32-bit word");
    const int lane = lane_id();
    const int wave = uniform(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (wave >= g.n_tiles) {
        const int w = wave - g.n_tiles;
        if (w >= g.n_direct)
            return;
        const int row = g.direct_rows[w];
        const size_t off = (size_t)row * a.d + lane * VEC;
        float t[VEC];
        preload_acc<VEC>(a, off, t);
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k)
            acc[k] = 0.0f;
        accumulate_wave<VEC, (VEC == 4 ? 8 : 16)>(a, a.rowptr[row], a.rowptr[row + 1], lane, acc);
        epilogue_pre<VEC>(a, off, acc, t);
        return;
    }
    const int2 meta = g.tile_meta[wave];
    const int n_ent = meta.y;
    if (n_ent == 0)
        return;
    int slot = meta.x;
    const size_t base = (size_t)wave * g.tile_entries;
    const int *__restrict__ ec = g.ent_col + base;
    const float *__restrict__ ev = g.ent_val + base;
    const unsigned long long *__restrict__ ef = g.ent_flags + (base >> 6);
    const char *__restrict__ Xb = reinterpret_cast<const char *>(a.X);
    const unsigned lane_off = lane * VEC * 4;
    float *__restrict__ wsl = a.ws + lane * VEC;
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k)
        acc[k] = 0.0f;
    int c = ec[lane];  // tiles are stored whole: no bounds test
    float v = ev[lane];
    unsigned long long flags = ef[0];
    for (int off = 0; off < n_ent; off += kWave) {
        const int n = min(kWave, n_ent - off);  // uniform
        int c_nxt = 0;
        float v_nxt = 0.0f;
        unsigned long long f_nxt = 0;
        if (off + kWave < n_ent) {  // uniform
            c_nxt = ec[off + kWave + lane];
            v_nxt = ev[off + kWave + lane];
            f_nxt = ef[(off >> 6) + 1];
        }
        if (n == kWave) {
#pragma unroll
            for (int j = 0; j < kWave; j += UNROLL)
                tile_batch<VEC, UNROLL, true>(Xb, lane_off, wsl, c, v, j, n, (unsigned)(flags >> j), slot, acc);
        } else {
            for (int j = 0; j < n; j += UNROLL)
                tile_batch<VEC, UNROLL, false>(Xb, lane_off, wsl, c, v, j, n, (unsigned)(flags >> j), slot, acc);
        }
        c = c_nxt, v = v_nxt, flags = f_nxt;
    }
}

template <int VEC>
int launch_seg(const SpmmArgs &a, const SegArgs &g, int unroll, int grid, hipStream_t s)
{
    // `unroll` = row gathers in flight per tile wave (direct rows keep accumulate_wave's default)
    if (unroll == 0)
        unroll = VEC == 4 ? 8 : VEC == 2 ? 16 : 32;  // measured on config 2 (d = 64): 32 in flight beats 16 by 4 %
    switch (unroll) {
        case 8: hipLaunchKernelGGL((k_spmm_seg<VEC, 8>), dim3(grid), dim3(256), 0, s, a, g); break;
        case 32:
            if constexpr (VEC == 1) {
                hipLaunchKernelGGL((k_spmm_seg<VEC, 32>), dim3(grid), dim3(256), 0, s, a, g);
                break;
            }
            [[fallthrough]];
        default: hipLaunchKernelGGL((k_spmm_seg<VEC, 16>), dim3(grid), dim3(256), 0, s, a, g); break;
    }
    return check_launch("k_spmm_seg");
}

// one wave per segmented row: y = ((p0 + p1) + p2) + ... over the row's pieces in column order, then the epilogue
template <int VEC>
__device__ __forceinline__ void seg_reduce_row(const SpmmArgs &a, const int *__restrict__ rows, const int *__restrict__ row_slot_ptr,
                                               const int *__restrict__ row_slots, int l, int lane)
{
    const int row = rows[l];
    const int s0 = row_slot_ptr[l], s1 = row_slot_ptr[l + 1];
    const size_t off = (size_t)row * a.d + lane * VEC;
    float t[VEC];
    if (a.acc_out)
        load_vec<VEC>(a.acc_in + off, t);  // early: independent of the pieces
    const float *__restrict__ wsl = a.ws + lane * VEC;
    const size_t d = (size_t)a.d;
    float y[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k)
        y[k] = 0.0f;
    bool first = true;
    for (int sb = s0; sb < s1; sb += kWave) {
        const int n_s = min(kWave, s1 - sb);
        const int sl = row_slots[sb + min(lane, n_s - 1)];
        for (int j = 0; j < n_s; j += 8) {
            float p[8][VEC];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                load_vec<VEC>(wsl + (size_t)__builtin_amdgcn_readlane(sl, min(j + u, n_s - 1)) * d, p[u]);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (j + u < n_s) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k)
                        y[k] = first ? p[u][k] : y[k] + p[u][k];
                    first = false;
                }
            }
        }
    }
    if (a.Y)
        store_vec<VEC>(a.Y + off, y);
    if (a.acc_out) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            t[k] = t[k] + y[k];
            if (a.acc_div != 1.0f)
                t[k] = t[k] / a.acc_div;
        }
        store_vec<VEC>(a.acc_out + off, t);
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void k_spmm_seg_reduce(const SpmmArgs a, const int *__restrict__ rows,
                                                         const int *__restrict__ row_slot_ptr,
                                                         const int *__restrict__ row_slots, int n)
{
    const int lane = lane_id();
    const int l = uniform(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (l >= n)
        return;
    seg_reduce_row<VEC>(a, rows, row_slot_ptr, row_slots, l, lane);
}

// The second launch of the two-launch form (flags bit 16 of tgcn_spmm_segmented_f32): the segmented rows' piece sums are
// added up by the first waves WHILE the remaining waves chain the direct rows -- the reduce is a bandwidth-bound stream (its
// pieces come back from the Infinity Cache), the direct rows are latency-bound gathers: side by side they share the launch.
template <int VEC>
__global__ __launch_bounds__(256) void k_spmm_reduce_direct(const SpmmArgs a, const int *__restrict__ rows,
                                                            const int *__restrict__ row_slot_ptr,
                                                            const int *__restrict__ row_slots, int n_red,
                                                            const int *__restrict__ direct_rows, int n_direct)
{
    const int lane = lane_id();
    const int wave = uniform(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (wave < n_red) {
        seg_reduce_row<VEC>(a, rows, row_slot_ptr, row_slots, wave, lane);
        return;
    }
    const int w = wave - n_red;
    if (w >= n_direct)
        return;
    const int row = direct_rows[w];
    const size_t off = (size_t)row * a.d + lane * VEC;
    float t[VEC];
    preload_acc<VEC>(a, off, t);
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k)
        acc[k] = 0.0f;
    accumulate_wave<VEC, (VEC == 4 ? 8 : 16)>(a, a.rowptr[row], a.rowptr[row + 1], lane, acc);
    epilogue_pre<VEC>(a, off, acc, t);
}

// ---- row groups: one wave owns up to R CONSECUTIVE rows (tgcn_spmm_groups_f32) ---------------------------------------------
// A wave that owns one short row pays four dependent round trips (rowptr -> (col, val) -> one batch of gathers -> store) for a
// dozen gathers, and its gather pipeline drains at the row's end.  The entries of consecutive rows are one contiguous range of
// colidx / vals: a wave that owns rows [first, first + n) loads their n + 1 row pointers with one instruction, issues the n
// acc_in rows at once, and walks the range 64 entries at a time with UNROLL row gathers in flight ACROSS row ends.  An entry
// that ends a row (found by comparing its offset + 1 with the n row ends, held in SGPRs; one ballot per 64 entries) parks the
// running sum in the wave's LDS rows -- lane-private slots, no barrier -- and restarts the chain from +0, so every row is still
// ONE sequential fmaf chain in column order: the same bits as one wave per row.  The Y / acc epilogues of the group's rows are
// issued together at the end.  Groups are built on the host (graph.row_groups): consecutive rows up to ~64 entries or R rows,
// never across a row the split plan cuts; handed out longest first where the launch's tail matters.
template <int VEC>
struct GroupShape {
    static constexpr int R = VEC == 4 ? 4 : 8;      // rows per group: R * VEC registers of acc_in, R * 256 * VEC bytes of LDS per wave
};

template <int VEC>
__device__ __forceinline__ void lds_store_vec(float *p, const float (&x)[VEC])
{
    if constexpr (VEC == 1)
        *p = x[0];
    else if constexpr (VEC == 2)
        *reinterpret_cast<float2 *>(p) = make_float2(x[0], x[1]);
    else
        *reinterpret_cast<float4 *>(p) = make_float4(x[0], x[1], x[2], x[3]);
}

// UNROLL gathers in flight, then the chain; bit u of `fm` set: entry j + u is the last of its row
template <int VEC, int UNROLL, bool FULL>
__device__ __forceinline__ void group_batch(const char *__restrict__ Xb, unsigned lane_off, float *__restrict__ yl, int c, float v, int j,
                                            int n, unsigned fm, unsigned &rem, float (&acc)[VEC])
{
    constexpr unsigned kRowShift = VEC == 1 ? 8 : VEC == 2 ? 9 : 10;  // log2(4 * d)
    constexpr int D = 64 * VEC;
    float x[UNROLL][VEC];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const unsigned cj = (unsigned)__builtin_amdgcn_readlane(c, FULL ? j + u : min(j + u, n - 1));
        load_vec<VEC>(reinterpret_cast<const float *>(Xb + ((cj << kRowShift) + lane_off)), x[u]);
    }
#pragma unroll
    for (int u4 = 0; u4 < UNROLL; u4 += 4) {
        const bool any = (fm >> u4) & 0xfu;  // scalar
#pragma unroll
        for (int u = u4; u < u4 + 4; ++u) {
            if (FULL || j + u < n) {
                const float vj = readlane_f(v, j + u);
#pragma unroll
                for (int k = 0; k < VEC; ++k)
                    acc[k] = fmaf(vj, x[u][k], acc[k]);
                if (any && (fm & (1u << u))) {  // last entry of the lowest non-empty row still open
                    lds_store_vec<VEC>(yl + __builtin_ctz(rem) * D, acc);
                    rem &= rem - 1;
#pragma unroll
                    for (int k = 0; k < VEC; ++k)
                        acc[k] = 0.0f;
                }
            }
        }
    }
}

template <int VEC, int UNROLL>
__device__ __forceinline__ void group_wave(const SpmmArgs &a, int first, int n, int lane, float *__restrict__ yl /* this wave's LDS rows + lane * VEC */)
{
    constexpr int R = GroupShape<VEC>::R;
    constexpr int D = 64 * VEC;
    // row pointers of the group: lane l < n holds row l's begin and end
    const int li = min(lane, n - 1);
    const int rp = a.rowptr[first + li];
    const int re = a.rowptr[first + li + 1];
    const size_t off0 = (size_t)first * D + lane * VEC;
    float t[R][VEC];
#pragma unroll
    for (int i = 0; i < R; ++i) {
#pragma unroll
        for (int k = 0; k < VEC; ++k)
            t[i][k] = 0.0f;
        if (i < n)                   // uniform
            preload_acc<VEC>(a, off0 + (size_t)i * D, t[i]);
    }
    {
        float z[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k)
            z[k] = 0.0f;
#pragma unroll
        for (int i = 0; i < R; ++i)  // rows without entries are never closed: their sum is +0
            lds_store_vec<VEC>(yl + i * D, z);
    }
    const int beg = __builtin_amdgcn_readlane(rp, 0);
    const int end = __builtin_amdgcn_readlane(re, n - 1);
    int s_re[R];
#pragma unroll
    for (int i = 0; i < R; ++i)
        s_re[i] = i < n ? __builtin_amdgcn_readlane(re, i) : -1;
    unsigned rem = (unsigned)__ballot(lane < n && re > rp);   // rows with entries, lowest = the open one
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k)
        acc[k] = 0.0f;
    if (end > beg) {
        const char *__restrict__ Xb = reinterpret_cast<const char *>(a.X);
        const unsigned lane_off = lane * VEC * 4;
        int c = 0;
        float v = 0.0f;
        if (beg + lane < end) {
            c = ld_stream(a.colidx + beg + lane);
            v = ld_stream(a.vals + beg + lane);
        }
        for (int base = beg; base < end; base += kWave) {
            const int ns = min(kWave, end - base);  // uniform
            int c_nxt = 0;
            float v_nxt = 0.0f;
            if (base + kWave + lane < end) {
                c_nxt = ld_stream(a.colidx + base + kWave + lane);
                v_nxt = ld_stream(a.vals + base + kWave + lane);
            }
            const int e1 = base + lane + 1;
            bool f = false;
#pragma unroll
            for (int i = 0; i < R; ++i)
                f |= e1 == s_re[i];
            const unsigned long long flags = __ballot(f);
            if (ns == kWave) {
#pragma unroll
                for (int j = 0; j < kWave; j += UNROLL)
                    group_batch<VEC, UNROLL, true>(Xb, lane_off, yl, c, v, j, ns, (unsigned)(flags >> j), rem, acc);
            } else {
                for (int j = 0; j < ns; j += UNROLL)
                    group_batch<VEC, UNROLL, false>(Xb, lane_off, yl, c, v, j, ns, (unsigned)(flags >> j), rem, acc);
            }
            c = c_nxt, v = v_nxt;
        }
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
        if (i < n) {  // uniform
            float y[VEC];
            if constexpr (VEC == 1) {
                y[0] = yl[i * D];
            } else if constexpr (VEC == 2) {
                const float2 q = *reinterpret_cast<const float2 *>(yl + i * D);
                y[0] = q.x, y[1] = q.y;
            } else {
                const float4 q = *reinterpret_cast<const float4 *>(yl + i * D);
                y[0] = q.x, y[1] = q.y, y[2] = q.z, y[3] = q.w;
            }
            epilogue_pre<VEC>(a, off0 + (size_t)i * D, y, t[i]);
        }
    }
}

template <int VEC, int UNROLL>
__device__ __forceinline__ void single_row_wave(const SpmmArgs &a, int row, int lane)
{
    const size_t off = (size_t)row * a.d + lane * VEC;
    float t[VEC];
    preload_acc<VEC>(a, off, t);
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k)
        acc[k] = 0.0f;
    accumulate_wave<VEC, UNROLL>(a, a.rowptr[row], a.rowptr[row + 1], lane, acc);
    epilogue_pre<VEC>(a, off, acc, t);
}

template <int VEC, int UNROLL>
__global__ __launch_bounds__(256) void k_spmm_groups(const SpmmArgs a, const int2 *__restrict__ groups, int n_groups)
{
    constexpr int R = GroupShape<VEC>::R;
    __shared__ float ybuf[4][R * 64 * VEC];
    const int lane = lane_id();
    const int wave = uniform(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (wave < a.n_chunks) {
        chunk_wave<VEC, UNROLL>(a, wave, lane);
        return;
    }
    const int w = wave - a.n_chunks;
    if (w >= n_groups)
        return;
    const int2 g = groups[w];
    if (g.y == 1) {  // a row long enough to be a group of its own (graph.row_groups single_len): the one-wave-per-row walk
        single_row_wave<VEC, UNROLL>(a, g.x, lane);
        return;
    }
    group_wave<VEC, UNROLL>(a, g.x, g.y, lane, ybuf[threadIdx.x >> 6] + lane * VEC);
}

// second launch of the two-launch segmented form with the direct rows in groups
template <int VEC>
__global__ __launch_bounds__(256) void k_spmm_reduce_groups(const SpmmArgs a, const int *__restrict__ rows,
                                                            const int *__restrict__ row_slot_ptr,
                                                            const int *__restrict__ row_slots, int n_red,
                                                            const int2 *__restrict__ groups, int n_groups)
{
    constexpr int R = GroupShape<VEC>::R;
    __shared__ float ybuf[4][R * 64 * VEC];
    const int lane = lane_id();
    const int wave = uniform(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (wave < n_red) {
        seg_reduce_row<VEC>(a, rows, row_slot_ptr, row_slots, wave, lane);
        return;
    }
    const int w = wave - n_red;
    if (w >= n_groups)
        return;
    const int2 g = groups[w];
    if (g.y == 1) {
        single_row_wave<VEC, (VEC == 4 ? 8 : 16)>(a, g.x, lane);
        return;
    }
    group_wave<VEC, (VEC == 4 ? 8 : 16)>(a, g.x, g.y, lane, ybuf[threadIdx.x >> 6] + lane * VEC);
}

template <int VEC>
int launch_groups(const SpmmArgs &a, const int2 *groups, int n_groups, int unroll, int grid, hipStream_t s)
{
    if (unroll == 0)
        unroll = VEC == 4 ? 8 : 16;
    switch (unroll) {
        case 8: hipLaunchKernelGGL((k_spmm_groups<VEC, 8>), dim3(grid), dim3(256), 0, s, a, groups, n_groups); break;
        case 32:
            if constexpr (VEC <= 2) {
                hipLaunchKernelGGL((k_spmm_groups<VEC, 32>), dim3(grid), dim3(256), 0, s, a, groups, n_groups);
                break;
            }
            [[fallthrough]];
        default: hipLaunchKernelGGL((k_spmm_groups<VEC, 16>), dim3(grid), dim3(256), 0, s, a, groups, n_groups); break;
    }
    return check_launch("k_spmm_groups");
}

// any d: wave per row, one 64-column slab at a time (re-walks the row per slab; d <= 64 is one pass)
__global__ __launch_bounds__(256) void k_spmm_generic(const SpmmArgs a)
{
    const int lane = lane_id();
    const int row = uniform(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (row >= a.n_rows)
        return;
    const int beg = a.rowptr[row];
    const int end = a.rowptr[row + 1];
    const size_t d = (size_t)a.d;
    for (int c0 = 0; c0 < a.d; c0 += kWave) {
        const int col = c0 + lane;
        const bool on = col < a.d;
        float acc = 0.0f;
        for (int base = beg; base < end; base += kWave) {
            const int n = min(kWave, end - base);
            int c = 0;
            float v = 0.0f;
            if (lane < n) {
                c = a.colidx[base + lane];
                v = a.vals[base + lane];
            }
            for (int j = 0; j < n; j += 4) {
                float x[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int cj = __builtin_amdgcn_readlane(c, min(j + u, n - 1));
                    x[u] = on ? a.X[(size_t)cj * d + col] : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (j + u < n)
                        acc = fmaf(readlane_f(v, min(j + u, n - 1)), x[u], acc);
            }
        }
        if (on) {
            const float y[1] = {acc};
            epilogue<1>(a, (size_t)row * d + col, y);
        }
    }
}

// one wave per split row: y = ((p0 + p1) + p2) + ...  in chunk order, then the usual epilogue
template <int VEC>
__global__ __launch_bounds__(256) void k_spmm_long_reduce(const SpmmArgs a, const int *__restrict__ long_rows,
                                                          const int *__restrict__ long_chunk_ptr, int n_long)
{
    const int lane = lane_id();
    const int l = uniform(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (l >= n_long)
        return;
    const int row = long_rows[l];
    const int c0 = long_chunk_ptr[l], c1 = long_chunk_ptr[l + 1];
    float t[VEC];
    preload_acc<VEC>(a, (size_t)row * a.d + lane * VEC, t);
    float y[VEC];
    load_vec<VEC>(a.ws + (size_t)c0 * a.d + lane * VEC, y);
    // loads of up to 8 chunk sums are issued together; the additions stay in chunk order
    for (int c = c0 + 1; c < c1; c += 8) {
        float p[8][VEC];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            load_vec<VEC>(a.ws + (size_t)min(c + u, c1 - 1) * a.d + lane * VEC, p[u]);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (c + u < c1)
#pragma unroll
                for (int k = 0; k < VEC; ++k)
                    y[k] = y[k] + p[u][k];
    }
    epilogue_pre<VEC>(a, (size_t)row * a.d + lane * VEC, y, t);
}

template <int VEC>
int launch_wave(const SpmmArgs &a, int unroll, int grid, hipStream_t s)
{
    if (unroll == 0)
        unroll = VEC == 4 ? 8 : 16;
    switch (unroll) {
        case 4: hipLaunchKernelGGL((k_spmm_wave<VEC, 4>), dim3(grid), dim3(256), 0, s, a); break;
        case 8: hipLaunchKernelGGL((k_spmm_wave<VEC, 8>), dim3(grid), dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((k_spmm_wave<VEC, 16>), dim3(grid), dim3(256), 0, s, a); break;
    }
    return check_launch("k_spmm_wave");
}

}  // namespace
}  // namespace tgcn

using namespace tgcn;

extern "C" int tgcn_spmm_csr_f32(const int32_t *rowptr, const int32_t *colidx, const float *vals, int64_t n_rows,
                                 const float *X, int64_t n_src_rows, int32_t d, float *Y, const float *acc_in,
                                 float *acc_out, float acc_div, const tgcn_split_plan_t *plan, const int32_t *row_order,
                                 uint32_t flags, tgcn_stream_t stream)
{
    TGCN_REQUIRE(n_rows >= 0 && n_rows < INT_MAX - 256, "n_rows out of range");
    TGCN_REQUIRE(d > 0 && d <= 4096, "d out of range");
    TGCN_REQUIRE(n_src_rows >= 0 && n_src_rows < INT_MAX, "n_src_rows out of range");
    if (n_rows == 0)
        return TGCN_OK;
    TGCN_REQUIRE(rowptr && X, "rowptr / X is NULL");
    TGCN_REQUIRE(colidx && vals, "colidx / vals is NULL");
    TGCN_REQUIRE(Y || acc_out, "both Y and acc_out are NULL: nothing to compute");
    TGCN_REQUIRE(!acc_out || acc_in, "acc_out given without acc_in");
    TGCN_REQUIRE(acc_div != 0.0f, "acc_div must be non-zero");
    hipStream_t s = static_cast<hipStream_t>(stream);

    const int variant = flags & 0xff;
    const int unroll = (flags >> 8) & 0xff;
    const bool vec_ok = (d == 64 || d == 128 || d == 256);
    // one kernel shape: a wave per row with 16 row loads in flight.  (A 16-lane-group-per-row layout was measured within 5 %
    // on every graph tried and never ahead -- profiles/r01_spmm_variants.md -- and was removed.)
    TGCN_REQUIRE(variant == TGCN_SPMM_AUTO || variant == TGCN_SPMM_WAVE_PER_ROW, "unknown kernel variant");

    SpmmArgs a;
    a.rowptr = rowptr, a.colidx = colidx, a.vals = vals, a.X = X, a.Y = Y;
    a.acc_in = acc_in, a.acc_out = acc_out, a.acc_div = acc_div;
    a.n_rows = (int)n_rows, a.d = d;
    a.threshold = INT_MAX, a.n_chunks = 0, a.chunk_beg = nullptr, a.chunk_end = nullptr, a.ws = nullptr;
    a.row_order = row_order;
    const bool narrow_ok = (d == 8 || d == 16 || d == 32);
    const bool split = plan && plan->n_chunks > 0 && (vec_ok || narrow_ok);
    if (split) {
        TGCN_REQUIRE(plan->threshold > 0, "plan->threshold must be positive");
        TGCN_REQUIRE(plan->chunk_beg && plan->chunk_end && plan->long_rows && plan->long_chunk_ptr && plan->workspace,
                     "split plan has NULL members");
        TGCN_REQUIRE(plan->n_long > 0, "split plan has chunks but no long rows");
        a.threshold = plan->threshold, a.n_chunks = plan->n_chunks;
        a.chunk_beg = plan->chunk_beg, a.chunk_end = plan->chunk_end, a.ws = plan->workspace;
    }

    int rc;
    if (narrow_ok) {
        const long long items = (long long)a.n_chunks + a.n_rows;
        const long long grid = (items * (d / 4) + 255) / 256;
        TGCN_REQUIRE(grid < INT_MAX, "too many rows for one narrow launch");
        if (d == 32)
            hipLaunchKernelGGL((k_spmm_narrow<8, 8>), dim3((unsigned)grid), dim3(256), 0, s, a);
        else if (d == 16)
            hipLaunchKernelGGL((k_spmm_narrow<4, 4>), dim3((unsigned)grid), dim3(256), 0, s, a);
        else
            hipLaunchKernelGGL((k_spmm_narrow<2, 2>), dim3((unsigned)grid), dim3(256), 0, s, a);
        if ((rc = check_launch("k_spmm_narrow")) != TGCN_OK)
            return rc;
        if (split) {
            const long long rgrid = ((long long)plan->n_long * d + 255) / 256;
            hipLaunchKernelGGL(k_spmm_narrow_reduce, dim3((unsigned)rgrid), dim3(256), 0, s, a, plan->long_rows, plan->long_chunk_ptr, plan->n_long);
            rc = check_launch("k_spmm_narrow_reduce");
        }
        return rc;
    }
    if (!vec_ok) {
        a.row_waves = a.n_rows;
        const int grid = (a.n_rows + 3) / 4;
        hipLaunchKernelGGL(k_spmm_generic, dim3(grid), dim3(256), 0, s, a);
        return check_launch("k_spmm_generic");
    }
    a.row_waves = a.n_rows;
    {
        const int grid = (a.row_waves + a.n_chunks + 3) / 4;
        rc = d == 64 ? launch_wave<1>(a, unroll, grid, s) : d == 128 ? launch_wave<2>(a, unroll, grid, s) : launch_wave<4>(a, unroll, grid, s);
    }
    if (rc != TGCN_OK)
        return rc;
    if (split) {
        const int grid = (plan->n_long + 3) / 4;
        if (d == 64)
            hipLaunchKernelGGL((k_spmm_long_reduce<1>), dim3(grid), dim3(256), 0, s, a, plan->long_rows, plan->long_chunk_ptr, plan->n_long);
        else if (d == 128)
            hipLaunchKernelGGL((k_spmm_long_reduce<2>), dim3(grid), dim3(256), 0, s, a, plan->long_rows, plan->long_chunk_ptr, plan->n_long);
        else
            hipLaunchKernelGGL((k_spmm_long_reduce<4>), dim3(grid), dim3(256), 0, s, a, plan->long_rows, plan->long_chunk_ptr, plan->n_long);
        rc = check_launch("k_spmm_long_reduce");
    }
    return rc;
}

extern "C" int tgcn_spmm_groups_f32(const int32_t *rowptr, const int32_t *colidx, const float *vals, int64_t n_rows,
                                    const float *X, int64_t n_src_rows, int32_t d, float *Y, const float *acc_in,
                                    float *acc_out, float acc_div, const tgcn_split_plan_t *plan, const int32_t *groups,
                                    int64_t n_groups, uint32_t flags, tgcn_stream_t stream)
{
    TGCN_REQUIRE(n_rows >= 0 && n_rows < INT_MAX - 256, "n_rows out of range");
    TGCN_REQUIRE(d == 64 || d == 128 || d == 256, "row groups support d in {64, 128, 256}");
    TGCN_REQUIRE(n_src_rows >= 0 && n_src_rows < INT_MAX, "n_src_rows out of range");
    TGCN_REQUIRE(n_groups >= 0 && n_groups <= n_rows, "n_groups out of range");
    if (n_rows == 0)
        return TGCN_OK;
    TGCN_REQUIRE(rowptr && X && colidx && vals, "rowptr / colidx / vals / X is NULL");
    TGCN_REQUIRE(n_groups == 0 || groups, "groups is NULL");
    TGCN_REQUIRE(Y || acc_out, "both Y and acc_out are NULL: nothing to compute");
    TGCN_REQUIRE(!acc_out || acc_in, "acc_out given without acc_in");
    TGCN_REQUIRE(acc_div != 0.0f, "acc_div must be non-zero");
    TGCN_REQUIRE((uint64_t)n_src_rows * (uint64_t)d * 4u < (1ull << 32), "row groups need a gathered table below 4 GB");
    hipStream_t s = static_cast<hipStream_t>(stream);
    SpmmArgs a;
    a.rowptr = rowptr, a.colidx = colidx, a.vals = vals, a.X = X, a.Y = Y;
    a.acc_in = acc_in, a.acc_out = acc_out, a.acc_div = acc_div;
    a.n_rows = (int)n_rows, a.d = d, a.row_waves = 0, a.row_order = nullptr;
    a.threshold = INT_MAX, a.n_chunks = 0, a.chunk_beg = nullptr, a.chunk_end = nullptr, a.ws = nullptr;
    const bool split = plan && plan->n_chunks > 0;
    if (split) {
        TGCN_REQUIRE(plan->threshold > 0, "plan->threshold must be positive");
        TGCN_REQUIRE(plan->chunk_beg && plan->chunk_end && plan->long_rows && plan->long_chunk_ptr && plan->workspace,
                     "split plan has NULL members");
        TGCN_REQUIRE(plan->n_long > 0, "split plan has chunks but no long rows");
        a.threshold = plan->threshold, a.n_chunks = plan->n_chunks;
        a.chunk_beg = plan->chunk_beg, a.chunk_end = plan->chunk_end, a.ws = plan->workspace;
    }
    const int unroll = (flags >> 8) & 0xff;
    const long long waves = (long long)a.n_chunks + n_groups;
    const int grid = (int)((waves + 3) / 4);
    const int2 *g2 = reinterpret_cast<const int2 *>(groups);
    int rc = TGCN_OK;
    if (grid > 0)
        rc = d == 64 ? launch_groups<1>(a, g2, (int)n_groups, unroll, grid, s)
                     : d == 128 ? launch_groups<2>(a, g2, (int)n_groups, unroll, grid, s) : launch_groups<4>(a, g2, (int)n_groups, unroll, grid, s);
    if (rc != TGCN_OK)
        return rc;
    if (split) {
        const int rgrid = (plan->n_long + 3) / 4;
        if (d == 64)
            hipLaunchKernelGGL((k_spmm_long_reduce<1>), dim3(rgrid), dim3(256), 0, s, a, plan->long_rows, plan->long_chunk_ptr, plan->n_long);
        else if (d == 128)
            hipLaunchKernelGGL((k_spmm_long_reduce<2>), dim3(rgrid), dim3(256), 0, s, a, plan->long_rows, plan->long_chunk_ptr, plan->n_long);
        else
            hipLaunchKernelGGL((k_spmm_long_reduce<4>), dim3(rgrid), dim3(256), 0, s, a, plan->long_rows, plan->long_chunk_ptr, plan->n_long);
        rc = check_launch("k_spmm_long_reduce");
    }
    return rc;
}

extern "C" int tgcn_spmm_segmented_f32(const tgcn_segment_plan_t *plan, const int32_t *rowptr, const int32_t *colidx,
                                       const float *vals, int64_t n_rows, const float *X, int64_t n_src_rows, int32_t d,
                                       float *Y, const float *acc_in, float *acc_out, float acc_div, uint32_t flags,
                                       tgcn_stream_t stream)
{
    TGCN_REQUIRE(plan, "plan is NULL");
    TGCN_REQUIRE(n_rows >= 0 && n_rows < INT_MAX - 256, "n_rows out of range");
    TGCN_REQUIRE(d == 64 || d == 128 || d == 256, "segmented SpMM supports d in {64, 128, 256}");
    TGCN_REQUIRE(n_src_rows >= 0 && n_src_rows < INT_MAX, "n_src_rows out of range");
    if (n_rows == 0)
        return TGCN_OK;
    TGCN_REQUIRE(rowptr && colidx && vals && X, "rowptr / colidx / vals / X is NULL");
    TGCN_REQUIRE(Y || acc_out, "both Y and acc_out are NULL: nothing to compute");
    TGCN_REQUIRE(!acc_out || acc_in, "acc_out given without acc_in");
    TGCN_REQUIRE(acc_div != 0.0f, "acc_div must be non-zero");
    TGCN_REQUIRE(plan->n_tiles >= 0 && plan->n_tiles % 4 == 0, "n_tiles must be a multiple of 4");
    TGCN_REQUIRE(plan->tile_entries > 0 && plan->tile_entries % 64 == 0, "tile_entries must be a positive multiple of 64");
    TGCN_REQUIRE((int64_t)plan->n_tiles * plan->tile_entries < INT_MAX, "segment streams too large");
    TGCN_REQUIRE(plan->n_seg_rows >= 0 && plan->n_direct_rows >= 0 && plan->n_slots >= 0 && plan->n_direct_groups >= 0, "negative plan counts");
    TGCN_REQUIRE((int64_t)plan->n_seg_rows + plan->n_direct_rows == n_rows, "plan does not cover every row once");
    TGCN_REQUIRE(plan->n_tiles == 0 || (plan->tile_meta && plan->ent_col && plan->ent_val && plan->ent_flags && plan->workspace),
                 "segment stream arrays are NULL");
    TGCN_REQUIRE(plan->n_tiles == 0 || (uint64_t)n_src_rows * (uint64_t)d * 4u < (1ull << 32),
                 "segmented SpMM needs a gathered table below 4 GB (it is meant for tables of a few L2 sizes)");
    TGCN_REQUIRE(plan->n_seg_rows == 0 || (plan->seg_rows && plan->row_slot_ptr && plan->row_slots && plan->workspace),
                 "segment row arrays are NULL");
    TGCN_REQUIRE(plan->n_direct_rows == 0 || plan->direct_rows, "direct_rows is NULL");
    hipStream_t s = static_cast<hipStream_t>(stream);

    SpmmArgs a;
    a.rowptr = rowptr, a.colidx = colidx, a.vals = vals, a.X = X, a.Y = Y;
    a.acc_in = acc_in, a.acc_out = acc_out, a.acc_div = acc_div;
    a.n_rows = (int)n_rows, a.d = d, a.row_waves = 0, a.row_order = nullptr;
    a.threshold = INT_MAX, a.n_chunks = 0, a.chunk_beg = nullptr, a.chunk_end = nullptr, a.ws = plan->workspace;
    SegArgs g;
    g.tile_meta = reinterpret_cast<const int2 *>(plan->tile_meta), g.ent_col = plan->ent_col, g.ent_val = plan->ent_val;
    g.ent_flags = reinterpret_cast<const unsigned long long *>(plan->ent_flags);
    g.direct_rows = plan->direct_rows;
    g.n_tiles = plan->n_tiles, g.tile_entries = plan->tile_entries, g.n_direct = plan->n_direct_rows;
    const int unroll = (flags >> 8) & 0xff;
    const bool two_phase = (flags >> 16) & 1u;       // tiles alone, then piece reduce + direct rows side by side
    if (two_phase && plan->n_seg_rows > 0) {
        const int n_direct = g.n_direct;
        g.n_direct = 0;
        const int grid = (g.n_tiles + 3) / 4;
        if (grid > 0) {
            const int rc = d == 64 ? launch_seg<1>(a, g, unroll, grid, s) : d == 128 ? launch_seg<2>(a, g, unroll, grid, s)
                                                                                    : launch_seg<4>(a, g, unroll, grid, s);
            if (rc != TGCN_OK)
                return rc;
        }
        if (plan->n_direct_groups > 0) {   // the direct rows in groups of consecutive rows (graph.row_groups)
            TGCN_REQUIRE(plan->direct_groups, "direct_groups is NULL");
            TGCN_REQUIRE((uint64_t)n_src_rows * (uint64_t)d * 4u < (1ull << 32), "row groups need a gathered table below 4 GB");
            const int2 *g2 = reinterpret_cast<const int2 *>(plan->direct_groups);
            const int ggrid = (plan->n_seg_rows + plan->n_direct_groups + 3) / 4;
            if (d == 64)
                hipLaunchKernelGGL((k_spmm_reduce_groups<1>), dim3(ggrid), dim3(256), 0, s, a, plan->seg_rows, plan->row_slot_ptr, plan->row_slots, plan->n_seg_rows, g2, plan->n_direct_groups);
            else if (d == 128)
                hipLaunchKernelGGL((k_spmm_reduce_groups<2>), dim3(ggrid), dim3(256), 0, s, a, plan->seg_rows, plan->row_slot_ptr, plan->row_slots, plan->n_seg_rows, g2, plan->n_direct_groups);
            else
                hipLaunchKernelGGL((k_spmm_reduce_groups<4>), dim3(ggrid), dim3(256), 0, s, a, plan->seg_rows, plan->row_slot_ptr, plan->row_slots, plan->n_seg_rows, g2, plan->n_direct_groups);
            return check_launch("k_spmm_reduce_groups");
        }
        const int rgrid = (plan->n_seg_rows + n_direct + 3) / 4;
        if (d == 64)
            hipLaunchKernelGGL((k_spmm_reduce_direct<1>), dim3(rgrid), dim3(256), 0, s, a, plan->seg_rows, plan->row_slot_ptr, plan->row_slots, plan->n_seg_rows, plan->direct_rows, n_direct);
        else if (d == 128)
            hipLaunchKernelGGL((k_spmm_reduce_direct<2>), dim3(rgrid), dim3(256), 0, s, a, plan->seg_rows, plan->row_slot_ptr, plan->row_slots, plan->n_seg_rows, plan->direct_rows, n_direct);
        else
            hipLaunchKernelGGL((k_spmm_reduce_direct<4>), dim3(rgrid), dim3(256), 0, s, a, plan->seg_rows, plan->row_slot_ptr, plan->row_slots, plan->n_seg_rows, plan->direct_rows, n_direct);
        return check_launch("k_spmm_reduce_direct");
    }
    const int grid = (g.n_tiles + g.n_direct + 3) / 4;
    if (grid > 0) {
        const int rc = d == 64 ? launch_seg<1>(a, g, unroll, grid, s) : d == 128 ? launch_seg<2>(a, g, unroll, grid, s)
                                                                                : launch_seg<4>(a, g, unroll, grid, s);
        if (rc != TGCN_OK)
            return rc;
    }
    if (plan->n_seg_rows > 0) {
        const int rgrid = (plan->n_seg_rows + 3) / 4;
        if (d == 64)
            hipLaunchKernelGGL((k_spmm_seg_reduce<1>), dim3(rgrid), dim3(256), 0, s, a, plan->seg_rows, plan->row_slot_ptr, plan->row_slots, plan->n_seg_rows);
        else if (d == 128)
            hipLaunchKernelGGL((k_spmm_seg_reduce<2>), dim3(rgrid), dim3(256), 0, s, a, plan->seg_rows, plan->row_slot_ptr, plan->row_slots, plan->n_seg_rows);
        else
            hipLaunchKernelGGL((k_spmm_seg_reduce<4>), dim3(rgrid), dim3(256), 0, s, a, plan->seg_rows, plan->row_slot_ptr, plan->row_slots, plan->n_seg_rows);
        return check_launch("k_spmm_seg_reduce");
    }
    return TGCN_OK;
}
