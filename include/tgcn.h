/*
 * tgcn.h -- C ABI of the MI355X-native LightGCN propagation + scoring path.
 *
 * This is the drop-in boundary.  The reference (sergey-volokhin/TextGCN) has no FFI:
 * its plug-in API is Python subclass/override, and every numeric step is a torch ATen call.  Each
 * entry point below replaces the ATen call(s) named in its comment (file:line in /root/reference);
 * the Python class textgcn_amd.LightGCN keeps the reference's BaseModel method surface on top of it
 * (INTEGRATION.md shows the binding a TextGCN maintainer would add).
 *
 * Conventions
 *   - plain pointers + sizes only; all pointers are DEVICE pointers unless the name says `_host`;
 *   - every buffer is caller-owned; the library allocates nothing and keeps no state between calls;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the null stream) and the
 *     call returns without synchronising;
 *   - return value: TGCN_OK (0) or a negative error code; tgcn_last_error() gives the message of the
 *     last failure on the calling thread;
 *   - matrices are row-major fp32; index arrays are int32 unless stated (all reference configs have
 *     N = |U|+|I| < 2^24 and nnz(A) < 2^31).
 */
#ifndef TGCN_H_
#define TGCN_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TGCN_ABI_VERSION 9

#define TGCN_OK 0
#define TGCN_ERR_ARG (-1)         /* bad argument (null pointer, size, unsupported d/k ...) */
#define TGCN_ERR_HIP (-2)         /* a HIP runtime call failed */
#define TGCN_ERR_UNSUPPORTED (-3) /* valid request this build has no kernel for */

typedef void *tgcn_stream_t; /* hipStream_t */

int tgcn_abi_version(void);
const char *tgcn_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Long-row split plan (optional).  A row whose stored-entry count exceeds `threshold` is cut into
 * chunks [chunk_beg[c], chunk_end[c]) of at most `threshold` consecutive entries; long row l is matrix row
 * long_rows[l] and owns chunks
 * [long_chunk_ptr[l], long_chunk_ptr[l+1]).  Each chunk is summed sequentially by one wavefront into
 * workspace[c, 0:d]; a second launch adds a row's chunk sums in chunk order (deterministic, no
 * atomics).  Rows at or below the threshold are summed exactly as the reference's CPU kernel does.
 * Built on the host by textgcn_amd.graph.SplitPlan from rowptr.
 */
typedef struct tgcn_split_plan {
    int32_t threshold;
    int32_t n_chunks;
    int32_t n_long;
    int32_t _pad;
    const int32_t *chunk_beg;      /* [n_chunks] offsets into colidx/vals */
    const int32_t *chunk_end;      /* [n_chunks] */
    const int32_t *long_rows;      /* [n_long]   local row ids */
    const int32_t *long_chunk_ptr; /* [n_long+1] */
    float *workspace;              /* [n_chunks, d] fp32 scratch */
} tgcn_split_plan_t;

/* kernel selection for tgcn_spmm_csr_f32 (`flags & 0xff`); bits 8..15 = row gathers in flight per wave (0: default) */
#define TGCN_SPMM_AUTO 0
#define TGCN_SPMM_WAVE_PER_ROW 1  /* one wave64 per row, lane = d/64 consecutive columns (what AUTO selects) */

/* K1 + K3: one LightGCN layer  Y = A . X  with the layer combination fused in.
 *   replaces torch.sparse.mm(norm_matrix, emb_matrix)              TextGCN/base_model.py:148
 *   and the running part of torch.mean(torch.stack(vectors), 0)    TextGCN/base_model.py:157
 * A is CSR over `n_rows` local rows (rowptr[0..n_rows], absolute offsets into colidx/vals; column
 * ids index rows of X; entries of a row in ascending column order = the reference's coalesced COO
 * order, TextGCN/dataset.py:138).  X is [n_src_rows, d].
 *   y[r,:]        = sum over the row's entries, in order, of fmaf(val, X[col,:], y)   (exact fp32 chain)
 *   Y[r,:]        = y                                   if Y       != NULL
 *   acc_out[r,:]  = (acc_in[r,:] + y) / acc_div         if acc_out != NULL  (acc_div == 1: no division)
 * acc_in may alias acc_out.  Layer k of K:  acc_in = E0 rows (k = 1) or the running sum, acc_div =
 * K+1 on the last layer -- exactly the reference's sequential sum then one division.
 * `plan` may be NULL (every row summed by one wave: bit-identical to the reference's CPU result).
 * `row_order` (optional, [n_rows], a permutation of the row ids): the order in which rows are handed to
 * wavefronts.  Results do not depend on it; descending row length (longest work first) shortens the launch's
 * tail -- 6 % on BASELINE config 2.  Honoured by the wave-per-row kernel.
 * d in {8, 16, 32} (a rank's share of the columns under the feature partition, dist.ColumnShardedPropagator) runs the
 * narrow form: d/4 lanes per source row, 64 / (d/4) entries per wave instruction, the same chains and long-row chunks. */
int tgcn_spmm_csr_f32(const int32_t *rowptr, const int32_t *colidx, const float *vals, int64_t n_rows,
                      const float *X, int64_t n_src_rows, int32_t d, float *Y, const float *acc_in,
                      float *acc_out, float acc_div, const tgcn_split_plan_t *plan, const int32_t *row_order,
                      uint32_t flags, tgcn_stream_t stream);

/* The same product with the rows at or below the split threshold handed out in GROUPS of consecutive rows (ABI v9; d in
 * {64, 128, 256}; gathered table below 4 GB).  groups [n_groups][2] = {first row, rows}: 1 <= rows <= 8 (d = 256: 4); every row
 * that the split plan does not cut must lie in exactly one group, no cut row in any.  One wavefront owns a group: the n + 1 row
 * pointers are one load, the entries of consecutive rows one contiguous range walked with the row gathers in flight across row
 * ends, the acc_in rows are requested before anything else and the n epilogues issued together.  Every row is still one
 * sequential fmaf chain from +0 in column order: results are bit-identical to tgcn_spmm_csr_f32 (rows cut by `plan` as there).
 * Groups are handed to wavefronts in array order (after the plan's chunks): longest first shortens the launch's tail.
 * Built on the host by textgcn_amd.graph.row_groups.  flags: bits 8..15 = row gathers in flight per wavefront (0: default). */
int tgcn_spmm_groups_f32(const int32_t *rowptr, const int32_t *colidx, const float *vals, int64_t n_rows,
                         const float *X, int64_t n_src_rows, int32_t d, float *Y, const float *acc_in,
                         float *acc_out, float acc_div, const tgcn_split_plan_t *plan, const int32_t *groups,
                         int64_t n_groups, uint32_t flags, tgcn_stream_t stream);

/* XCD-affine segmented form of the same product (d in {64, 128, 256}).  The plan holds its own copy of the
 * entries of the segmented rows, as streams: stream x of a row range keeps the entries whose column falls in column
 * blocks x, x+8, ... ordered by (block, row, column).  A row's run inside one block is a segment; streams are cut
 * into tiles of `tile_entries` entries, one wavefront each, and a segment crossing a cut becomes two pieces.  The
 * last entry of a piece has its bit set in `ent_flags`; the wavefront sums a piece as a sequential fmaf chain
 * from +0 into workspace[slot] (slots count pieces in tile order from tile_meta[t][0]).  A second launch adds the
 * pieces row_slots[row_slot_ptr[i] .. row_slot_ptr[i+1]) of row seg_rows[i] in that (column) order and applies the
 * Y / acc epilogue.  Tiles are laid out so that a workgroup (4 consecutive tiles) stays in one stream and workgroups
 * g, g+8, g+16 ... share a stream: workgroups are dealt round-robin over the 8 XCDs, so every XCD's 4 MB L2 serves
 * 1/8 of the gathered table instead of missing to the Infinity Cache.  Placement is a speed matter only.
 * `direct_rows` are summed by one wavefront each from rowptr/colidx/vals exactly as in tgcn_spmm_csr_f32.
 * Deterministic; differs from the one-chain-per-row result in rounding only (same contract as the long-row split).
 * Built on the host by textgcn_amd.graph.segment_plan_arrays.
 * flags: bits 8..15 = row gathers in flight per tile wavefront (0: default); bit 16 = the tiles run as a launch of their own
 * (every L2 then holds only its block of the gathered table) and ONE further launch adds up the pieces while its other
 * wavefronts chain the direct rows -- same results bit for bit, config 2 6 % faster. */
typedef struct tgcn_segment_plan {
    int32_t n_tiles;      /* multiple of 4 */
    int32_t tile_entries; /* multiple of 64 */
    int32_t n_seg_rows;
    int32_t n_direct_rows;
    int32_t n_slots;
    int32_t _pad;
    const int32_t *tile_meta;    /* [n_tiles][2]: {first slot, entries in the tile (0: padding)} */
    const int32_t *ent_col;      /* [n_tiles * tile_entries] column ids */
    const float *ent_val;        /* [n_tiles * tile_entries] */
    const uint64_t *ent_flags;   /* [n_tiles * tile_entries / 64]: bit i of word w set = entry 64 w + i is the last of its piece */
    const int32_t *seg_rows;     /* [n_seg_rows] local row ids */
    const int32_t *row_slot_ptr; /* [n_seg_rows + 1] */
    const int32_t *row_slots;    /* [n_slots] a row's pieces in column order */
    const int32_t *direct_rows;  /* [n_direct_rows] local row ids */
    float *workspace;            /* [n_slots, d] fp32 scratch */
    const int32_t *direct_groups; /* ABI v9, optional: [n_direct_groups][2] = {first row, rows}: the direct rows as groups of
                                   * consecutive rows (see tgcn_spmm_groups_f32), each direct row in exactly one group; used by
                                   * the two-launch form instead of one wavefront per direct row.  Same bits. */
    int32_t n_direct_groups;      /* 0: one wavefront per direct row */
    int32_t _pad2;
} tgcn_segment_plan_t;

int tgcn_spmm_segmented_f32(const tgcn_segment_plan_t *plan, const int32_t *rowptr, const int32_t *colidx,
                            const float *vals, int64_t n_rows, const float *X, int64_t n_src_rows, int32_t d,
                            float *Y, const float *acc_in, float *acc_out, float acc_div, uint32_t flags,
                            tgcn_stream_t stream);

/* K5: S[b, i] = <U[user_ids[b], :], It[i, :]>  (user_ids == NULL: U rows 0..B-1), S row stride lds.
 *   replaces torch.matmul(users_emb, items_emb.t())                TextGCN/base_model.py:179
 *   (+ the users_emb[batch_users] gather at base_model.py:254)
 * fp32 MFMA (v_mfma_f32_32x32x2_f32): each dot product is the k-ordered fmaf chain from +0. */
int tgcn_score_dense_f32(const float *U, const int64_t *user_ids, int32_t B, const float *It, int32_t I,
                         int32_t d, float *S, int64_t lds, tgcn_stream_t stream);

/* K6: S[b, mask_items[e]] = -inf for e in [mask_rowptr[b], mask_rowptr[b+1]).
 *   replaces the pandas explode + advanced-index assignment        TextGCN/base_model.py:257-258 */
int tgcn_mask_f32(float *S, int64_t lds, int32_t B, int32_t I, const int32_t *mask_rowptr,
                  const int32_t *mask_items, tgcn_stream_t stream);

/* K7 (+K8): per-row top-k of S[B, I], ordered by (value descending, index ascending).
 *   replaces torch.topk(rating, k=max(self.k))                      TextGCN/base_model.py:261
 *   round4 != 0 also applies probs.round(decimals=4)                TextGCN/base_model.py:263
 *   (ATen: nearbyintf(x * 1e4f) / 1e4f).  1 <= k <= 64, k <= I.
 * NaN scores: the order is built on `>` / `==`, so a NaN never enters a list (it ranks below -inf), whereas torch.topk
 * ranks NaN first.  The reference asserts a NaN-free loss every step (base_model.py:123) and the path produces none from
 * finite inputs; callers that can hold NaN scores must clean them first.  A row with fewer than k non-NaN scores leaves
 * (-inf, TGCN_NO_ITEM) in the list positions nothing could fill: NOT an index -- check before using the ids as positions. */
#define TGCN_NO_ITEM 2147483647
int tgcn_topk_f32(const float *S, int64_t lds, int32_t B, int32_t I, int32_t k, int32_t round4,
                  float *out_val, int64_t *out_idx, tgcn_stream_t stream);

/* K5+K6+K7+K8 fused: out = top-k over items of the masked scores of B users, without materialising [B, I].
 *   replaces, per predict batch: torch.matmul + explode/-inf scatter + torch.topk + round
 *                                                                  TextGCN/base_model.py:254-263
 * Users are U[user_ids[b], :] (user_ids == NULL: rows 0..B-1); mask_rowptr/mask_items is a CSR over the B
 * batch rows of each user's train items, ascending (mask_rowptr == NULL: no mask).  Result identical to
 * tgcn_score_dense_f32 -> tgcn_mask_f32 -> tgcn_topk_f32 on the same inputs (same k-ordered fmaf chains, same
 * (value desc, index asc) order; masked items appear, with -inf, only when fewer than k items are unmasked).
 * `workspace`: device scratch of at least tgcn_score_topk_workspace_bytes(B, I, d, k) bytes, 256-byte aligned,
 * caller-owned; nothing in it needs initialising or survives the call.  1 <= k <= 64, k <= I. */
int64_t tgcn_score_topk_workspace_bytes(int32_t B, int32_t I, int32_t d, int32_t k);
int tgcn_score_topk_f32(const float *U, const int64_t *user_ids, int32_t B, const float *It, int32_t I, int32_t d,
                        const int32_t *mask_rowptr, const int32_t *mask_items, int32_t k, int32_t round4,
                        float *out_val, int64_t *out_idx, void *workspace, int64_t workspace_bytes,
                        tgcn_stream_t stream);

/* The same call with the candidates found by a bf16 matrix pass and rescored with the fp32 chains: the test
 * approx(u, i) + bound(u, i) > tau_u, with a proven bound on the bf16 error (tgcn_score_prefilter.hip), keeps a superset of
 * {i : score > tau_u}; every kept pair gets its k-ordered fp32 fmaf score and is dropped again unless score > tau_u.  Results
 * are bit-identical to tgcn_score_topk_f32 (indices, scores, tie order, fallback); only the cost of finding the candidates
 * changes.  Applies to I > 8192 and d <= 128, or d <= 1024 with d % 8 == 0 (the folded ltr_linear operands, K = 896 / 960);
 * any other shape runs tgcn_score_topk_f32's own path.  Same workspace.
 * `item_pack`: device pointer to the packed item operand written by tgcn_item_pack_bf16 for this item table (the table is
 * fixed across the batches of a predict call), or NULL: packed inside the call, into the workspace. */
int tgcn_score_topk_prefilter_f32(const float *U, const int64_t *user_ids, int32_t B, const float *It, int32_t I,
                                  int32_t d, const int32_t *mask_rowptr, const int32_t *mask_items, int32_t k,
                                  int32_t round4, const void *item_pack, float *out_val, int64_t *out_idx,
                                  void *workspace, int64_t workspace_bytes, tgcn_stream_t stream);
/* The item operand of the bf16 pass: per item row its bf16 image (round-to-nearest-even, zero-padded to the width the filter walks)
 * followed by the row's factors of the error bound, 144 (d <= 64) or 272 (d <= 128) bytes per row -- the exact bytes the filter
 * kernel stages, so a stage is one contiguous copy; wider rows (d % 8 == 0, d <= 1024) are padded to 256 / 512 / 896 / 960 / 1024
 * elements: 2 x that + 16 bytes per row.  tgcn_item_pack_bytes: size of `out` (0: no bf16 pass for this width, the
 * prefilter entry then runs the fp32 path and ignores `item_pack`; < 0: bad argument).  `out` 16-byte aligned. */
int64_t tgcn_item_pack_bytes(int32_t I, int32_t d);
int tgcn_item_pack_bf16(const float *It, int32_t I, int32_t d, void *out, tgcn_stream_t stream);
/* Diagnostic (the factors the pack carries, in fp32): out[2 i] = |It[i]|_2 (elements floored at 2^-50), out[2 i + 1] = |It[i] - bf16(It[i])|_2 (the row's rounding residual under
 * round-to-nearest-even), each times (1 + 2^-12); +inf if not finite. */
int tgcn_item_norms_f32(const float *It, int32_t I, int32_t d, float *out, tgcn_stream_t stream);
/* Diagnostic: how many of the B users of the LAST call that used `workspace` (either entry point, same B, I, d, k) took the
 * exact fallback (threshold too high, log overflow, fewer than k unmasked candidates).  Copies one int to the HOST pointer and
 * synchronises the stream.  The results never depend on it; a high count means the call paid for k_brute_part. */
int tgcn_score_topk_fallback_count(const void *workspace, int32_t B, int32_t I, int32_t d, int32_t k, int32_t *out_host,
                                   tgcn_stream_t stream);
/* Diagnostic (ABI v8): what the LAST call that used `workspace` (same B, I, d, k; `prefilter` = which entry point it was) did,
 * summed over its users -- out_host[0] users sent to the exact fallback, [1] pairs kept (fp32 score > tau_u), [2] pairs rescored by
 * an fp32 chain (the bf16 pass's candidates; wide rows: what k_refine let through), [3] pairs logged by the filter (fp32 filters
 * and the wide bf16 filter).  bench.py prices the rescoring's row gathers with [2].  Runs a small kernel, copies four int64 to
 * the HOST pointer and synchronises the stream; the results never depend on it. */
int tgcn_score_topk_stats(void *workspace, int32_t B, int32_t I, int32_t d, int32_t k, int32_t prefilter, int64_t *out_host,
                          tgcn_stream_t stream);

/* K11-K13: LTR text-feature head (config 5) folded into one GEMM of width K = tgcn_ltr_folded_width(d, t).
 *   replaces get_user_vectors / get_item_vectors / get_features_batchwise + nn.Linear(5, 1)
 *                                                                  TextGCN/ltr_models.py:95-146,181-204
 *   score = w0 e_u.e_i + w1 r_u.r_i + w2 d_u.d_i + w3 r_u.d_i + w4 d_u.r_i + b
 *         = [w0 e_u | w1 r_u + w4 d_u | w2 d_u + w3 r_u | b, 0..] . [e_i | r_i | d_i | 1, 0..]
 * fold_users writes [B, K] rows for users emb_ids[b] (rows of users_emb) / text_ids[b] (rows of the two [U, t]
 * text tables; either id array may be NULL: row b); pack_items writes [I, K].  `w5_host` is a HOST pointer to the
 * five effective weights.  Score the folded operands with tgcn_score_dense_f32 / tgcn_score_topk_f32 (d = K). */
int32_t tgcn_ltr_folded_width(int32_t d, int32_t t);
int tgcn_ltr_fold_users_f32(const float *users_emb, const float *users_reviews, const float *users_desc,
                            const int64_t *emb_ids, const int64_t *text_ids, int32_t B, int32_t d, int32_t t,
                            const float *w5_host, float bias, float *out, tgcn_stream_t stream);
int tgcn_ltr_pack_items_f32(const float *items_emb, const float *items_reviews, const float *items_desc, int32_t I,
                            int32_t d, int32_t t, float *out, tgcn_stream_t stream);
/* The five PAIRWISE features of n gathered (user, item) rows (training batches), [n, 5] row-major:
 *   replaces get_user_vectors / get_item_vectors / get_features_pairwise             TextGCN/ltr_models.py:116-128,148-166
 *   feats[p] = [e_p . e'_p, r_u . r_i, d_u . d_i, r_u . d_i, d_u . r_i],  u = users[p], i = items[p]
 * users_emb_rows / items_emb_rows are the ALREADY gathered embedding rows [n, d] (the reference's calling convention:
 * score_pairwise(users_emb[users], items_emb[items], users, items)); the text rows are read from their [U, t] / [I, t] tables by
 * id.  nn.Linear over the five features stays with the caller (it owns the trainable weights). */
int tgcn_ltr_pair_features_f32(const float *users_emb_rows, const float *items_emb_rows, const float *users_reviews,
                               const float *users_desc, const float *items_reviews, const float *items_desc, const int64_t *users,
                               const int64_t *items, int64_t n, int32_t d, int32_t t, float *feats, tgcn_stream_t stream);

/* K9: out[r] = <U[users[r], :], V[items[r], :]>   (users/items may be NULL: row r itself).
 *   replaces torch.sum(users_emb * items_emb, dim=1)               TextGCN/base_model.py:171
 *   and the gathers at base_model.py:189-193 */
int tgcn_score_pairwise_f32(const float *U, const int64_t *users, const float *V, const int64_t *items,
                            int64_t n, int32_t d, float *out, tgcn_stream_t stream);

/* N4 (dynamic negative sampling): out[b, j] = <U[users[b], :], It[cand[b, j], :]> for per-user candidate lists
 * cand [B, m] (int64 item ids); with a mask CSR over ALL users (mask_rowptr indexed by user id) a candidate that
 * is a train item of its user scores -inf.
 *   replaces torch.matmul(users_emb.unsqueeze(1), items_emb.transpose(1, 2)).squeeze()
 *                                                                  TextGCN/advanced_sampling.py:37-44
 *   and the positives filter subtract_tensor_as_set               advanced_sampling.py:64, utils.py:121-128 */
int tgcn_score_candidates_f32(const float *U, const int64_t *users, const float *It, const int64_t *cand, int32_t B,
                              int32_t m, int32_t d, const int32_t *mask_rowptr, const int32_t *mask_items, float *out,
                              tgcn_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * N1: the pieces of one BPR training step around the propagation.
 *
 * Edge dropout as value masking on the fixed CSR.
 *   replaces BaseModel._dropout_norm_matrix                        TextGCN/base_model.py:77-86
 *   (torch.rand(nnz) on the CPU, index_select, new COO + coalesce -- a device sort -- and an H->D copy per mini-batch)
 * Entry e is kept iff u_e < keep_prob (= 1 - p, base_model.py:82-83); a kept entry's value is stored_vals[e] / keep_prob (the
 * reference's fp32 division, :84), a dropped one 0.  u_e = rand_u[e] when rand_u != NULL (e.g. the reference's own CPU stream,
 * for parity), else a Philox4x32-10 draw keyed by (seed, e).  Because the draw is a function of e alone, the same launch also
 * writes, in stream order and without a second pass:
 *   vals_t[e]     = value of entry perm[e]  (the backward's matrix A^T on the same structure; the dropped matrix is not
 *                   symmetric).  stored_vals_t[e] = stored_vals[perm[e]], or NULL when that equals stored_vals[e].
 *   ent_val[s]    = value of entry ent_src[s]    -- a segment plan's stream order; ent_stored = the plan's own ent_val
 *   ent_val_t[s]  = value of entry ent_src_t[s] (= perm[ent_src[s]]); ent_stored_t as stored_vals_t.
 * Optional outputs (and the inputs only they need) may be NULL. */
int tgcn_dropout_values_f32(const float *stored_vals, const float *stored_vals_t, const float *rand_u, uint64_t seed,
                            float keep_prob, const int32_t *perm, const float *ent_stored, const float *ent_stored_t,
                            const int32_t *ent_src, const int32_t *ent_src_t, int64_t nnz, int64_t n_stream, float *vals,
                            float *vals_t, float *ent_val, float *ent_val_t, tgcn_stream_t stream);

/* BPR pairs: terms[j, r] = selu(s(u_r, n_jr) - s(u_r, p_r)), s = row dot product of the PROPAGATED tables, for b batch rows
 * (users[r], pos[r], negs[j, r]), j < m; loss = sum(terms) / (b m).
 *   replaces the gathers + score_pairwise + F.selu + mean          TextGCN/base_model.py:186-198 (:171)
 *   and their autograd backward: with gradient tables given, d loss / d users_emb, d items_emb are ADDED (float atomics)
 *   into grad_users / grad_items, which the caller zero-fills; every gradient is multiplied by grad_scale (the layer mean's
 *   1 / (K + 1), folded in) and by *upstream, a DEVICE scalar (autograd's d L / d loss; NULL = 1) -- no host round trip.
 * terms or the gradient pair may be NULL (values only / gradients only).
 * A row with users[r] < 0 is PADDING (ABI v9): its terms are 0, it adds no gradient, its other ids are not read -- ragged per-user
 * triple lists (advanced_sampling.py:61-69) can stay a dense [B x P x N] block on the device; the caller then divides by its own
 * count of real rows (fold count_all / count_real into *upstream for the gradients).  Same rule in tgcn_reg_rows_f32. */
int tgcn_bpr_pairs_f32(const float *users_emb, const float *items_emb, const int64_t *users, const int64_t *pos,
                       const int64_t *negs, int32_t b, int32_t m, int32_t d, float grad_scale, const float *upstream,
                       float *terms, float *grad_users, float *grad_items, tgcn_stream_t stream);

/* L2 term: terms[r] = |E_u[users[r]]|^2 + |E_i[pos[r]]|^2 + sum_j |E_i[negs[j, r]]|^2 on the layer-0 tables; the
 * reference's reg_loss is lambda / (2 b) * sum(terms)               TextGCN/base_model.py:200-210
 * With gradient tables given, coef * *upstream * row is ADDED to the row of every occurrence (coef = lambda / b). */
int tgcn_reg_rows_f32(const float *e_users, const float *e_items, const int64_t *users, const int64_t *pos,
                      const int64_t *negs, int32_t b, int32_t m, int32_t d, float coef, const float *upstream,
                      float *terms, float *grad_users, float *grad_items, tgcn_stream_t stream);

/* (e) Multi-GPU exchange step of the row partition: all-gather of the freshly propagated row blocks.
 * The reference is single-device -- there is no call to replace (grep nccl|torch.distributed|all_gather in TextGCN/ -> 0
 * hits); the entry point exists so that a host without torch.distributed (or a C/C++ host) can drive the sharded forward
 * with this library alone.  One communicator per process/GPU; RCCL (librccl) is bound at run time.
 *   tgcn_comm_unique_id    fills TGCN_COMM_ID_BYTES host bytes on ONE rank; the caller hands them to the other ranks
 *   tgcn_comm_init_rank    collective over all `world` ranks, on the calling thread's current device
 *   tgcn_allgather_rows    full[p * rows_local .. (p+1) * rows_local, 0:d] = rank p's local[0:rows_local, 0:d], fp32 on the
 *                          wire, enqueued on `stream`; `local` may be the rank's own block inside `full` (in place).
 *                          Every rank passes the same rows_local and d (pad blocks to the largest). */
typedef void *tgcn_comm_t; /* ncclComm_t */
#define TGCN_COMM_ID_BYTES 128
int tgcn_comm_unique_id(void *id_host);
int tgcn_comm_init_rank(tgcn_comm_t *comm, int32_t world, int32_t rank, const void *id_host);
int tgcn_comm_destroy(tgcn_comm_t comm);
int tgcn_allgather_rows(tgcn_comm_t comm, const float *local, float *full, int64_t rows_local, int32_t d,
                        tgcn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TGCN_H_ */
