// Internal helpers shared by the translation units of libtgcn.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "tgcn.h"

namespace tgcn {

void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

inline int fail_arg(const char *what)
{
    set_error("invalid argument: %s", what);
    return TGCN_ERR_ARG;
}

// checks the launch that was just enqueued (does not synchronise)
inline int check_launch(const char *kernel)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", kernel, hipGetErrorString(e));
        return TGCN_ERR_HIP;
    }
    return TGCN_OK;
}

#define TGCN_REQUIRE(cond, what) \
    do {                         \
        if (!(cond))             \
            return ::tgcn::fail_arg(what); \
    } while (0)

constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

// wave-uniform value -> SGPR (lets the compiler use scalar loads / scalar address math)
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ float readlane_f(float v, int lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}


// max over the 64 lanes, returned to every lane.  DPP steps inside the rows of 16 (quad_perm, row_half_mirror, row_mirror),
// then row_bcast15 / row_bcast31 (GFX9 family incl. gfx950) carry the row maxima into row 3: seven VALU instructions and a
// readlane, where six __shfl_xor are six ds_bpermute round trips (~100 cycles each on a serial chain).  NaNs are ignored
// (fmaxf).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_max_step(float v)
{
    const int x = __float_as_int(v);
    return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(x, x, CTRL, ROW_MASK, 0xf, false)));
}

__device__ __forceinline__ float wave_max_all(float v)
{
    v = dpp_max_step<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
    v = dpp_max_step<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
    v = dpp_max_step<0x141, 0xf>(v);   // row_half_mirror
    v = dpp_max_step<0x140, 0xf>(v);   // row_mirror: every lane holds its row's maximum
    v = dpp_max_step<0x142, 0xa>(v);   // row_bcast15 into rows 1 and 3
    v = dpp_max_step<0x143, 0xc>(v);   // row_bcast31 into rows 2 and 3
    return readlane_f(v, 63);
}

// ---- the error bound of the bf16 candidate pass (tgcn_score_prefilter.hip holds the derivation) ----------------------------
// Per row x two factors: nrm = |max(|x_j|, 2^-50)|_2 and res = |x - bf16(x)|_2 (the ACTUAL rounding residual of the conversion the
// filter performs, v_cvt_pk_bf16_f32), each with a 2^-12 margin for its own fp32 arithmetic; non-finite -> +inf.
constexpr float kNormFloor = 0x1p-50f;
constexpr float kResFloor = 0x1p-58f;       // keeps the squares of tiny residuals out of the underflow range
constexpr float kAccumBudget = 0x1p-11f;    // fp32 accumulation of the matrix pipe + the fp32 chain's own rounding, relative to |x||y|
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi)
{
    using f32x2_t = __attribute__((ext_vector_type(2))) float;
    using bf16x2_t = __attribute__((ext_vector_type(2))) __bf16;
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));   // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
}
__device__ __forceinline__ float floored_sq(float x)
{
    const float f = fmaxf(fabsf(x), kNormFloor);
    return f * f;
}
__device__ __forceinline__ float residual_sq(float x)      // (x - bf16(x))^2, floored
{
    const float r = x - __uint_as_float(pack_bf16(x, 0.0f) << 16);
    const float f = fmaxf(fabsf(r), kResFloor);
    return r == r ? f * f : INFINITY;                       // a NaN residual (non-finite x): +inf
}
__device__ __forceinline__ float bound_factor(float sq)    // sqrt with the margin; non-finite -> +inf
{
    const float f = sqrtf(sq) * (1.0f + 0x1p-12f);
    return f < INFINITY ? f : INFINITY;      // (a NaN compares false)
}
// smallest bf16 >= x for x >= 0 (or +inf), as the high half of a float: the bound's factors enter the bf16 product rounded UP
__device__ __forceinline__ unsigned bf16_up_bits(float x)
{
    return x < INFINITY ? (__float_as_uint(x) + 0xFFFFu) >> 16 : 0x7F80u;
}

// launchers shared between translation units (arguments already validated by the caller)
int launch_score_dense(const float *U, const int64_t *user_ids, int B, const float *It, int I, int d, int item_mul, float *S,
                       int64_t lds, hipStream_t stream);
int launch_score_dense_filter(const float *U, const int64_t *user_ids, int B, const float *It, int I, int d, const float *tau,
                              int tau_stride, void *logs, int *counts, int S, int items_per_split, int cap2, hipStream_t stream);
// tgcn_score_prefilter.hip: candidates from a bf16 pass, rescored in fp32
bool prefilter_supports(int d);
int launch_item_norms(const float *It, int I, int d, float *norms /* [I][2] */, hipStream_t stream);
int launch_sample_bf16(const float *U, const int64_t *user_ids, int B, const float *It, int m, int d, int stride, float *S, int64_t ld,
                       hipStream_t stream);      // d <= 128
bool sample_top_supports(int d, int m);     // m sampled items
int launch_sample_top(const float *U, const int64_t *user_ids, int B, const float *It, int m, int d, int stride, const int *mask_rowptr,
                      const int *mask_items, unsigned *bits, float *S, int64_t ld, hipStream_t stream);      // d <= 128
int launch_sample_pack_top(const float *U, const int64_t *user_ids, int B, const void *ipack, int m, int d, int stride, const int *mask_rowptr,
                           const int *mask_items, unsigned *bits, float *S, int64_t ld, hipStream_t stream);  // the same, rows from the pack
int launch_user_bound(const float *U, const int64_t *user_ids, int B, int d, float *ubound /* [B][2] */, hipStream_t stream);
size_t item_pack_bytes(int I, int d);      // 0: no bf16 candidate pass for this width
int launch_item_pack(const float *It, int I, int d, void *pack, hipStream_t stream);
// The narrow bf16 filter's stage summary (large catalogues): one bit per LDS stage and (user, row half) -- some pass-bit word of the
// stage is non-zero -- as [B padded to 256][2][n_splits][sw] words; k_rescore then reads the flagged stages' words only.
struct PassSummary {
    unsigned *words = nullptr;      // NULL: none kept (the consumer scans every pass-bit word)
    int sw = 0;                     // summary words per (user, row half, split) = ceil(stages per split / 32)
    int n_splits = 0, items_per_split = 0, stage_items = 0;
};
int prefilter_stage_items(int d, bool long_stages);   // items per LDS stage of launch_prefilter's kernel for this width (long: the ring of two)
int launch_prefilter(const float *U, const int64_t *user_ids, int B, const void *ipack, int I, int d, const float *tau, int tau_stride,
                     const float *ubound, unsigned *mask, int Wh, int S, int items_per_split, bool wide, bool long_stages,
                     const PassSummary &summ, hipStream_t stream);
int pack_row_bytes(int d);                 // bytes of a packed item row; its 16-byte factor chunk is the last one
int launch_prefilter_wide(const float *U, const int64_t *user_ids, int B, const void *ipack, int I, int d, const float *tau, int tau_stride,
                          const float *ubound, void *logs, int *counts, int S, int items_per_split, int cap2, hipStream_t stream);
int launch_sample_wide(const float *U, const int64_t *user_ids, int B, const void *ipack, int I, int m, int d, int stride, float *S,
                       int64_t ld, hipStream_t stream);
int launch_rescore_list(const float *U, const int64_t *user_ids, int B, const float *It, int d, const float *tau, int tau_stride,
                        const int *surv, const int *surv_n, int surv_cap, void *lists, int *totals, int list_cap, hipStream_t stream);
int launch_rescore(const float *U, const int64_t *user_ids, int B, const float *It, int d, const float *tau, int tau_stride,
                   const unsigned *mask, int Wh, int n_units, const PassSummary &summ, void *lists, int *totals, int list_cap,
                   hipStream_t stream);
// the narrow rows' fp32 chains AND the exact selection of the top k in one launch (k_rescore<.., SELECT>): writes out_val / out_idx,
// flags the users it leaves to the exact fallback
int launch_rescore_select(const float *U, const int64_t *user_ids, int B, const float *It, int d, const float *tau, int tau_stride,
                          const unsigned *mask, int Wh, int n_units, const PassSummary &summ, int *totals, const int *mask_rowptr,
                          const int *mask_items, int k, int do_round, float *out_val, int64_t *out_idx, int *flagged, hipStream_t stream);
int launch_topk(const float *S, int64_t lds, int B, int I, int k, int do_round, float *out_val, int64_t *out_idx,
                hipStream_t stream);
int launch_mask(float *S, int64_t lds, int B, int I, const int *mask_rowptr, const int *mask_items, int item_div,
                hipStream_t stream);

}  // namespace tgcn
