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


// launchers shared between translation units (arguments already validated by the caller)
int launch_score_dense(const float *U, const int64_t *user_ids, int B, const float *It, int I, int d, int item_mul, float *S,
                       int64_t lds, hipStream_t stream);
int launch_score_dense_filter(const float *U, const int64_t *user_ids, int B, const float *It, int I, int d, const float *tau,
                              int tau_stride, void *logs, int *counts, int S, int items_per_split, int cap2, hipStream_t stream);
int launch_topk(const float *S, int64_t lds, int B, int I, int k, int do_round, float *out_val, int64_t *out_idx,
                hipStream_t stream);
int launch_mask(float *S, int64_t lds, int B, int I, const int *mask_rowptr, const int *mask_items, int item_div,
                hipStream_t stream);

}  // namespace tgcn
