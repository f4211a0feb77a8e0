// Wave-cooperative top-k list (k <= 64) kept in registers: lane j holds the j-th best entry,
// ordered by (value descending, index ascending).  Shared by the scoring kernels.
#pragma once
#include <climits>

#include "tgcn_internal.h"

namespace tgcn {

struct TopList {
    float v;
    int i;
};

__device__ __forceinline__ bool better(float av, int ai, float bv, int bi)
{
    return av > bv || (av == bv && ai < bi);
}

// lane i <- lane i-1 across the whole wave as ONE VALU instruction (DPP wave_shr:1, GFX9 family incl. gfx950);
// __shfl_up would go through the LDS crossbar (ds_bpermute, ~100 cycles) -- the insert below is a serial chain, so
// that latency was most of the selection kernels' time.  Lane 0 keeps its own value (it is never read: pos >= 0).
__device__ __forceinline__ int wave_shr1(int v)
{
    return __builtin_amdgcn_update_dpp(v, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

// insert (cv, ci) (wave-uniform) into the sorted 64-entry list; entries past the end fall off
__device__ __forceinline__ void list_insert(TopList &e, float cv, int ci, int lane)
{
    const bool beats = better(e.v, e.i, cv, ci);
    const int pos = __popcll(__ballot(beats));  // sorted list: `beats` is a prefix of lanes
    const float uv = __int_as_float(wave_shr1(__float_as_int(e.v)));
    const int ui = wave_shr1(e.i);
    if (lane == pos) {
        e.v = cv;
        e.i = ci;
    } else if (lane > pos) {
        e.v = uv;
        e.i = ui;
    }
}

// offer one value per lane (sv at index si, `on` = lane holds a real element); k-th entry is the bar
__device__ __forceinline__ void list_offer(TopList &e, float sv, int si, bool on, int k, int lane)
{
    float tv = readlane_f(e.v, k - 1);
    int ti = __builtin_amdgcn_readlane(e.i, k - 1);
    unsigned long long m = __ballot(on && better(sv, si, tv, ti));
    while (m) {
        const int f = __ffsll((long long)m) - 1;
        const float cv = readlane_f(sv, f);
        const int ci = __builtin_amdgcn_readlane(si, f);
        list_insert(e, cv, ci, lane);
        tv = readlane_f(e.v, k - 1);
        ti = __builtin_amdgcn_readlane(e.i, k - 1);
        m &= ~(1ull << f);
        m &= __ballot(on && better(sv, si, tv, ti));
    }
}


// ascending-sorted int array membership (train-item mask lookups)
__device__ __forceinline__ bool sorted_contains(const int *__restrict__ a, int beg, int end, int x)
{
    int lo = beg, hi = end;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const int v = a[mid];
        if (v < x)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo < end && a[lo] == x;
}

__device__ __forceinline__ float round4(float v) { return nearbyintf(v * 10000.0f) / 10000.0f; }  // ATen round(decimals=4)

}  // namespace tgcn
