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

// ---- exact selection of the k best of a user's candidates (k_select / k_select_flat, tgcn_score_fused.hip; the fused tail of
// k_rescore, tgcn_score_prefilter.hip) ----------------------------------------------------------------------------------------
constexpr int kSelCap = 1024;          // candidates per user held in LDS (8 KB per wave)
constexpr int kSelVPL = kSelCap / kWave;

__device__ __forceinline__ unsigned ordered_key(float v)
{
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// bitonic sort of one (value, item) pair per lane, best first: (value desc, item asc)
__device__ __forceinline__ void wave_sort_desc(float &v, int &i, int lane)
{
#pragma unroll
    for (int size = 2; size <= kWave; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const float ov = __shfl_xor(v, stride);
            const int oi = __shfl_xor(i, stride);
            const bool lower = (lane & stride) == 0;
            const bool first_half = (lane & size) == 0;       // this block sorts best-first; the other half worst-first
            const bool keep_better = lower == first_half;
            const bool other_better = better(ov, oi, v, i);
            if (other_better == keep_better) {
                v = ov;
                i = oi;
            }
        }
    }
}

// the k best of VPL (key, index) pairs per lane (key 0 = dropped / padding), sorted into lanes 0..k-1
template <int VPL>
__device__ __forceinline__ void select_core(const unsigned (&key)[VPL], const int (&idx)[VPL], int k, int lane, float &out_v,
                                            int &out_i, float2 *__restrict__ pack)
{
    // T = k-th largest key: the largest T with |{key >= T}| >= k
    unsigned T = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned c = T | (1u << bit);
        int cnt = 0;
#pragma unroll
        for (int s = 0; s < VPL; ++s)
            cnt += __popcll(__ballot(key[s] >= c));
        if (cnt >= k)
            T = c;
    }
    int above = 0;
#pragma unroll
    for (int s = 0; s < VPL; ++s)
        above += __popcll(__ballot(key[s] > T));
    // ties at T: the k - above smallest item ids among them.  I = the (k - above)-th smallest tied id (same search, on ids)
    const int need = k - above;
    unsigned I = 0xFFFFFFFFu;   // take every tie unless there are more than needed
    int ties = 0;
#pragma unroll
    for (int s = 0; s < VPL; ++s)
        ties += __popcll(__ballot(key[s] == T));
    if (ties > need) {
        // largest J with |{tied, ~id >= J}| >= need  <=>  smallest ids first (ids are non-negative: ~id order reverses them)
        unsigned J = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const unsigned c = J | (1u << bit);
            int cnt = 0;
#pragma unroll
            for (int s = 0; s < VPL; ++s)
                cnt += __popcll(__ballot(key[s] == T && ~(unsigned)idx[s] >= c));
            if (cnt >= need)
                J = c;
        }
        I = ~J;   // ids <= I are taken
    }
    // pack the k winners into LDS slots 0..k-1 (any order), then one per lane
    int base = 0;
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int s = 0; s < VPL; ++s) {
        const bool take = key[s] > T || (key[s] == T && (unsigned)idx[s] <= I);
        const unsigned long long m = __ballot(take);
        if (take)
            pack[base + __popcll(m & lt)] = make_float2(__uint_as_float(key[s]), __int_as_float(idx[s]));
        base += __popcll(m);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's LDS writes before its reads
    __builtin_amdgcn_wave_barrier();
    float v = -INFINITY;
    int i = INT_MAX;
    if (lane < k) {
        const float2 t = pack[lane];
        const unsigned kk = __float_as_uint(t.x);
        v = __uint_as_float((kk & 0x80000000u) ? (kk & 0x7FFFFFFFu) : ~kk);   // inverse of ordered_key
        i = __float_as_int(t.y);
    }
    wave_sort_desc(v, i, lane);
    out_v = v;
    out_i = i;
}

__device__ __forceinline__ float round4(float v) { return nearbyintf(v * 10000.0f) / 10000.0f; }  // ATen round(decimals=4)

}  // namespace tgcn
