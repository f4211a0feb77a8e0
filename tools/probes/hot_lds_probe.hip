// Micro-probe (development only, never linked into libtgcn.so): would keeping the HOTTEST rows of the gathered table in LDS speed up
// the direct (user) rows of config 2's layer?  Their columns are items drawn Zipf-like (p_i ~ (i + 1)^-0.8 over 50 000 items, ids
// permuted): the 640 most popular rows (160 KB) take ~34 % of the gathers, the 320 most popular ~28 %.
//
//   wave_per_row : the production shape -- 256-thread workgroups, one wave per row of 50 entries, 16 gathers in flight
//   persistent   : G workgroups of T threads, H hot rows copied into LDS first, waves stride over the rows; an entry whose (remapped)
//                  column is negative reads LDS slot -(c + 1), the others gather from the table as before
// Same fmaf chain per row in every variant (outputs compared).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/hot tools/probes/hot_lds_probe.hip && tools/probes/bin/hot
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <random>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

constexpr int D = 64;

__device__ __forceinline__ float readlane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// one gather, or one LDS read when the (wave-uniform) column is a hot slot: the branch and both forms are assembly, because hipcc's
// wait-count pass answers a branch around a load with s_waitcnt vmcnt(0) lgkmcnt(0) behind every one of them (measured in the
// ISA of the plain C++ form: 16 full waits per batch of 16)
__device__ __forceinline__ void gather_or_lds(float &x, const float *row, int lane4, int cj, int slot_bytes)
{
    asm volatile("s_cmp_lt_i32 %3, 0\n\t"
                 "s_cbranch_scc1 1f\n\t"
                 "global_load_dword %0, %2, %1\n\t"
                 "s_branch 2f\n"
                 "1:\n\t"
                 "v_add_u32 %0, %4, %2\n\t"
                 "ds_read_b32 %0, %0\n"
                 "2:"
                 : "=&v"(x)
                 : "s"(row), "v"(lane4), "s"(cj), "s"(slot_bytes)
                 : "scc", "memory");
}

template <bool HOT>
__device__ __forceinline__ float row_chain(const float *__restrict__ X, const float *hot, const int *__restrict__ col,
                                           const float *__restrict__ val, int beg, int end, int lane)
{
    const float *__restrict__ Xl = X + lane;
    float acc = 0.0f;
    for (int base = beg; base < end; base += 64) {
        const int n = min(64, end - base);
        int c = 0;
        float v = 0.0f;
        if (lane < n) {
            c = col[base + lane];
            v = val[base + lane];
        }
        for (int j = 0; j < n; j += 16) {
            float x[16];
            if constexpr (HOT) {
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int cj = __builtin_amdgcn_readlane(c, min(j + u, n - 1));
                    gather_or_lds(x[u], X + (size_t)max(cj, 0) * D, lane * 4, cj, (-cj - 1) * (D * 4));
                }
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                             : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]),
                               "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]));
            } else {
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int cj = __builtin_amdgcn_readlane(c, min(j + u, n - 1));
                    x[u] = Xl[(size_t)cj * D];
                }
            }
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (j + u < n)
                    acc = fmaf(readlane_f(v, j + u), x[u], acc);
        }
    }
    return acc;
}

__global__ __launch_bounds__(256) void k_wave_per_row(const float *__restrict__ X, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                      const float *__restrict__ val, float *__restrict__ out, int n_rows)
{
    const int lane = threadIdx.x & 63;
    const int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (row >= n_rows)
        return;
    out[(size_t)row * D + lane] = row_chain<false>(X, nullptr, col, val, rowptr[row], rowptr[row + 1], lane);
}

template <bool HOT, int T>
__global__ __launch_bounds__(T) void k_persistent(const float *__restrict__ X, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                  const float *__restrict__ val, const int *__restrict__ hot_rows, int H,
                                                  float *__restrict__ out, int n_rows)
{
    extern __shared__ float hot[];
    const int lane = threadIdx.x & 63;
    if (HOT) {
        for (int i = threadIdx.x; i < H * (D / 4); i += T) {       // 16 bytes per thread and step
            const int r = i / (D / 4), q = i % (D / 4);
            reinterpret_cast<float4 *>(hot)[i] = *reinterpret_cast<const float4 *>(X + (size_t)hot_rows[r] * D + 4 * q);
        }
        __syncthreads();
    }
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (T / 64) + (threadIdx.x >> 6));
    const int n_waves = gridDim.x * (T / 64);
    for (int row = wave; row < n_rows; row += n_waves)
        out[(size_t)row * D + lane] = row_chain<HOT>(X, hot, col, val, rowptr[row], rowptr[row + 1], lane);
}

int main()
{
    const int I = 50000, R = 100000, DEG = 50;
    std::mt19937_64 rng(7);
    // Zipf-like popularity over permuted ids
    std::vector<double> cdf(I);
    double s = 0;
    for (int i = 0; i < I; ++i)
        cdf[i] = (s += std::pow(i + 1.0, -0.8));
    std::vector<int> perm(I);
    std::iota(perm.begin(), perm.end(), 0);
    std::shuffle(perm.begin(), perm.end(), rng);
    std::vector<int> rowptr(R + 1, 0), col((size_t)R * DEG);
    std::vector<float> val((size_t)R * DEG);
    std::uniform_real_distribution<double> uni(0.0, s);
    std::vector<long long> freq(I, 0);
    for (int r = 0; r < R; ++r) {
        int *c = &col[(size_t)r * DEG];
        for (int e = 0; e < DEG; ++e) {
            const int rank = (int)(std::lower_bound(cdf.begin(), cdf.end(), uni(rng)) - cdf.begin());
            c[e] = perm[std::min(rank, I - 1)];
        }
        std::sort(c, c + DEG);
        for (int e = 0; e < DEG; ++e) {
            ++freq[c[e]];
            val[(size_t)r * DEG + e] = 0.01f + (rng() % 100) * 1e-4f;
        }
        rowptr[r + 1] = (r + 1) * DEG;
    }
    std::vector<int> by_freq(I);
    std::iota(by_freq.begin(), by_freq.end(), 0);
    std::sort(by_freq.begin(), by_freq.end(), [&](int a, int b) { return freq[a] > freq[b]; });
    std::vector<float> hX((size_t)I * D);
    for (auto &x : hX)
        x = (float)((rng() % 2001) - 1000) * 1e-3f;

    float *dX, *dval, *dout, *dref;
    int *drp, *dcol, *dcol_hot, *dhot;
    const size_t E = col.size();
    CK(hipMalloc(&dX, hX.size() * 4));
    CK(hipMalloc(&dval, E * 4));
    CK(hipMalloc(&dcol, E * 4));
    CK(hipMalloc(&dcol_hot, E * 4));
    CK(hipMalloc(&drp, (R + 1) * 4));
    CK(hipMalloc(&dhot, I * 4));
    CK(hipMalloc(&dout, (size_t)R * D * 4));
    CK(hipMalloc(&dref, (size_t)R * D * 4));
    CK(hipMemcpy(dX, hX.data(), hX.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dval, val.data(), E * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dcol, col.data(), E * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(drp, rowptr.data(), (R + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dhot, by_freq.data(), I * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<float> ref((size_t)R * D), got((size_t)R * D);
    auto run = [&](const char *name, int H, int T, int per_cu, double share, auto launch, bool is_ref) {
        for (int i = 0; i < 3; ++i)
            launch();
        CK(hipEventRecord(e0));
        const int reps = 20;
        for (int i = 0; i < reps; ++i)
            launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        bool same = true;
        if (is_ref) {
            CK(hipMemcpy(ref.data(), dout, ref.size() * 4, hipMemcpyDeviceToHost));
        } else {
            CK(hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost));
            same = std::memcmp(ref.data(), got.data(), ref.size() * 4) == 0;
        }
        printf("{\"variant\": \"%s\", \"hot_rows\": %d, \"threads\": %d, \"workgroups_per_cu\": %d, \"hot_share\": %.3f, \"us\": %.1f, "
               "\"gather_TBs\": %.2f, \"identical\": %s}\n", name, H, T, per_cu, share, us, (double)E * 256.0 / us / 1e6, same ? "true" : "false");
        fflush(stdout);
        CK(hipMemset(dout, 0, (size_t)R * D * 4));
    };
    run("wave_per_row", 0, 256, 0, 0.0, [&] { hipLaunchKernelGGL(k_wave_per_row, dim3((R + 3) / 4), dim3(256), 0, 0, dX, drp, dcol, dval, dout, R); }, true);
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_persistent<true, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_persistent<true, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int per_cu : {1, 2, 4})
        run("persistent_no_hot", 0, 1024, per_cu, 0.0,
            [&] { hipLaunchKernelGGL((k_persistent<false, 1024>), dim3(256 * per_cu), dim3(1024), 0, 0, dX, drp, dcol, dval, dhot, 0, dout, R); }, false);
    struct V { int H, T, per_cu; };
    for (const V v : {V{640, 1024, 1}, V{320, 1024, 2}, V{160, 1024, 2}, V{320, 512, 2}, V{160, 512, 4}, V{128, 1024, 2}, V{64, 1024, 2}}) {
        std::vector<int> slot(I, -1);
        long long hot_entries = 0;
        for (int k = 0; k < v.H; ++k)
            slot[by_freq[k]] = k, hot_entries += freq[by_freq[k]];
        std::vector<int> ch(E);
        for (size_t e = 0; e < E; ++e)
            ch[e] = slot[col[e]] >= 0 ? -(slot[col[e]] + 1) : col[e];
        CK(hipMemcpy(dcol_hot, ch.data(), E * 4, hipMemcpyHostToDevice));
        const size_t lds = (size_t)v.H * D * 4;
        const double share = (double)hot_entries / (double)E;
        if (v.T == 1024)
            run("persistent_hot_lds", v.H, v.T, v.per_cu, share,
                [&] { hipLaunchKernelGGL((k_persistent<true, 1024>), dim3(256 * v.per_cu), dim3(1024), lds, 0, dX, drp, dcol_hot, dval, dhot, v.H, dout, R); }, false);
        else
            run("persistent_hot_lds", v.H, v.T, v.per_cu, share,
                [&] { hipLaunchKernelGGL((k_persistent<true, 512>), dim3(256 * v.per_cu), dim3(512), lds, 0, dX, drp, dcol_hot, dval, dhot, v.H, dout, R); }, false);
    }
    return 0;
}
