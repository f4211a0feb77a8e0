// Micro-probe (development only, never linked into libtgcn.so): does the WIDTH of the per-lane load change the rate at which a
// wave gathers random 256-byte rows (d = 64 fp32) from the level the table lives in?
//
// Every wave owns a tile of `tile` (col, val) entries, as a tile wave of k_spmm_seg does, and accumulates val * X[col, :]:
//   W1: one row per wave instruction  -- lane l loads 4 bytes of the row (global_load_dword; what k_spmm_wave / k_spmm_seg do)
//   W2: two rows per instruction      -- lanes 0-31 / 32-63 take 8 bytes each of rows e, e + 1 (global_load_dwordx2)
//   W4: four rows per instruction     -- lane group q = lane / 16 takes 16 bytes of row e + q (global_load_dwordx4)
// The partial sums of W2 / W4 are combined across lane groups at the end of the tile (a different summation order: fine for
// the non-exact mode only).  Column ids are drawn uniformly from the slice of the table the workgroup's XCD class owns
// (blockIdx % 8, the segmented kernel's placement) or from the whole table (--affine 0).
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gwp tools/probes/gather_width_probe.hip && /tmp/gwp
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
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

template <int UNROLL>
__global__ __launch_bounds__(256) void k_w1(const float *__restrict__ X, const int *__restrict__ col, const float *__restrict__ val,
                                            float *__restrict__ out, int tile)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int *ec = col + (size_t)wave * tile;
    const float *ev = val + (size_t)wave * tile;
    const char *Xb = reinterpret_cast<const char *>(X);
    float acc = 0.f;
    for (int off = 0; off < tile; off += 64) {
        const int c = ec[off + lane];
        const float v = ev[off + lane];
#pragma unroll
        for (int j = 0; j < 64; j += UNROLL) {
            float x[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const unsigned cj = (unsigned)__builtin_amdgcn_readlane(c, j + u);
                x[u] = *reinterpret_cast<const float *>(Xb + ((size_t)cj << 8) + lane * 4);
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                acc = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j + u)), x[u], acc);
        }
    }
    out[(size_t)wave * D + lane] = acc;
}

// four rows per instruction: entry e = off + 4 * t + q is taken by lane group q at step t
template <int UNROLL>
__global__ __launch_bounds__(256) void k_w4(const float *__restrict__ X, const int *__restrict__ col, const float *__restrict__ val,
                                            float *__restrict__ out, int tile)
{
    const int lane = threadIdx.x & 63;
    const int q = lane >> 4, l = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int *ec = col + (size_t)wave * tile;
    const float *ev = val + (size_t)wave * tile;
    const char *Xb = reinterpret_cast<const char *>(X);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int off = 0; off < tile; off += 64) {
        const int c = ec[off + lane];
        const float v = ev[off + lane];
#pragma unroll
        for (int t = 0; t < 16; t += UNROLL) {
            float4 x[UNROLL];
            float vv[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int src = 4 * (t + u) + q;
                const unsigned cj = (unsigned)__shfl(c, src, 64);
                vv[u] = __shfl(v, src, 64);
                x[u] = *reinterpret_cast<const float4 *>(Xb + ((size_t)cj << 8) + l * 16);
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                acc.x = fmaf(vv[u], x[u].x, acc.x);
                acc.y = fmaf(vv[u], x[u].y, acc.y);
                acc.z = fmaf(vv[u], x[u].z, acc.z);
                acc.w = fmaf(vv[u], x[u].w, acc.w);
            }
        }
    }
    // combine the four lane groups (rows of 16 lanes): lanes 0-15 end with the sums
    float r[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        r[k] += __shfl_down(r[k], 32, 64);
        r[k] += __shfl_down(r[k], 16, 64);
    }
    if (q == 0)
        *reinterpret_cast<float4 *>(out + (size_t)wave * D + l * 4) = make_float4(r[0], r[1], r[2], r[3]);
}

// two rows per instruction
template <int UNROLL>
__global__ __launch_bounds__(256) void k_w2(const float *__restrict__ X, const int *__restrict__ col, const float *__restrict__ val,
                                            float *__restrict__ out, int tile)
{
    const int lane = threadIdx.x & 63;
    const int q = lane >> 5, l = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int *ec = col + (size_t)wave * tile;
    const float *ev = val + (size_t)wave * tile;
    const char *Xb = reinterpret_cast<const char *>(X);
    float2 acc = make_float2(0.f, 0.f);
    for (int off = 0; off < tile; off += 64) {
        const int c = ec[off + lane];
        const float v = ev[off + lane];
#pragma unroll
        for (int t = 0; t < 32; t += UNROLL) {
            float2 x[UNROLL];
            float vv[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int src = 2 * (t + u) + q;
                const unsigned cj = (unsigned)__shfl(c, src, 64);
                vv[u] = __shfl(v, src, 64);
                x[u] = *reinterpret_cast<const float2 *>(Xb + ((size_t)cj << 8) + l * 8);
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                acc.x = fmaf(vv[u], x[u].x, acc.x);
                acc.y = fmaf(vv[u], x[u].y, acc.y);
            }
        }
    }
    acc.x += __shfl_down(acc.x, 32, 64);
    acc.y += __shfl_down(acc.y, 32, 64);
    if (q == 0)
        *reinterpret_cast<float2 *>(out + (size_t)wave * D + l * 2) = acc;
}

int main(int argc, char **argv)
{
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");      // w1 only (the width question is settled: profiles/r03_experiments.md)
    const int tile = argc > 2 ? atoi(argv[2]) : 1024;
    const long long entries = 1ll << 23;     // 8 M gathers per launch = 2 GB of rows
    const int n_waves = (int)(entries / tile);
    const int grid = n_waves / 4;
    std::mt19937_64 rng(3);
    std::vector<float> hv(entries);
    for (auto &f : hv)
        f = 1.0f + (rng() % 1000) * 1e-4f;
    float *dval, *dout;
    int *dcol;
    CK(hipMalloc(&dval, entries * 4));
    CK(hipMalloc(&dcol, entries * 4));
    CK(hipMalloc(&dout, (size_t)n_waves * D * 4));
    CK(hipMemcpy(dval, hv.data(), entries * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const size_t sizes_mb[] = {2, 13, 25, 38, 256, 2048};      // 25 MB affine = the segmented kernel's 8 x 3.2 MB; 38 MB: config 2's whole table
    for (size_t mb : sizes_mb) {
        const long long rows = (long long)(mb << 20) / 256;
        float *dX;
        CK(hipMalloc(&dX, (size_t)rows * 256));
        CK(hipMemset(dX, 0, (size_t)rows * 256));
        for (int affine = 0; affine < 4; ++affine) {
            // 0: uniform over the table; 1: SPATIAL blocks (workgroup id % 8 picks the eighth: the segmented kernel's placement);
            // 2 / 3: TEMPORAL blocks -- every wave walks the table's 8 (2) or 16 (3) column blocks in the same order, tile / NB entries
            // from each, so at any time the whole chip gathers from one block (replicated in the eight L2s); no placement involved
            if (affine >= 2 && mb < 25)
                continue;
            std::vector<int> hc(entries);
            const int nb = affine == 3 ? 16 : 8;
            const long long per = rows / nb;
            for (long long w = 0; w < n_waves; ++w) {
                const long long blk = (w / 4) % 8;     // workgroup id % 8: the XCD class under round-robin placement
                for (int e = 0; e < tile; ++e) {
                    if (affine >= 2)
                        hc[w * tile + e] = (int)((long long)(e / (tile / nb)) * per + rng() % per);
                    else
                        hc[w * tile + e] = affine ? (int)(blk * per + rng() % per) : (int)(rng() % rows);
                }
            }
            CK(hipMemcpy(dcol, hc.data(), entries * 4, hipMemcpyHostToDevice));
            auto run = [&](const char *name, auto launch) {
                for (int i = 0; i < 2; ++i)
                    launch();
                CK(hipEventRecord(e0));
                const int reps = 10;
                for (int i = 0; i < reps; ++i)
                    launch();
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                const double us = ms * 1e3 / reps;
                printf("{\"table_MB\": %zu, \"affine\": %d, \"variant\": \"%s\", \"us\": %.1f, \"gather_TBs\": %.2f}\n", mb, affine, name, us,
                       entries * 256.0 / us / 1e6);
                fflush(stdout);
            };
            run("w1_u16", [&] { hipLaunchKernelGGL((k_w1<16>), dim3(grid), dim3(256), 0, 0, dX, dcol, dval, dout, tile); });
            run("w1_u32", [&] { hipLaunchKernelGGL((k_w1<32>), dim3(grid), dim3(256), 0, 0, dX, dcol, dval, dout, tile); });
            if (!quick) {
                run("w2_u16", [&] { hipLaunchKernelGGL((k_w2<16>), dim3(grid), dim3(256), 0, 0, dX, dcol, dval, dout, tile); });
                run("w4_u8", [&] { hipLaunchKernelGGL((k_w4<8>), dim3(grid), dim3(256), 0, 0, dX, dcol, dval, dout, tile); });
                run("w4_u16", [&] { hipLaunchKernelGGL((k_w4<16>), dim3(grid), dim3(256), 0, 0, dX, dcol, dval, dout, tile); });
            }
        }
        CK(hipFree(dX));
    }
    return 0;
}
