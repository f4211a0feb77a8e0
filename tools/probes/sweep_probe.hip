// Micro-probe (development only, never linked into libtgcn.so): the inner loop of a "column sweep" SpMM -- persistent waves
// that OWN output rows (their running sums live in LDS slots) and walk the gathered table's column blocks in one global order,
// so that at any time the whole chip gathers from one block small enough for every XCD's L2; a row's entries are consumed in
// ascending column order by one wave (one fmaf chain per row: bit-exact, no piece sums, no reduce launch).
//
// Synthetic plan: W waves x R rows x NB blocks, every (row, block) segment SEGLEN entries with uniformly random columns inside the
// block.  A wave's stream is its segments in (block, row) order; bit i of the flag words marks the last entry of a segment; at
// a flagged entry the wave stores its sum to the row's LDS slot and continues with the next row's (prefetched) sum.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/sweep tools/probes/sweep_probe.hip && tools/probes/bin/sweep
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

__device__ __forceinline__ float readlane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// one wave: n_ent entries (a multiple of 64, zero-padded), slots[] = LDS slot of every segment in stream order
template <int UNROLL, bool SEGMENTS>
__global__ __launch_bounds__(256) void k_sweep(const float *__restrict__ X, const int *__restrict__ col, const float *__restrict__ val,
                                               const unsigned long long *__restrict__ flags, const unsigned char *__restrict__ slots,
                                               float *__restrict__ out, int n_ent, int n_seg, int R)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    const int wl = threadIdx.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wl);
    float *__restrict__ acc_lds = lds + (size_t)wl * R * D + lane;
    for (int r = 0; r < R; ++r)
        acc_lds[r * D] = 0.0f;
    const int *ec = col + (size_t)wave * n_ent;
    const float *ev = val + (size_t)wave * n_ent;
    const unsigned long long *ef = flags + (size_t)wave * (n_ent / 64);
    const unsigned char *sl = slots + (size_t)wave * n_seg;
    const char *Xb = reinterpret_cast<const char *>(X);
    int k = 0;                                  // segment cursor
    int slot_tab = sl[min(lane, n_seg - 1)];    // 64 slots at a time
    int cur = __builtin_amdgcn_readlane(slot_tab, 0);
    float acc = 0.0f;                           // (slot values start at zero)
    int nxt = __builtin_amdgcn_readlane(slot_tab, min(1, n_seg - 1));
    float acc_next = acc_lds[nxt * D];
    int c = ec[lane];
    float v = ev[lane];
    unsigned long long f = ef[0];
    for (int off = 0; off < n_ent; off += 64) {
        int c_n = 0;
        float v_n = 0.f;
        unsigned long long f_n = 0;
        if (off + 64 < n_ent) {
            c_n = ec[off + 64 + lane];
            v_n = ev[off + 64 + lane];
            f_n = ef[(off >> 6) + 1];
        }
#pragma unroll
        for (int j = 0; j < 64; j += UNROLL) {
            float x[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const unsigned cj = (unsigned)__builtin_amdgcn_readlane(c, j + u);
                x[u] = *reinterpret_cast<const float *>(Xb + ((size_t)cj << 8) + lane * 4);
            }
            const unsigned fm = (unsigned)(f >> j);
#pragma unroll
            for (int u4 = 0; u4 < UNROLL; u4 += 4) {
                const bool any = (fm >> u4) & 0xfu;
#pragma unroll
                for (int u = u4; u < u4 + 4; ++u) {
                    acc = fmaf(readlane_f(v, j + u), x[u], acc);
                    if (SEGMENTS && any && (fm & (1u << u))) {      // last entry of a segment: park the sum, take the next row's
                        acc_lds[cur * D] = acc;
                        acc = acc_next;
                        cur = nxt;
                        ++k;
                        const int k2 = min(k + 1, n_seg - 1);
                        if ((k2 & 63) == 0)
                            slot_tab = sl[min(k2 + lane, n_seg - 1)];
                        nxt = __builtin_amdgcn_readlane(slot_tab, k2 & 63);
                        acc_next = acc_lds[nxt * D];
                    }
                }
            }
        }
        c = c_n, v = v_n, f = f_n;
    }
    if (!SEGMENTS)
        acc_lds[0] = acc;
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int r = 0; r < R; ++r)
        out[((size_t)wave * R + r) * D + lane] = acc_lds[r * D];
}

int main(int argc, char **argv)
{
    struct Cfg { int mb, NB, R, L, W; };
    const Cfg cfgs[] = {{25, 8, 13, 12, 4096}, {25, 16, 13, 6, 4096}, {25, 16, 26, 6, 2048}, {25, 16, 7, 6, 8192},
                        {13, 4, 25, 12, 4096}, {13, 8, 25, 6, 4096},  {13, 8, 13, 6, 8192}, {25, 32, 13, 3, 4096}};
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::mt19937_64 rng(5);
    for (const Cfg &g : cfgs) {
        const long long rows = (long long)(g.mb << 20) / 256, per = rows / g.NB;
        const int n_seg = g.NB * g.R;
        const int raw = n_seg * g.L, n_ent = (raw + 63) / 64 * 64;
        const long long entries = (long long)g.W * n_ent;
        std::vector<int> hc(entries, 0);
        std::vector<float> hv(entries, 0.f);
        std::vector<unsigned long long> hf(entries / 64, 0ull);
        std::vector<unsigned char> hs((size_t)g.W * n_seg);
        for (int w = 0; w < g.W; ++w) {
            int e = 0;
            for (int b = 0; b < g.NB; ++b)
                for (int r = 0; r < g.R; ++r) {
                    hs[(size_t)w * n_seg + b * g.R + r] = (unsigned char)r;
                    for (int i = 0; i < g.L; ++i, ++e) {
                        hc[(size_t)w * n_ent + e] = (int)(b * per + rng() % per);
                        hv[(size_t)w * n_ent + e] = 1.0f + (rng() % 100) * 1e-3f;
                    }
                    const size_t last = (size_t)w * n_ent + e - 1;
                    hf[last / 64] |= 1ull << (last % 64);
                }
        }
        float *dX, *dval, *dout;
        int *dcol;
        unsigned long long *dfl;
        unsigned char *dsl;
        CK(hipMalloc(&dX, (size_t)rows * 256));
        CK(hipMemset(dX, 0, (size_t)rows * 256));
        CK(hipMalloc(&dval, entries * 4));
        CK(hipMalloc(&dcol, entries * 4));
        CK(hipMalloc(&dfl, entries / 8));
        CK(hipMalloc(&dsl, hs.size()));
        CK(hipMalloc(&dout, (size_t)g.W * g.R * D * 4));
        CK(hipMemcpy(dval, hv.data(), entries * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dcol, hc.data(), entries * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dfl, hf.data(), entries / 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dsl, hs.data(), hs.size(), hipMemcpyHostToDevice));
        const size_t lds = (size_t)4 * g.R * D * 4;
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
            printf("{\"table_MB\": %d, \"blocks\": %d, \"rows_per_wave\": %d, \"seg_len\": %d, \"waves\": %d, \"entries\": %lld, \"variant\": \"%s\", "
                   "\"us\": %.1f, \"gather_TBs\": %.2f}\n", g.mb, g.NB, g.R, g.L, g.W, (long long)g.W * raw, name, us, (double)g.W * raw * 256.0 / us / 1e6);
            fflush(stdout);
        };
        run("segments_u16", [&] { hipLaunchKernelGGL((k_sweep<16, true>), dim3(g.W / 4), dim3(256), lds, 0, dX, dcol, dval, dfl, dsl, dout, n_ent, n_seg, g.R); });
        run("segments_u32", [&] { hipLaunchKernelGGL((k_sweep<32, true>), dim3(g.W / 4), dim3(256), lds, 0, dX, dcol, dval, dfl, dsl, dout, n_ent, n_seg, g.R); });
        run("no_switch_u16", [&] { hipLaunchKernelGGL((k_sweep<16, false>), dim3(g.W / 4), dim3(256), lds, 0, dX, dcol, dval, dfl, dsl, dout, n_ent, n_seg, g.R); });
        CK(hipFree(dX));
        CK(hipFree(dval));
        CK(hipFree(dcol));
        CK(hipFree(dfl));
        CK(hipFree(dsl));
        CK(hipFree(dout));
    }
    return 0;
}
