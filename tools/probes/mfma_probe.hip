// Micro-probe (development only, never linked into libtgcn.so): what keeps a SIMD's fp32 matrix pipe from 100 % when two
// or three in-order waves share it?  Each variant runs UNITS units of 64 v_mfma_f32_32x32x2_f32 per wave (the filter
// kernel's unit), with the ingredients of the real loop added one at a time:
//   bit 0: a workgroup barrier after every unit
//   bit 1: 32 ds_read_b64 per unit feeding the A operand (otherwise registers)
//   bit 2: one compare + exec-masked (never taken) branch per MFMA pair on the other accumulator set
//   bit 3: 4 global loads per unit + 8 ds_write_b64 before the barrier (the staging of the next item stage)
// Prints shader cycles per unit per wave (s_memtime around the loop, wave 0 of each workgroup).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int MODE>
__global__ __launch_bounds__(256) void k_probe(const float *__restrict__ src, float *__restrict__ out, unsigned long long *__restrict__ cyc,
                                               int units, float tau)
{
    __shared__ __attribute__((aligned(16))) float smem[2 * 64 * 66];
    const int lane = threadIdx.x & 63;
    const int r32 = lane & 31, h = lane >> 5;
    for (int i = threadIdx.x; i < 2 * 64 * 66; i += 256)
        smem[i] = src[i % 4096] * 1e-3f;
    __syncthreads();
    f32x16 a0, a1, p0, p1;
    for (int r = 0; r < 16; ++r)
        a0[r] = a1[r] = p0[r] = p1[r] = 0.0f;
    float bq[16];
    for (int q = 0; q < 16; ++q)
        bq[q] = src[(lane + q * 64) % 4096];
    float areg = src[lane];
    int cnt = 0;
    int buf = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int u = 0; u < units; ++u) {
        float4 nxt[4];
        if (MODE & 8) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                nxt[i] = *reinterpret_cast<const float4 *>(src + ((u * 1024 + i * 256 + threadIdx.x) * 4) % 4096);
        }
        const float *pi = smem + buf * 64 * 66 + r32 * 66 + 2 * h;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            float2 x0 = make_float2(areg, areg), x1 = make_float2(areg, areg);
            if (MODE & 2) {
                x0 = *reinterpret_cast<const float2 *>(pi + q * 4);
                x1 = *reinterpret_cast<const float2 *>(pi + 32 * 66 + q * 4);
            }
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.x, bq[q], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1.x, bq[q], a1, 0, 0, 0);
            if (MODE & 4) {
                if (p0[q] > tau) {
                    out[(cnt & 1023) * 64 + lane] = p0[q];
                    ++cnt;
                }
            }
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0.y, bq[q], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1.y, bq[q], a1, 0, 0, 0);
            if (MODE & 4) {
                if (p1[q] > tau) {
                    out[(cnt & 1023) * 64 + lane] = p1[q];
                    ++cnt;
                }
            }
        }
        if (MODE & 4) {   // the accumulators just finished become the tested set of the next unit
            p0 = a0, p1 = a1;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                a0[r] = a1[r] = 0.0f;
        }
        if (MODE & 8) {
            float *dst = smem + (buf ^ 1) * 64 * 66;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int f = i * 256 + threadIdx.x;
                float *o = dst + (f >> 4) * 66 + (f & 15) * 4;
                *reinterpret_cast<float2 *>(o) = make_float2(nxt[i].x, nxt[i].z);
                *reinterpret_cast<float2 *>(o + 2) = make_float2(nxt[i].y, nxt[i].w);
            }
        }
        if (MODE & 1)
            __syncthreads();
        if (MODE & 8)
            buf ^= 1;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int r = 0; r < 16; ++r)
        s += a0[r] + a1[r] + p0[r] + p1[r];
    out[blockIdx.x * 256 + threadIdx.x] = s + cnt;
    if (threadIdx.x == 0)
        cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const float *src, float *out, unsigned long long *cyc, int wgs_per_cu, int units)
{
    const int grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL(k_probe<MODE>, dim3(grid), dim3(256), 0, 0, src, out, cyc, units, 1e30f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_probe<MODE>, dim3(grid), dim3(256), 0, 0, src, out, cyc, units, 1e30f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double ideal = 64.0 * 64.0 * wgs_per_cu;   // cycles per unit if the pipe never idles
    printf("{\"mode\": %d, \"workgroups_per_cu\": %d, \"cycles_per_unit_median\": %.0f, \"cycles_per_unit_max\": %.0f, "
           "\"ideal_cycles_per_unit\": %.0f, \"pipe_busy_median\": %.3f, \"kernel_us\": %.1f, \"tflops\": %.1f}\n",
           MODE, wgs_per_cu, (double)h[grid / 2] / units, (double)h[grid - 1] / units, ideal, ideal / ((double)h[grid / 2] / units),
           ms * 1e3, 2.0 * 32 * 32 * 2 * 64.0 * units * grid * 4 / (ms * 1e-3) / 1e12);
}

int main(int argc, char **argv)
{
    const int units = 100;
    float *src, *out;
    unsigned long long *cyc;
    hipMalloc(&src, 4096 * sizeof(float) * 4);
    hipMalloc(&out, 1 << 24);
    hipMalloc(&cyc, 4096 * sizeof(unsigned long long));
    std::vector<float> h(4096 * 4);
    for (size_t i = 0; i < h.size(); ++i)
        h[i] = (float)((i * 2654435761u) % 1000) / 1000.0f - 0.5f;
    hipMemcpy(src, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    for (int w = 1; w <= 3; ++w) {
        run<0>(src, out, cyc, w, units);
        run<1>(src, out, cyc, w, units);
        run<2>(src, out, cyc, w, units);
        run<3>(src, out, cyc, w, units);
        run<7>(src, out, cyc, w, units);
        run<11>(src, out, cyc, w, units);
        run<15>(src, out, cyc, w, units);
    }
    return 0;
}
