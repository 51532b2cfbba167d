// transpose_bench.hip — the layout kernel (time-major [T][C] -> cell-major [C][T], float32) at bench.py's shape, as a function
// of its tile: a tile of TR time steps x TC cells is read in 256-byte row pieces and written in 256-byte pieces of TR * 4
// contiguous bytes per cell.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/ubench/build/transpose_bench scripts/ubench/transpose_bench.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)

__global__ void k_fill(float *p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (float)(i % 1000003);
}

__global__ void k_copy(const float4 *__restrict__ s, float4 *__restrict__ d, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}

template <int TR, int TC, int NT, bool NTS>
__global__ __launch_bounds__(NT) void k_tr(const float *__restrict__ src, int64_t T, int64_t C, float *__restrict__ dst)
{
    extern __shared__ float tile[];  // [TR][TC + 1]
    constexpr int LD = TC + 1, NW = NT / 64;
    const int64_t t0 = (int64_t)blockIdx.y * TR, c0 = (int64_t)blockIdx.x * TC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = wave; i < TR; i += NW) {
        const int64_t t = t0 + i;
#pragma unroll
        for (int h = 0; h < TC / 64; h++) {
            const int64_t c = c0 + h * 64 + lane;
            if (t < T && c < C) tile[i * LD + h * 64 + lane] = src[t * C + c];
        }
    }
    __syncthreads();
    for (int j = wave; j < TC; j += NW) {
        const int64_t c = c0 + j;
#pragma unroll
        for (int h = 0; h < TR / 64; h++) {
            const int64_t t = t0 + h * 64 + lane;
            if (t < T && c < C) {
                const float v = tile[(h * 64 + lane) * LD + j];
                if (NTS) __builtin_nontemporal_store(v, dst + c * T + t);
                else dst[c * T + t] = v;
            }
        }
    }
}

// the same tile with every load of a wave issued before the first LDS store (PER = rows per wave)
template <int TR, int TC, int NT>
__global__ __launch_bounds__(NT) void k_tr_regs(const float *__restrict__ src, int64_t T, int64_t C, float *__restrict__ dst)
{
    extern __shared__ float tile[];
    constexpr int LD = TC + 1, NW = NT / 64, PER = TR / NW, H = TC / 64, HT = TR / 64, PERC = TC / NW;
    const int64_t t0 = (int64_t)blockIdx.y * TR, c0 = (int64_t)blockIdx.x * TC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float v[PER][H];
#pragma unroll
    for (int i = 0; i < PER; i++)
#pragma unroll
        for (int h = 0; h < H; h++) {
            const int64_t t = t0 + wave + i * NW, c = c0 + h * 64 + lane;
            v[i][h] = (t < T && c < C) ? src[t * C + c] : 0.f;
        }
#pragma unroll
    for (int i = 0; i < PER; i++)
#pragma unroll
        for (int h = 0; h < H; h++) tile[(wave + i * NW) * LD + h * 64 + lane] = v[i][h];
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < PERC; jj++) {
        const int j = wave + jj * NW;
        const int64_t c = c0 + j;
#pragma unroll
        for (int h = 0; h < HT; h++) {
            const int64_t t = t0 + h * 64 + lane;
            if (t < T && c < C) dst[c * T + t] = tile[(h * 64 + lane) * LD + j];
        }
    }
}

template <int TR, int TC, int NT, bool NTS, bool REGS = false> static void run(const float *tm, float *cm, int T, int C, hipEvent_t e0, hipEvent_t e1)
{
    auto kern = REGS ? k_tr_regs<TR, TC, NT> : k_tr<TR, TC, NT, NTS>;
    const size_t lds = (size_t)TR * (TC + 1) * 4;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((C + TC - 1) / TC, (T + TR - 1) / TR);
    float best = 1e9f;
    for (int r = 0; r < 4; r++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern, grid, dim3(NT), lds, 0, tm, (int64_t)T, (int64_t)C, cm);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float m;
        CK(hipEventElapsedTime(&m, e0, e1));
        best = std::min(best, m);
    }
    std::vector<float> a((size_t)T * 3), b((size_t)3 * T);
    CK(hipMemcpy2D(a.data(), 3 * 4, tm + (C - 3), (size_t)C * 4, 3 * 4, T, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), cm + (size_t)(C - 3) * T, (size_t)3 * T * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (int c = 0; c < 3; c++) for (int t = 0; t < T; t++) bad += a[(size_t)t * 3 + c] != b[(size_t)c * T + t];
    printf("tile %3d time x %3d cells, %4d threads, %s stores: %.3f ms = %.2f TB/s (read + write)  mismatches %zu\n", TR, TC, NT, REGS ? "rg" : NTS ? "nt" : "  ", best,
           2.0 * T * C * 4 / best * 1e-9, bad);
    CK(hipMemset(cm, 0, (size_t)T * C * 4));
}

int main(int argc, char **argv)
{
    const int C = argc > 1 ? atoi(argv[1]) : 64800, T = argc > 2 ? atoi(argv[2]) : 10950;
    float *tm, *cm;
    CK(hipMalloc(&tm, (size_t)C * T * 4)); CK(hipMalloc(&cm, (size_t)C * T * 4));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, tm, (size_t)C * T);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 3; r++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, (const float4 *)tm, (float4 *)cm, (size_t)C * T / 4);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float m;
        CK(hipEventElapsedTime(&m, e0, e1));
        printf("plain copy: %.3f ms = %.2f TB/s\n", m, 2.0 * T * C * 4 / m * 1e-9);
    }
    printf("time-major [T][C] -> cell-major [C][T]\n");
    run<64, 64, 256, false>(tm, cm, T, C, e0, e1);
    run<256, 64, 1024, false, true>(tm, cm, T, C, e0, e1);
    run<256, 64, 1024, true>(tm, cm, T, C, e0, e1);
    run<512, 64, 1024, false, true>(tm, cm, T, C, e0, e1);
    run<384, 64, 1024, false>(tm, cm, T, C, e0, e1);
    run<256, 64, 512, false, true>(tm, cm, T, C, e0, e1);
    printf("cell-major [C][T] -> time-major [T][C]\n");
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, cm, (size_t)C * T);
    run<64, 64, 256, false>(cm, tm, C, T, e0, e1);
    run<64, 256, 1024, false, true>(cm, tm, C, T, e0, e1);
    run<256, 64, 1024, false, true>(cm, tm, C, T, e0, e1);
    run<128, 128, 1024, false, true>(cm, tm, C, T, e0, e1);
    run<64, 512, 1024, false, true>(cm, tm, C, T, e0, e1);
    run<128, 256, 1024, false, true>(cm, tm, C, T, e0, e1);
    return 0;
}
