// overlap_probe.hip — can a layout change (time-major -> cell-major) run UNDER k_train_cell3?  The training kernel owns the LDS of
// every CU (159 KB of 160) but only 4 of 8 wave slots per SIMD and ~450 of 512 VGPRs: a transpose that needs no LDS and at most
// 64 VGPRs can be resident beside it.  This probe times (a) the training kernel alone, (b) the register transpose alone, (c) both
// on two streams.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Ixsdba_amd/csrc -o scripts/ubench/build/overlap_probe scripts/ubench/overlap_probe.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <iterator>
#include <vector>

#include "train_cell3.hpp"
#include "kernels.hpp"

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)

// time-major src[t][c] -> cell-major dst[c][t], no LDS: a wave takes 64 consecutive cells x TT consecutive time steps; lane = cell
template <int TT>
__global__ __launch_bounds__(256) void k_transpose_regs(const float *__restrict__ src, int64_t T, int64_t C, float *__restrict__ dst)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwave = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t ntc = (C + 63) / 64, ntt = (T + TT - 1) / TT;
    for (int64_t tile = wave; tile < ntc * ntt; tile += nwave) {
        const int64_t c0 = (tile % ntc) * 64, t0 = (tile / ntc) * TT;
        const int64_t c = c0 + lane;
        float v[TT];
#pragma unroll
        for (int i = 0; i < TT; i++) v[i] = (c < C && t0 + i < T) ? src[(t0 + i) * C + c] : 0.f;
        if (c < C) {
            float *d = dst + c * T + t0;
            if (t0 + TT <= T && ((reinterpret_cast<uintptr_t>(d) & 15) == 0)) {
#pragma unroll
                for (int i = 0; i < TT; i += 4) *reinterpret_cast<float4 *>(d + i) = make_float4(v[i], v[i + 1], v[i + 2], v[i + 3]);
            } else {
#pragma unroll
                for (int i = 0; i < TT; i++) if (t0 + i < T) d[i] = v[i];
            }
        }
    }
}

template <typename T> static T *upload(const std::vector<T> &v)
{
    T *d;
    CK(hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(T)));
    CK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

int main(int argc, char **argv)
{
    const int C = argc > 1 ? atoi(argv[1]) : 64800;
    const int T = 10950, G = 365, W = 31, Q = 20;
    float *ref, *hist, *af, *hq, *tm, *cm;
    int *flags;
    CK(hipMalloc(&ref, (size_t)C * T * 4)); CK(hipMalloc(&hist, (size_t)C * T * 4));
    CK(hipMalloc(&tm, (size_t)C * T * 4)); CK(hipMalloc(&cm, (size_t)C * T * 4));
    CK(hipMalloc(&af, (size_t)C * G * Q * 4)); CK(hipMalloc(&hq, (size_t)C * G * Q * 4));
    CK(hipMalloc(&flags, ((size_t)C + 1) * 4));
    hipLaunchKernelGGL(xs::k_synth<float>, dim3(4096), dim3(256), 0, 0, ref, (int64_t)T, (int64_t)C, (int64_t)1, (int64_t)T, 0, 0, (uint64_t)1, (int64_t)0, 0.f, 0.f);
    hipLaunchKernelGGL(xs::k_synth<float>, dim3(4096), dim3(256), 0, 0, hist, (int64_t)T, (int64_t)C, (int64_t)1, (int64_t)T, 0, 1, (uint64_t)1, (int64_t)0, 0.f, 0.f);
    hipLaunchKernelGGL(xs::k_synth<float>, dim3(4096), dim3(256), 0, 0, tm, (int64_t)T, (int64_t)C, (int64_t)C, (int64_t)1, 0, 2, (uint64_t)1, (int64_t)0, 0.f, 0.f);
    // plan (as train3_bench.hip)
    std::vector<int> row_ptr(G + 1, 0);
    const int hw = W / 2;
    for (int t = 0; t < T; t++) row_ptr[t % G + 1] += std::min(T - 1, t - hw + W - 1) - std::max(0, t - hw) + 1;
    for (int i = 0; i < G; i++) row_ptr[i + 1] += row_ptr[i];
    std::vector<int> rows(row_ptr[G]), rp(row_ptr.begin(), row_ptr.end() - 1);
    for (int t = 0; t < T; t++) for (int j = 0; j < W; j++) { const int tt = t + j - hw; if (tt >= 0 && tt < T) rows[rp[t % G]++] = tt; }
    std::vector<int> t64((size_t)G * 64, T);
    for (int i = 0; i < G; i++) std::sort(rows.begin() + row_ptr[i], rows.begin() + row_ptr[i + 1]);
    for (int i = 0; i + 1 < G; i++) {
        std::vector<int> d;
        std::set_symmetric_difference(rows.begin() + row_ptr[i], rows.begin() + row_ptr[i + 1], rows.begin() + row_ptr[i + 1], rows.begin() + row_ptr[i + 2], std::back_inserter(d));
        std::copy(d.begin(), d.end(), t64.begin() + (size_t)(i + 1) * 64);
    }
    int *d_rp = upload(row_ptr), *d_rows = upload(rows), *d_t64 = upload(t64);
    std::vector<float> q(Q);
    for (int k = 0; k < Q; k++) q[k] = (float)((k + 0.5) / Q);
    float *dq = upload(q);
    auto kern = xs::k_train_cell3<11, true, false, 0>;
    const size_t lds = xs::CellCfg3<11>::lds_bytes;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    CK(hipMemset(flags, 0, 4));
    auto train = [&](hipStream_t s) {
        hipLaunchKernelGGL(kern, dim3(1024), dim3(1024), lds, s, ref, hist, (int64_t)T, (int64_t)T, d_rp, d_rows, G, dq, Q, (int)'+', 1, af, hq, (float *)nullptr,
                           flags, flags + 1, (int64_t)C, d_t64, (const int *)nullptr, (const int *)nullptr, (const int *)nullptr, 31);
    };
    auto transp = [&](hipStream_t s, int nblk) { hipLaunchKernelGGL(k_transpose_regs<32>, dim3(nblk), dim3(256), 0, s, tm, (int64_t)T, (int64_t)C, cm); };
    auto transp_lds = [&](hipStream_t s) {
        dim3 grid((unsigned)((C + 63) / 64), (unsigned)((T + 63) / 64));
        hipLaunchKernelGGL(xs::k_transpose_tile<float>, grid, dim3(256), 0, s, tm, (int64_t)T, (int64_t)C, (int64_t)C, (int64_t)1, cm, (int64_t)1, (int64_t)T);
    };
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1, f0, f1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
    auto ms = [&](hipEvent_t a, hipEvent_t b) { float m; CK(hipEventElapsedTime(&m, a, b)); return m; };
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0, sa)); train(sa); CK(hipEventRecord(e1, sa)); CK(hipStreamSynchronize(sa));
        printf("train alone                          %.3f ms\n", ms(e0, e1));
        CK(hipEventRecord(f0, sb)); transp_lds(sb); CK(hipEventRecord(f1, sb)); CK(hipStreamSynchronize(sb));
        printf("LDS tile transpose alone             %.3f ms\n", ms(f0, f1));
        for (int nblk : {256, 1024, 4096}) {
            CK(hipEventRecord(f0, sb)); transp(sb, nblk); CK(hipEventRecord(f1, sb)); CK(hipStreamSynchronize(sb));
            printf("register transpose alone, %4d blocks %.3f ms\n", nblk, ms(f0, f1));
        }
        for (int nblk : {256, 1024}) {
            CK(hipEventRecord(e0, sa)); train(sa); CK(hipEventRecord(e1, sa));
            CK(hipEventRecord(f0, sb)); transp(sb, nblk); CK(hipEventRecord(f1, sb));
            CK(hipDeviceSynchronize());
            printf("both, %4d transpose blocks: train %.3f ms, register transpose %.3f ms\n", nblk, ms(e0, e1), ms(f0, f1));
        }
        CK(hipEventRecord(e0, sa)); train(sa); CK(hipEventRecord(e1, sa));
        CK(hipEventRecord(f0, sb)); transp_lds(sb); CK(hipEventRecord(f1, sb));
        CK(hipDeviceSynchronize());
        printf("both, LDS tile transpose: train %.3f ms, transpose %.3f ms\n", ms(e0, e1), ms(f0, f1));
    }
    // check the register transpose
    std::vector<float> a((size_t)T * 64), b((size_t)64 * T);
    CK(hipMemcpy2D(a.data(), 64 * 4, tm, (size_t)C * 4, 64 * 4, T, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), cm, (size_t)64 * T * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (int c = 0; c < 64; c++) for (int t = 0; t < T; t++) bad += a[(size_t)t * 64 + c] != b[(size_t)c * T + t];
    printf("register transpose mismatches (first 64 cells): %zu\n", bad);
    return 0;
}
