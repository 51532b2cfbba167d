// sel_bench.hip — k_train_sel (un-windowed groups: BASELINE configs[1], EQM time.month) in isolation: the launch of
// xsdba_hip_eqm_train for 12 monthly groups on bench.py's eqm_month cubes, a few-second build, and the FNV-1a checksum of
// (hist_q, af) so that two builds can be told bit-identical.  -DXS_SEL_STOP=n builds stop every array pass after phase n.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Ixsdba_amd/csrc -o scripts/ubench/build/sel_bench scripts/ubench/sel_bench.hip
//   scripts/ubench/build/sel_bench [cells=64800] [reps=7] [groups: 12 (month) | 1 (time)]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "train_select.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static uint64_t fnv(const void *p, size_t n, uint64_t h = 1469598103934665603ull)
{
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char **argv)
{
    const int C = argc > 1 ? atoi(argv[1]) : 64800, reps = argc > 2 ? atoi(argv[2]) : 7, G = argc > 3 ? atoi(argv[3]) : 12;
    const int T = 10950, Q = 20;
    constexpr int EPT2 = 11, LT = 14;
    float *ref, *hist, *af, *hq, *dq;
    int *flags, *dg;
    CK(hipMalloc(&ref, (size_t)C * T * 4)); CK(hipMalloc(&hist, (size_t)C * T * 4));
    CK(hipMalloc(&af, (size_t)C * G * Q * 4)); CK(hipMalloc(&hq, (size_t)C * G * Q * 4));
    CK(hipMalloc(&flags, ((size_t)C + 1) * 4)); CK(hipMalloc(&dq, Q * 4)); CK(hipMalloc(&dg, T * 4));
    // bench.py's eqm_month cubes: generator 0, seed 1 (eqm_time: seed 0), arrays 0 (ref) and 1 (hist)
    const uint64_t seed = G == 1 ? 0 : 1;
    hipLaunchKernelGGL(xs::k_synth<float>, dim3(4096), dim3(256), 0, 0, ref, (int64_t)T, (int64_t)C, (int64_t)1, (int64_t)T, 0, 0, seed, (int64_t)0, 0.f, 0.f);
    hipLaunchKernelGGL(xs::k_synth<float>, dim3(4096), dim3(256), 0, 0, hist, (int64_t)T, (int64_t)C, (int64_t)1, (int64_t)T, 0, 1, seed, (int64_t)0, 0.f, 0.f);
    std::vector<float> q(Q);
    for (int k = 0; k < Q; k++) q[k] = (float)((k + 0.5) / Q);
    CK(hipMemcpy(dq, q.data(), Q * 4, hipMemcpyHostToDevice));
    std::vector<int> gidx(T);
    const int cum[13] = {0, 31, 59, 90, 120, 151, 181, 212, 243, 273, 304, 334, 365};
    for (int t = 0; t < T; t++) {
        const int d = t % 365;
        int m = 0;
        while (cum[m + 1] <= d) m++;
        gidx[t] = G == 1 ? 0 : m;
    }
    CK(hipMemcpy(dg, gidx.data(), T * 4, hipMemcpyHostToDevice));
    auto kern = xs::k_train_sel<EPT2, LT, false>;
    const size_t lds = xs::sel_layout(2 * EPT2 * xs::SEL_NT, G, Q, LT).bytes;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned nblk = (unsigned)std::min(C, 2 * 256);
    CK(hipMemset(flags, 0, 4));
    auto launch = [&]() {
        hipLaunchKernelGGL(kern, dim3(nblk), dim3(xs::SEL_NT), lds, 0, ref, hist, (int64_t)T, (int64_t)T, dg, G, dq, Q, (int)'+', 1, af, hq,
                           (float *)nullptr, flags, flags + 1, (int64_t)C);
    };
    launch();
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ms(reps);
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(e0, 0));
        launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms[r], e0, e1));
    }
    std::sort(ms.begin(), ms.end());
    const size_t n = (size_t)C * G * Q;
    std::vector<float> h(n);
    CK(hipMemcpy(h.data(), hq, n * 4, hipMemcpyDeviceToHost));
    uint64_t c = fnv(h.data(), n * 4);
    CK(hipMemcpy(h.data(), af, n * 4, hipMemcpyDeviceToHost));
    c = fnv(h.data(), n * 4, c);
    int nflag = 0;
    CK(hipMemcpy(&nflag, flags, 4, hipMemcpyDeviceToHost));
    printf("k_train_sel<11,14> G=%d stop=%d lds=%zu: %8.3f ms/launch (median of %d, min %.3f) = %7.3f ms per 64800 cells  checksum %016llx flagged %d\n",
           G, XS_SEL_STOP, lds, ms[reps / 2], reps, ms[0], ms[reps / 2] * 64800.0 / C, (unsigned long long)c, nflag);
    return 0;
}
