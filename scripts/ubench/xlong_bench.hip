// xlong_bench.hip — k_train_xlong (series of 22 529 .. 65 535 steps) in isolation, with its phase stops (debug_phase 2: the chunk
// sorts and their runs, 1: + merges, 3: + rank[] / rank-ordered steps, 0: everything).  Same plan as xsdba_hip_eqm_train for
// Grouper("time.dayofyear", 31) on T / 365 noleap years.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Ixsdba_amd/csrc -o scripts/ubench/build/xlong_bench scripts/ubench/xlong_bench.hip
//   scripts/ubench/build/xlong_bench [cells=4096] [reps=3] [only phase=-1] [iid NaN fraction=0] [T=43800]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <iterator>
#include <vector>

#include "train_xlong.hpp"
#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)

template <typename T> static T *upload(const std::vector<T> &v)
{
    T *d;
    CK(hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(T)));
    CK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

static uint64_t fnv(const void *p, size_t n, uint64_t h = 1469598103934665603ull)
{
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char **argv)
{
    const int C = argc > 1 ? atoi(argv[1]) : 4096, reps = argc > 2 ? atoi(argv[2]) : 3, only = argc > 3 ? atoi(argv[3]) : -1;
    const float nan_iid = argc > 4 ? (float)atof(argv[4]) : 0.f;
    const int T = argc > 5 ? atoi(argv[5]) : 43800, G = 365, W = 31, Q = 20;
    float *ref, *hist, *af, *hq;
    int *flags;
    CK(hipMalloc(&ref, (size_t)C * T * 4)); CK(hipMalloc(&hist, (size_t)C * T * 4));
    CK(hipMalloc(&af, (size_t)C * G * Q * 4)); CK(hipMalloc(&hq, (size_t)C * G * Q * 4));
    CK(hipMalloc(&flags, ((size_t)C + 1) * 4));
    hipLaunchKernelGGL(xs::k_synth<float>, dim3(4096), dim3(256), 0, 0, ref, (int64_t)T, (int64_t)C, (int64_t)1, (int64_t)T, 0, 0, (uint64_t)1, (int64_t)0, 0.f, nan_iid);
    hipLaunchKernelGGL(xs::k_synth<float>, dim3(4096), dim3(256), 0, 0, hist, (int64_t)T, (int64_t)C, (int64_t)1, (int64_t)T, 0, 1, (uint64_t)1, (int64_t)0, 0.f, nan_iid);
    std::vector<int> row_ptr(G + 1, 0);
    const int hw = W / 2;
    for (int t = 0; t < T; t++) row_ptr[t % G + 1] += std::min(T - 1, t - hw + W - 1) - std::max(0, t - hw) + 1;
    int max_row = 0;
    for (int i = 0; i < G; i++) { max_row = std::max(max_row, row_ptr[i + 1]); row_ptr[i + 1] += row_ptr[i]; }
    std::vector<int> rows(row_ptr[G]), rp(row_ptr.begin(), row_ptr.end() - 1);
    for (int t = 0; t < T; t++) for (int j = 0; j < W; j++) { const int tt = t + j - hw; if (tt >= 0 && tt < T) rows[rp[t % G]++] = tt; }
    for (int i = 0; i < G; i++) std::sort(rows.begin() + row_ptr[i], rows.begin() + row_ptr[i + 1]);
    std::vector<int> tog_ptr(G, 0), tog_idx;
    size_t maxlen = 0;
    for (int i = 0; i + 1 < G; i++) {
        const size_t b = tog_idx.size();
        std::set_symmetric_difference(rows.begin() + row_ptr[i], rows.begin() + row_ptr[i + 1], rows.begin() + row_ptr[i + 1], rows.begin() + row_ptr[i + 2], std::back_inserter(tog_idx));
        tog_ptr[i + 1] = (int)tog_idx.size();
        maxlen = std::max(maxlen, tog_idx.size() - b);
    }
    // positions (api.hip: d_xl_pos): the members of group 0 in time order, then group 1's, ...; the kernel's lists hold positions
    std::vector<int> pos(T), tim(T);
    {
        int k = 0;
        for (int g = 0; g < G; g++)
            for (int t = g; t < T; t += G) { pos[t] = k; tim[k] = t; k++; }
    }
    for (auto &v : rows) v = pos[v];
    for (auto &v : tog_idx) v = pos[v];
    std::vector<int> t128((size_t)G * 256, -1);
    for (int i = 0; i + 1 < G; i++) {
        const int b = tog_ptr[i], n = tog_ptr[i + 1] - b;
        int *row = t128.data() + (size_t)(i + 1) * 256;
        if (n > 256) { fprintf(stderr, "toggle list of %d steps\n", n); return 1; }
        std::copy(tog_idx.begin() + b, tog_idx.begin() + b + n, row);
    }
    int *d_rp = upload(row_ptr), *d_rows = upload(rows), *d_tp = upload(tog_ptr), *d_ti = upload(tog_idx), *d_t128 = upload(t128);
    int *d_pos = upload(pos), *d_tim = upload(tim);
    std::vector<float> q(Q);
    for (int k = 0; k < Q; k++) q[k] = (float)((k + 0.5) / Q);
    float *dq = upload(q);
    const unsigned nblk = (unsigned)std::min(C, 256);
    using XG = xs::XLongGeo<22>;
    unsigned char *scratch;
    CK(hipMalloc(&scratch, XG::scratch_bytes * nblk));
    printf("# k_train_xlong: %d cells x %d days, %d groups, window %d, %d quantiles; max row %d, longest toggle list %zu, iid NaN %.2f\n", C, T, G, W, Q, max_row, maxlen, nan_iid);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto report = [&](const char *what, auto launch, bool checksum) {
        launch();
        CK(hipDeviceSynchronize());
        std::vector<float> ms(reps);
        for (int r = 0; r < reps; r++) {
            CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms[r], e0, e1));
        }
        std::sort(ms.begin(), ms.end());
        printf("%-40s %8.3f ms/launch = %7.3f ms per 32400 cells = %6.1f ps per element", what, ms[reps / 2], ms[reps / 2] * 32400.0 / C, ms[reps / 2] * 1e9 / ((double)C * T));
        if (checksum) {
            const size_t n = (size_t)C * G * Q;
            std::vector<float> h(n);
            CK(hipMemcpy(h.data(), hq, n * 4, hipMemcpyDeviceToHost));
            uint64_t c = fnv(h.data(), n * 4);
            CK(hipMemcpy(h.data(), af, n * 4, hipMemcpyDeviceToHost));
            c = fnv(h.data(), n * 4, c);
            int nflag = 0;
            CK(hipMemcpy(&nflag, flags, 4, hipMemcpyDeviceToHost));
            printf("  checksum %016llx flagged %d", (unsigned long long)c, nflag);
        }
        printf("\n");
        fflush(stdout);
    };
    {
        const bool wide = !getenv("CSR");
        auto kern = T <= 2 * xs::LongCfg<8>::TPAD ? (wide ? xs::k_train_xlong<false, true, 22, 8> : xs::k_train_xlong<false, false, 22, 8>)
                                                   : (wide ? xs::k_train_xlong<false, true, 22> : xs::k_train_xlong<false, false, 22>);
        const size_t lds = XG::lds_bytes;
        CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        for (int phase : {2, 5, 6, 4, 1, 3, 0}) {
            if (only >= 0 && only != phase) continue;
            CK(hipMemset(flags, 0, 4));
            if (getenv("POISON")) {
                const int pz = atoi(getenv("POISON"));
                if (pz & 1) { CK(hipMemset(hq, 0xFF, (size_t)C * G * Q * 4)); CK(hipMemset(af, 0xFF, (size_t)C * G * Q * 4)); }
                if (pz & 2) CK(hipMemset(scratch, 0xFF, XG::scratch_bytes * nblk));
                if (pz & 4) CK(hipMemset(scratch, 0x00, XG::scratch_bytes * nblk));
            }
            auto launch = [&]() {
                hipLaunchKernelGGL(kern, dim3(nblk), dim3(1024), lds, 0, ref, hist, (int64_t)T, (int64_t)T, d_rp, d_rows, d_tp, d_ti, G, dq, Q, (int)'+', 1, af, hq,
                                   (float *)nullptr, flags, flags + 1, (int64_t)C, wide ? (const int *)d_t128 : (const int *)nullptr, d_pos, d_tim, scratch, phase);
            };
            char what[64];
            snprintf(what, sizeof what, "k_train_xlong: debug_phase %d", phase);
            report(what, launch, phase == 0);
        }
    }
    return 0;
}
