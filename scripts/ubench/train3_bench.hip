// train3_bench.hip — k_train_cell3 (the metric's dominant kernel) in isolation: same plan, same data generator and same
// launch as xsdba_hip_eqm_train for Grouper("time.dayofyear", 31), but a 3-second build — the bench for kernel experiments —
// and the phase-stop instantiations (STOP = 1..4) whose differences attribute time and PMC counters to the kernel's phases.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Ixsdba_amd/csrc -Iscripts/ubench -o scripts/ubench/build/train3_bench scripts/ubench/train3_bench.hip
//   scripts/ubench/build/train3_bench [cells=64800] [reps=5] [stop=-1: all phases | 0..4]
// Prints per phase build: ms per launch and the FNV-1a checksum of (hist_q, af) — two builds of the full kernel (STOP 0) that
// print the same checksum computed the same tables, bit for bit.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <vector>

#include "train_cell3.hpp"
#include "train_cell4.hpp"
#include "train_chunk.hpp"
#include <map>

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

struct Plan {
    int G = 0, max_row = 0;
    int *d_row_ptr = nullptr, *d_row_sorted = nullptr, *d_tog64 = nullptr, *d_mem_ptr = nullptr, *d_mem_idx = nullptr, *d_mem_t32 = nullptr;
    int *d_rows_pos = nullptr, *d_tog64_pos = nullptr, *d_time_of = nullptr;  // k_train_cell4: the lists in positions
    std::vector<int> h_row_ptr, h_rows;  // host copies (rows as sorted sets of time steps)
};

// the chunk tables of k_train_chunk (api.hip: build_chunks)
struct Chunks {
    int nch = 0;
    int *d_g0 = nullptr, *d_time = nullptr, *d_tog = nullptr;
};
static Chunks make_chunks(const Plan &pl, int NE)
{
    const int G = pl.G, MAXE = NE - 32, PADE = NE - 1;
    auto rowb = [&](int g) { return pl.h_rows.begin() + pl.h_row_ptr[g]; };
    auto rowe = [&](int g) { return pl.h_rows.begin() + pl.h_row_ptr[g + 1]; };
    Chunks ch;
    std::vector<std::vector<int>> un;
    int GC = 0;
    for (int nch = 1; nch <= G; nch++) {
        GC = (G + nch - 1) / nch;
        un.assign((G + GC - 1) / GC, {});
        bool ok = true;
        for (size_t k = 0; k < un.size() && ok; k++) {
            std::vector<int> &u = un[k];
            for (int g = (int)k * GC; g < std::min(G, ((int)k + 1) * GC); g++) u.insert(u.end(), rowb(g), rowe(g));
            std::sort(u.begin(), u.end());
            u.erase(std::unique(u.begin(), u.end()), u.end());
            ok = (int)u.size() <= MAXE;
        }
        if (ok) break;
    }
    ch.nch = (int)un.size();
    std::vector<int> g0(ch.nch + 1), tim((size_t)ch.nch * NE, -1), tog((size_t)G * 64, PADE);
    for (int k = 0; k < ch.nch; k++) {
        g0[k] = k * GC;
        const int r0 = k * GC, r1 = std::min(G, (k + 1) * GC), n = r1 - r0;
        const int nA = (n + 2) / 3, nB = (n - nA + 1) / 2, gb = r0 + nB;
        std::map<int, int> loc;
        for (size_t e = 0; e < un[k].size(); e++) { tim[(size_t)k * NE + e] = un[k][e]; loc[un[k][e]] = (int)e; }
        for (auto it = rowb(gb); it != rowe(gb); ++it) tim[(size_t)k * NE + loc[*it]] |= 1 << 30;
        for (int g = r0 + 1; g < r1; g++) {
            std::vector<int> d;
            std::set_symmetric_difference(rowb(g - 1), rowe(g - 1), rowb(g), rowe(g), std::back_inserter(d));
            if (d.size() > 64) { fprintf(stderr, "toggle list of %zu entries\n", d.size()); exit(1); }
            int *t = tog.data() + (size_t)g * 64;
            for (size_t j = 0; j < d.size(); j++) t[j] = loc[d[j]];
            for (size_t j = d.size(), e = 0; j + 1 < 64 && e < d.size(); j += 2, e++) t[j] = t[j + 1] = loc[d[e]];
        }
    }
    g0[ch.nch] = G;
    ch.d_g0 = upload(g0); ch.d_time = upload(tim); ch.d_tog = upload(tog);
    printf("# chunks: %d of up to %d groups, NE %d\n", ch.nch, GC, NE);
    return ch;
}


// the plan of xsdba_hip_eqm_train (api.hip: get_plan) for day-of-year groups on a noleap calendar and a centred window
static Plan make_plan(int T, int G, int W)
{
    Plan pl;
    pl.G = G;
    std::vector<int> row_ptr(G + 1, 0);
    const int hw = W / 2;
    for (int t = 0; t < T; t++) {
        const int lo = std::max(0, t - hw), hi = std::min(T - 1, t - hw + W - 1);
        row_ptr[t % G + 1] += hi - lo + 1;
    }
    for (int i = 0; i < G; i++) { pl.max_row = std::max(pl.max_row, row_ptr[i + 1]); row_ptr[i + 1] += row_ptr[i]; }
    std::vector<int> rows(row_ptr[G]), rp(row_ptr.begin(), row_ptr.end() - 1);
    for (int t = 0; t < T; t++)
        for (int j = 0; j < W; j++) {
            const int tt = t + j - hw;
            if (tt >= 0 && tt < T) rows[rp[t % G]++] = tt;
        }
    std::vector<int> t64((size_t)G * 64, T);  // (padding: time step T, a padding step of the series)
    for (int i = 0; i < G; i++) std::sort(rows.begin() + row_ptr[i], rows.begin() + row_ptr[i + 1]);
    for (int i = 0; i + 1 < G; i++) {
        std::vector<int> d;
        std::set_symmetric_difference(rows.begin() + row_ptr[i], rows.begin() + row_ptr[i + 1], rows.begin() + row_ptr[i + 1],
                                      rows.begin() + row_ptr[i + 2], std::back_inserter(d));
        if (d.size() > 64) { fprintf(stderr, "toggle list of %zu entries\n", d.size()); exit(1); }
        std::copy(d.begin(), d.end(), t64.begin() + (size_t)(i + 1) * 64);
        // the lanes behind the list flip pairs of its own entries (twice = not at all) instead of all hitting the padding step's
        // one scratch word; an odd lane out keeps the padding step
        for (size_t j = d.size(), e = 0; j + 1 < 64 && e < d.size(); j += 2, e++) t64[(size_t)(i + 1) * 64 + j] = t64[(size_t)(i + 1) * 64 + j + 1] = d[e];
    }
    pl.d_row_ptr = upload(row_ptr);
    pl.d_row_sorted = upload(rows);
    pl.h_row_ptr = row_ptr; pl.h_rows = rows;
    pl.d_tog64 = upload(t64);
    // the members of every group in time order + the packed member table (api.hip: d_mem_t32), for dqm_train's group sums
    std::vector<int> mem_ptr(G + 1, 0), mem_idx, mt((size_t)16 * G, -1);
    for (int g = 0; g < G; g++) {
        int y = 0;
        for (int t = g; t < T; t += G, y++) {
            mem_idx.push_back(t);
            uint32_t w = (uint32_t)mt[(size_t)(y / 2) * G + g];
            w = (y & 1) ? ((w & 0x0000FFFFu) | ((uint32_t)t << 16)) : ((w & 0xFFFF0000u) | (uint32_t)t);
            mt[(size_t)(y / 2) * G + g] = (int)w;
        }
        mem_ptr[g + 1] = (int)mem_idx.size();
    }
    pl.d_mem_ptr = upload(mem_ptr); pl.d_mem_idx = upload(mem_idx); pl.d_mem_t32 = upload(mt);
    // positions (api.hip: get_plan): the members of group 0 in time order, then group 1's, ...; position T = time step T (padding)
    std::vector<int> pos(xs::Cell4Cfg::TPAD, 0), tim(T, 0), rows_p(rows.size()), t64p(t64.size());
    for (size_t k = 0; k < mem_idx.size(); k++) { pos[mem_idx[k]] = (int)k; tim[k] = mem_idx[k]; }
    for (int t = T; t < xs::Cell4Cfg::TPAD; t++) pos[t] = t;
    for (size_t k = 0; k < rows.size(); k++) rows_p[k] = pos[rows[k]];
    for (size_t k = 0; k < t64.size(); k++) t64p[k] = pos[t64[k]];
    pl.d_rows_pos = upload(rows_p); pl.d_tog64_pos = upload(t64p); pl.d_time_of = upload(pos);
    return pl;
}

static uint64_t fnv(const void *p, size_t n, uint64_t h = 1469598103934665603ull)
{
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

template <int STOP, bool DQM = false>
static void run(const char *what, const float *ref, const float *hist, int T, int C, const Plan &pl, const float *dq, int Q, float *af, float *hq,
                int *flags, int reps, bool checksum, float *scaling = nullptr)
{
    constexpr int EPT = 11;
    auto kern = xs::k_train_cell3<EPT, true, DQM, STOP>;
    const size_t lds = xs::CellCfg3<EPT>::lds_bytes + (DQM ? ((size_t)pl.G + 1) * sizeof(double) : 0);
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned nblk = (unsigned)std::min(C, 4 * 256);
    CK(hipMemset(flags, 0, 4));
    auto launch = [&]() {
        hipLaunchKernelGGL(kern, dim3(nblk), dim3(1024), lds, 0, ref, hist, (int64_t)T, (int64_t)T, pl.d_row_ptr, pl.d_row_sorted, pl.G, dq, Q, (int)'+', DQM ? 2 : 1,
                           af, hq, scaling, flags, flags + 1, (int64_t)C, pl.d_tog64, DQM ? pl.d_mem_ptr : (const int *)nullptr,
                           DQM ? pl.d_mem_idx : (const int *)nullptr, DQM ? pl.d_mem_t32 : (const int *)nullptr, 31);
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
    printf("%-44s stop=%d  %8.3f ms/launch (median of %d, min %.3f)  = %7.3f ms per 64800 cells", what, STOP, ms[reps / 2], reps, ms[0],
           ms[reps / 2] * 64800.0 / C);
    if (checksum && STOP == 0) {
        const size_t n = (size_t)C * pl.G * Q;
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
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

template <int STOP>
static void run4(const char *what, const float *ref, const float *hist, int T, int C, const Plan &pl, const float *dq, int Q, float *af, float *hq,
                 int *flags, int reps, bool checksum)
{
    auto kern = xs::k_train_cell4<true, STOP>;
    const size_t lds = xs::Cell4Cfg::lds_bytes;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int mult = getenv("K4_MULT") ? atoi(getenv("K4_MULT")) : 2;
    const unsigned nblk = (unsigned)std::min(C, mult * 256);
    static unsigned char *scratch = nullptr;
    static int *cu_count = nullptr;
    if (!scratch) { CK(hipMalloc(&scratch, xs::Cell4Cfg::scratch_bytes * 4096)); CK(hipMalloc(&cu_count, 2048 * 4)); }
    const int stagger = getenv("K4_STAGGER") ? atoi(getenv("K4_STAGGER")) : 0;
    int nb = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 512, lds));
    CK(hipMemset(flags, 0, 4));
    CK(hipMemset(hq, 0, (size_t)C * pl.G * Q * 4));
    CK(hipMemset(af, 0, (size_t)C * pl.G * Q * 4));
    auto launch = [&]() {
        if (stagger > 0) CK(hipMemsetAsync(cu_count, 0, 2048 * 4, 0));
        hipLaunchKernelGGL(kern, dim3(nblk), dim3(512), lds, 0, ref, hist, (int64_t)T, (int64_t)T, pl.d_row_ptr, pl.d_rows_pos, pl.G, dq, Q, (int)'+', 1,
                           af, hq, flags, flags + 1, (int64_t)C, pl.d_tog64_pos, pl.d_time_of, scratch, cu_count, stagger);
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
    printf("k4 (%d blocks/CU) %-36s stop=%d  %8.3f ms/launch (median of %d, min %.3f)  = %7.3f ms per 64800 cells", nb, what, STOP, ms[reps / 2], reps, ms[0],
           ms[reps / 2] * 64800.0 / C);
    if (checksum && STOP == 0) {
        const size_t n = (size_t)C * pl.G * Q;
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
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

template <int NJ, int STOP>
static void runc(const char *what, const float *ref, const float *hist, int T, int C, const Plan &pl, const Chunks &ch, const float *dq, int Q, float *af,
                 float *hq, int *flags, int reps, bool checksum)
{
    using Cfg = xs::ChunkCfg<NJ>;
    auto kern = xs::k_train_chunk<NJ, STOP>;
    const size_t lds = Cfg::lds_bytes;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int nb = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 256, lds));
    const int mult = getenv("KC_MULT") ? atoi(getenv("KC_MULT")) : nb;
    const unsigned nblk = (unsigned)std::min<int64_t>(((int64_t)C * ch.nch + 3) / 4, (int64_t)mult * 256);
    static int *cell_flag = nullptr;
    if (!cell_flag) CK(hipMalloc(&cell_flag, (size_t)C * 4));
    CK(hipMemset(cell_flag, 0, (size_t)C * 4));
    CK(hipMemset(flags, 0, 4));
    CK(hipMemset(hq, 0, (size_t)C * pl.G * Q * 4));
    CK(hipMemset(af, 0, (size_t)C * pl.G * Q * 4));
    xs::ChunkTabs tb{ch.d_g0, ch.d_time, ch.d_tog, ch.nch};
    auto launch = [&]() {
        hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds, 0, ref, hist, (int64_t)T, (int64_t)C, pl.G, dq, Q, (int)'+', 1, af, hq, flags, flags + 1, cell_flag, tb);
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
    printf("chunk<%d> (%d blocks/CU, %u blocks) %-30s stop=%d  %8.3f ms/launch (median of %d, min %.3f)  = %7.3f ms per 64800 cells", NJ, nb, nblk, what, STOP,
           ms[reps / 2], reps, ms[0], ms[reps / 2] * 64800.0 / C);
    if (checksum && STOP == 0) {
        const size_t n = (size_t)C * pl.G * Q;
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
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main(int argc, char **argv)
{
    const int C = argc > 1 ? atoi(argv[1]) : 64800, reps = argc > 2 ? atoi(argv[2]) : 5, stop = argc > 3 ? atoi(argv[3]) : -1;
    const int T = 10950, G = 365, W = 31, Q = 20;
    float *ref, *hist, *af, *hq, *dq;
    int *flags;
    CK(hipMalloc(&ref, (size_t)C * T * 4));
    CK(hipMalloc(&hist, (size_t)C * T * 4));
    CK(hipMalloc(&af, (size_t)C * G * Q * 4));
    CK(hipMalloc(&hq, (size_t)C * G * Q * 4));
    CK(hipMalloc(&flags, ((size_t)C + 1) * 4));
    // bench.py's eqm_doy31 cubes: generator 0, seed 1, arrays 0 (ref) and 1 (hist)
    hipLaunchKernelGGL(xs::k_synth<float>, dim3(4096), dim3(256), 0, 0, ref, (int64_t)T, (int64_t)C, (int64_t)1, (int64_t)T, 0, 0, (uint64_t)1, (int64_t)0, 0.f, 0.f);
    hipLaunchKernelGGL(xs::k_synth<float>, dim3(4096), dim3(256), 0, 0, hist, (int64_t)T, (int64_t)C, (int64_t)1, (int64_t)T, 0, 1, (uint64_t)1, (int64_t)0, 0.f, 0.f);
    std::vector<float> q(Q);
    for (int k = 0; k < Q; k++) q[k] = (float)((k + 0.5) / Q);
    dq = upload(q);
    Plan pl = make_plan(T, G, W);
    CK(hipDeviceSynchronize());
    printf("# k_train_cell3<11, P10, EQM>: %d cells x %d days, %d groups, window %d, %d quantiles; max row %d\n", C, T, G, W, Q, pl.max_row);
    // x0 = ref, x1 = hist in the kernel's argument order (mode 1: hist first, then ref)
    if (stop < 0 || stop == 1) run<1>("load (+ keys, min / max)", ref, hist, T, C, pl, dq, Q, af, hq, flags, reps, false);
    if (stop < 0 || stop == 2) run<2>("load + sort", ref, hist, T, C, pl, dq, Q, af, hq, flags, reps, false);
    if (stop < 0 || stop == 3) run<3>("load + sort + build / copies / fast-forward", ref, hist, T, C, pl, dq, Q, af, hq, flags, reps, false);
    if (stop < 0 || stop == 4) run<4>("... + group loop without the select", ref, hist, T, C, pl, dq, Q, af, hq, flags, reps, false);
    if (stop < 0 || stop == 0) run<0>("full kernel", ref, hist, T, C, pl, dq, Q, af, hq, flags, reps, true);
    if (getenv("K4")) {  // the two-blocks-per-CU kernel (train_cell4.hpp)
        const int s4 = atoi(getenv("K4"));  // 1: every phase, 2: the full kernel only
        if (s4 == 1) {
            run4<1>("load", ref, hist, T, C, pl, dq, Q, af, hq, flags, reps, false);
            run4<2>("load + sort", ref, hist, T, C, pl, dq, Q, af, hq, flags, reps, false);
            run4<5>("... + values, rank copy-out", ref, hist, T, C, pl, dq, Q, af, hq, flags, reps, false);
            run4<3>("... + build / copies / fast-forward", ref, hist, T, C, pl, dq, Q, af, hq, flags, reps, false);
            run4<4>("... + group loop without the select", ref, hist, T, C, pl, dq, Q, af, hq, flags, reps, false);
        }
        run4<0>("full kernel", ref, hist, T, C, pl, dq, Q, af, hq, flags, reps, true);
    }
    if (getenv("KC")) {  // a wave per (cell, chunk of groups): train_chunk.hpp
        const int sc_ = atoi(getenv("KC"));  // 1: every phase, 2: the full kernel only
        constexpr int NJ = 26;
        Chunks ch = make_chunks(pl, xs::ChunkCfg<NJ>::NE);
        if (sc_ == 1) {
            runc<NJ, 1>("gather + keys", ref, hist, T, C, pl, ch, dq, Q, af, hq, flags, reps, false);
            runc<NJ, 2>("... + sort", ref, hist, T, C, pl, ch, dq, Q, af, hq, flags, reps, false);
            runc<NJ, 3>("... + build / copies / fast-forward", ref, hist, T, C, pl, ch, dq, Q, af, hq, flags, reps, false);
            runc<NJ, 4>("... + loop without the select", ref, hist, T, C, pl, ch, dq, Q, af, hq, flags, reps, false);
        }
        runc<NJ, 0>("full kernel", ref, hist, T, C, pl, ch, dq, Q, af, hq, flags, reps, true);
    }
    if (getenv("DQM")) {  // dqm_train (mode 2): group sums by the load hook, normalised quantiles, scaling
        float *scaling;
        CK(hipMalloc(&scaling, (size_t)C * G * 4));
        printf("# k_train_cell3<11, P10, DQM>\n");
        run<1, true>("DQM: load + group sums", ref, hist, T, C, pl, dq, Q, af, hq, flags, reps, false, scaling);
        run<2, true>("DQM: ... + sort", ref, hist, T, C, pl, dq, Q, af, hq, flags, reps, false, scaling);
        run<3, true>("DQM: ... + build / copies / fast-forward", ref, hist, T, C, pl, dq, Q, af, hq, flags, reps, false, scaling);
        run<4, true>("DQM: ... + group loop without the select", ref, hist, T, C, pl, dq, Q, af, hq, flags, reps, false, scaling);
        run<0, true>("DQM: full kernel", ref, hist, T, C, pl, dq, Q, af, hq, flags, reps, true, scaling);
    }
    return 0;
}
