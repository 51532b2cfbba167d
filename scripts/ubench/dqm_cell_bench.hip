// dqm_cell_bench.hip — k_dqm_adjust_cell (dqm_adjust as one kernel per cell) in isolation with its phase-stop instantiations:
// STOP 1 series in LDS, 2 window means, 3 regression, 4 table staged, 0 the whole kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Ixsdba_amd/csrc -o scripts/ubench/build/dqm_cell_bench scripts/ubench/dqm_cell_bench.hip
//   scripts/ubench/build/dqm_cell_bench [cells=64800] [reps=5]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "dqm_adjust_cell.hpp"

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)

template <typename T> static T *upload(const std::vector<T> &v)
{
    T *d;
    CK(hipMalloc(&d, v.size() * sizeof(T)));
    CK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static float frand()
{
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (float)((rng_state >> 40) * (1.0 / 16777216.0));
}

template <int NT, int STOP>
static float run(int C, int T, int G, int Q, int W, const float *sim, const float *af, const float *hq, const float *sc, const int *gidx,
                 const int *mem, float *scen, int reps, uint64_t *sum)
{
    const xs::DqmCellLayout L = xs::dqm_cell_layout(T, G, Q, W);
    auto k = xs::k_dqm_adjust_cell<NT, STOP>;
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < reps + 1; r++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(C), dim3(NT), L.bytes, 0, sim, (int64_t)T, (int64_t)T, af, hq, sc, G, Q, gidx, W, mem, 1, 0.0, 1.0, 0, 0, scen,
                           (int64_t)T);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r && ms < best) best = ms;
    }
    if (sum) {
        std::vector<float> h((size_t)C * T);
        CK(hipMemcpy(h.data(), scen, h.size() * 4, hipMemcpyDeviceToHost));
        uint64_t f = 1469598103934665603ull;
        for (float v : h) { uint32_t u; memcpy(&u, &v, 4); f = (f ^ u) * 1099511628211ull; }
        *sum = f;
    }
    return best;
}

int main(int argc, char **argv)
{
    const int C = argc > 1 ? atoi(argv[1]) : 64800, reps = argc > 2 ? atoi(argv[2]) : 5;
    const int T = 10950, G = 365, Q = 20, W = 31;
    std::vector<float> sim((size_t)C * T), af((size_t)C * G * Q), hq((size_t)C * G * Q), sc((size_t)C * G);
    for (auto &v : sim) v = 280.f + 10.f * frand();
    for (size_t i = 0; i < hq.size(); i++) { hq[i] = -5.f + 10.f * ((i % Q) + frand()) / Q; af[i] = frand() - 0.5f; }
    for (auto &v : sc) v = frand() - 0.5f;
    std::vector<int> gidx(T), mem((size_t)16 * G, -1);
    std::vector<int> cnt(G, 0);
    for (int t = 0; t < T; t++) {
        const int g = t % G, y = cnt[g]++;
        gidx[t] = g;
        uint32_t w = (uint32_t)mem[(size_t)(y / 2) * G + g];
        w = (y & 1) ? ((w & 0xFFFFu) | ((uint32_t)t << 16)) : ((w & 0xFFFF0000u) | (uint32_t)t);
        mem[(size_t)(y / 2) * G + g] = (int)w;
    }
    float *dsim = upload(sim), *daf = upload(af), *dhq = upload(hq), *dsc = upload(sc), *dscen;
    int *dg = upload(gidx), *dm = upload(mem);
    CK(hipMalloc(&dscen, (size_t)C * T * 4));
    uint64_t sum = 0;
#define RUN(NT, S) printf("NT %d stop %d: %.3f ms\n", NT, S, run<NT, S>(C, T, G, Q, W, dsim, daf, dhq, dsc, dg, dm, dscen, reps, S == 0 ? &sum : nullptr))
    RUN(512, 1); RUN(512, 2); RUN(512, 3); RUN(512, 4); RUN(512, 0);
    printf("checksum %016llx\n", (unsigned long long)sum);
    RUN(640, 1); RUN(640, 2); RUN(640, 3); RUN(640, 4); RUN(640, 0);
    printf("checksum %016llx\n", (unsigned long long)sum);
    RUN(768, 1); RUN(768, 2); RUN(768, 3); RUN(768, 4); RUN(768, 0);
    printf("checksum %016llx\n", (unsigned long long)sum);
    return 0;
}
