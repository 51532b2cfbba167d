// dqm_adjust_cell.hpp — EXPERIMENT, not part of the library (measured slower than the two kernels it would replace: DESIGN.md
// Appendix B.1, profiles/r04_dqm_fused_phases.txt; driver: dqm_cell_bench.hip).  dqm_adjust (_adjustment.py:679-780) for one cell in ONE kernel: scaling, the detrending fit
// (PolyDetrend degree 0 / 1 on the groups' window means, detrending.py:196-208) and the quantile mapping.
//
// The two-kernel path (k_group_trend, then k_qm_adjust in DQM mode) reads `sim` twice and sends the regression coefficients
// ([cell][G][3] float64) through global memory.  Here a block keeps a cell from start to end:
//   1. the cell's scaling factors and its (scaled) series go into LDS; the NaN-skipping centred window means are formed in
//      registers (a thread owns 25 consecutive steps and slides the window) and written back over the series;
//   2. a lane per group fits the line through the group's window means (<= 32 members: the plan's packed member table; affine
//      time axis) — the coefficients (x_mean, y_mean, slope) stay in LDS;
//   3. the cell's (hist_q, af) table is staged over the series (dead by now) and the block maps the series: the values come
//      from global memory again — the lines this very block read a few microseconds ago: L2 hits, no HBM traffic —, the trend
//      of an element is evaluated from the coefficients, the mapping is k_qm_adjust's.
// Every expression is the one the two kernels evaluate: the results are bit-identical (checked by wiring it into xsdba_hip_dqm_adjust).
// LDS: max(series, table) + coefficients + scaling = 79 KB for a day-of-year table of 20 nodes: two blocks per CU.
#pragma once
#include "kernels.hpp"

namespace xs {

struct DqmCellLayout {
    size_t off_coef, off_scal, off_rcp, off_xs, off_tab, bytes;
};

__host__ __device__ inline DqmCellLayout dqm_cell_layout(int64_t T, int G, int Q, int W)
{
    DqmCellLayout L;
    L.off_coef = 0;
    L.off_scal = (size_t)3 * G * sizeof(double);
    size_t o = (L.off_scal + (size_t)G * sizeof(float) + 15) & ~(size_t)15;
    L.off_rcp = o;
    L.off_xs = L.off_rcp + (size_t)(W + 1) * sizeof(double);
    L.off_tab = o;  // the table takes the place of the reciprocals and the series
    const size_t e1 = L.off_xs + (size_t)T * sizeof(float), e2 = L.off_tab + ((tab_lds_bytes<float>(G, Q) + 15) & ~(size_t)15);
    L.bytes = e1 > e2 ? e1 : e2;
    return L;
}

// conditions (host): float32, G > 1 groups of <= 32 members with the packed member table, affine time axis, W >= 1,
// T <= TREND_CH * blockDim, degree 0 / 1, interp nearest, the trend array not asked for
// STOP > 0: the phase-stop builds of scripts/ubench/dqm_cell_bench.hip (the product instantiates STOP = 0)
template <int NT, int STOP = 0>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(NT / 128, NT / 128))) void k_dqm_adjust_cell(const float *__restrict__ sim, int64_t T, int64_t sc, const float *__restrict__ af,
                                  const float *__restrict__ hist_q, const float *__restrict__ scaling, int G, int Q,
                                  const int *__restrict__ gidx, int W, const int *__restrict__ mem_t16, int degree, double tn0, double tnd,
                                  int extrap, int kind, float *__restrict__ scen, int64_t osc)
{
    using R = float;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const DqmCellLayout L = dqm_cell_layout(T, G, Q, W);
    double *coef = reinterpret_cast<double *>(smem + L.off_coef);
    float *scl = reinterpret_cast<float *>(smem + L.off_scal);
    double *rcp = reinterpret_cast<double *>(smem + L.off_rcp);
    R *xs = reinterpret_cast<R *>(smem + L.off_xs);
    const int64_t c = blockIdx.x;
    const R *x = sim + c * sc;
    const int Ti = (int)T;
    auto TN = [&](int64_t t) { return tn0 + tnd * (double)t; };
    // ---- 1. scaling factors, reciprocal table, the scaled series
    for (int i = threadIdx.x; i < G; i += blockDim.x) scl[i] = scaling[(size_t)c * G + i];
    for (int i = threadIdx.x; i <= W; i += blockDim.x) rcp[i] = i ? 1.0 / (double)i : 0.0;
    __syncthreads();
    {
        constexpr int NB = 8;
        for (int64_t t0 = threadIdx.x; t0 < T; t0 += (int64_t)NB * blockDim.x) {
            R v[NB];
            int gg[NB];
#pragma unroll
            for (int k = 0; k < NB; k++) {
                const int64_t t = t0 + (int64_t)k * blockDim.x, tc = t < T ? t : T - 1;
                v[k] = x[tc];
                gg[k] = gidx[tc];
            }
#pragma unroll
            for (int k = 0; k < NB; k++) {
                const int64_t t = t0 + (int64_t)k * blockDim.x;
                if (t < T) xs[t] = apply_corr<R>(v[k], gg[k] >= 0 ? scl[gg[k]] : nan_of<R>(), kind);
            }
        }
    }
    __syncthreads();
    if constexpr (STOP == 1) { scen[c * osc + threadIdx.x] = xs[threadIdx.x]; return; }
    // ---- window means (k_group_trend's in-register path: same sums, same order, same division)
    const int h = W / 2;
    constexpr int CH = TREND_CH;
    auto mean_of = [&](double s, int n) {
        double mv = __builtin_nan("");
        if (n) {
            const double dn = (double)n, rc = rcp[n];
            double q0 = s * rc;
            double r0 = __builtin_fma(-q0, dn, s);
            mv = __builtin_fma(r0, rc, q0);
        }
        return (R)mv;
    };
    auto slide = [&](int64_t t, bool first, double &s, int &n, double &big) {
        if (!first) {
            int64_t tin = t + W - 1 - h, tout = t - 1 - h;
            if (tin < T) { R v = xs[tin]; if (v == v) { s += (double)v; n++; big = fmax(big, fabs((double)v)); } }
            if (tout >= 0) { R v = xs[tout]; if (v == v) { s -= (double)v; n--; big = fmax(big, fabs((double)v)); } }
            first = big > 1.0e6 * fabs(s);
        }
        if (first) {
            s = 0; n = 0; big = 0;
            for (int j = 0; j < W; j++) {
                int64_t tt = t + j - h;
                if (tt >= 0 && tt < T) { R v = xs[tt]; if (v == v) { s += (double)v; n++; } }
            }
        }
    };
    if (W > 1) {
        const int64_t t0 = (int64_t)threadIdx.x * CH;
        R mreg[CH];
        double s = 0, big = 0;
        int n = 0;
#pragma unroll
        for (int k = 0; k < CH; k++) {
            mreg[k] = nan_of<R>();
            if (t0 + k < T) { slide(t0 + k, k == 0, s, n, big); mreg[k] = mean_of(s, n); }
        }
        __syncthreads();  // every window has been read: the series can be overwritten by its window means
#pragma unroll
        for (int k = 0; k < CH; k++) if (t0 + k < T) xs[t0 + k] = mreg[k];
        __syncthreads();
    }
    if constexpr (STOP == 2) { scen[c * osc + threadIdx.x] = xs[threadIdx.x]; return; }
    // ---- 2. the line through every group's window means (k_group_trend: a lane per group, members from the packed table)
    const R *m = xs;
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        uint32_t w[16];
#pragma unroll
        for (int u = 0; u < 16; u++) w[u] = (uint32_t)mem_t16[(size_t)u * G + g];
        double sx = 0, sy = 0;
        int n = 0;
#pragma unroll
        for (int y = 0; y < 32; y++) {
            const uint32_t t = (w[y / 2] >> (16 * (y & 1))) & 0xFFFFu;
            const R v = t != 0xFFFFu ? m[t] : nan_of<R>();
            if (v == v) { sx += TN(t); sy += (double)v; n++; }
        }
        const double xb = n ? sx / n : 0.0, yb = n ? sy / n : __builtin_nan("");
        double slope = 0.0;
        if (degree >= 1 && n >= 2) {
            double sxx = 0, sxy = 0;
#pragma unroll
            for (int y = 0; y < 32; y++) {
                const uint32_t t = (w[y / 2] >> (16 * (y & 1))) & 0xFFFFu;
                const R v = t != 0xFFFFu ? m[t] : nan_of<R>();
                if (v == v) { const double dx = TN(t) - xb; sxx += dx * dx; sxy += dx * ((double)v - yb); }
            }
            slope = sxx > 0 ? sxy / sxx : 0.0;
        }
        coef[3 * g] = xb; coef[3 * g + 1] = yb; coef[3 * g + 2] = slope;
    }
    __syncthreads();  // the window means are dead: the table takes their place
    if constexpr (STOP == 3) { scen[c * osc + threadIdx.x] = (R)coef[threadIdx.x]; return; }
    // ---- 3. the table, then the mapping (k_qm_adjust<float, true, true, true> with the coefficient trend)
    TabLds<R> tl = tab_carve<R>(smem + L.off_tab, G, Q);
    CellTab<R> ct = tab_stage<R>(tl, hist_q + (size_t)c * G * Q, Q, af + (size_t)c * G * Q, G, Q);
    R *o = scen + c * osc;
    if constexpr (STOP == 4) { o[threadIdx.x] = tl.xs[threadIdx.x]; return; }
    constexpr int EU = 4;
    for (int t0 = threadIdx.x; t0 < Ti; t0 += EU * blockDim.x) {
        R v[EU], tr[EU], a[EU];
        int g0[EU], nd[EU];
#pragma unroll
        for (int e = 0; e < EU; e++) {
            const int t = min(t0 + e * (int)blockDim.x, Ti - 1);
            v[e] = x[t];
            g0[e] = gidx[t];
            const R s = g0[e] >= 0 ? scl[g0[e]] : nan_of<R>();
            v[e] = apply_corr<R>(v[e], s, kind);
            const int g = g0[e];
            const double xt = tn0 + tnd * (double)t;
            tr[e] = g >= 0 ? (R)(coef[3 * g + 1] + coef[3 * g + 2] * (xt - coef[3 * g])) : nan_of<R>();
            v[e] = apply_corr<R>(v[e], invert<R>(tr[e], kind), kind);
        }
#pragma unroll
        for (int e = 0; e < EU; e++) {
            if (g0[e] < 0) { nd[e] = -1; a[e] = nan_of<R>(); }
            else if (ct.dense) a[e] = interp2d_nearest_dense<R>(v[e], g0[e], ct, extrap, nd[e]);
            else a[e] = interp2d_nearest<R, R>(v[e], g0[e], ct, extrap, nd[e]);
        }
#pragma unroll
        for (int e = 0; e < EU; e++) {
            const int t = t0 + e * (int)blockDim.x;
            if (t < Ti) o[t] = apply_corr<R>(apply_corr<R>(v[e], a[e], kind), tr[e], kind);
        }
    }
}

}  // namespace xs
