// lin2d.hpp — grouped interp="linear" of the quantile-mapping adjustment (qm_adjust with a monthly / seasonal /
// day-of-year table, _adjustment.py:594-676 -> utils.interp_on_quantiles :408-513 -> _interp_on_quantiles_2d :380-400):
// the reference hands the (G + 2) * Q nodes (hist_q value, group coordinate) of a cell — its table wrapped with the
// cyclic rows of add_cyclic_bounds (:284-314) — to scipy.griddata(method="linear"), i.e. a Delaunay triangulation and the
// piecewise-linear interpolant on it, evaluated at (sim value, fractional group coordinate of the time step), then
// nbutils._extrapolate_on_quantiles (nbutils.py:397-416) replaces what lies below / above the (group-interpolated) first /
// last node by the (group-interpolated) first / last factor, or NaN.
//
//   k_lin2d_build : one THREAD per cell builds the cell's node list and triangulation (delaunay.hpp) in global scratch
//   k_lin2d_adjust: one BLOCK per cell stages the triangulation in LDS (when it fits) and maps the series
//
// Grouped interp="cubic" takes the same route with griddata(method="cubic") = scipy's CloughTocher2DInterpolator: the build
// kernel also estimates the nodal gradients (Gauss-Seidel sweeps, delaunay.hpp: dt_gradients) and the adjust kernel evaluates
// the Clough-Tocher element instead of the plane of the triangle (`cubic` argument / layout flag).
#pragma once
#include "delaunay.hpp"
#include "device_math.hpp"

namespace xs {

struct Lin2DLayout {
    size_t px, py, pv, tv, tn, vt, rowb, hull, meta, gc, hdr, total, staged;  // byte offsets inside a cell's slice; `staged`: bytes the adjust kernel needs
    size_t grad, adjp, adji;  // cubic only: nodal gradients (staged), vertex adjacency in CSR form (build only)
};

__host__ __device__ inline Lin2DLayout lin2d_layout(int G, int Q, bool cubic = false)
{
    const size_t N = (size_t)(G + 2) * Q;
    Lin2DLayout L;
    size_t o = 0;
    L.hdr = o; o += 16;                       // int n, ntri
    L.meta = o; o += 4 * sizeof(double) * (size_t)(G + 2);  // per cyclic row: first x, last x, first y, last y
    L.gc = o; o += sizeof(double) * (size_t)(G + 2);        // the row coordinates (a copy per cell: staged with the rest)
    L.px = o; o += sizeof(double) * N;
    L.py = o; o += sizeof(double) * N;
    L.pv = o; o += sizeof(double) * N;
    L.tv = o; o += sizeof(uint16_t) * 6 * N;
    L.tn = o; o += sizeof(uint16_t) * 6 * N;
    L.vt = o; o += (sizeof(uint16_t) * N + 3) & ~(size_t)3;   // a triangle incident to every node (walk starts)
    L.rowb = o; o += sizeof(int) * (size_t)(G + 3);            // first node of every cyclic row
    o = (o + 7) & ~(size_t)7;
    L.grad = o; if (cubic) o += sizeof(double) * 2 * N;
    L.staged = o;
    L.hull = o; o += sizeof(uint16_t) * 4 * N;  // hull ring (3N) + insertion order (N): build only
    o = (o + 7) & ~(size_t)7;
    L.adjp = o; if (cubic) o += sizeof(int) * (N + 2);
    L.adji = o; if (cubic) o += sizeof(uint16_t) * (6 * N + 8);
    L.total = (o + 15) & ~(size_t)15;
    return L;
}

template <typename R>
__global__ void k_lin2d_build(const R *__restrict__ xq, const R *__restrict__ yq, const double *__restrict__ gcoord, int G, int Q,
                              int64_t C, unsigned char *__restrict__ scratch, size_t stride, int xq_is_row, int cubic)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const Lin2DLayout L = lin2d_layout(G, Q, cubic != 0);
    unsigned char *base = scratch + (size_t)c * stride;
    int *hdr = reinterpret_cast<int *>(base + L.hdr);
    double *meta = reinterpret_cast<double *>(base + L.meta);
    double *px = reinterpret_cast<double *>(base + L.px), *py = reinterpret_cast<double *>(base + L.py);
    double *pv = reinterpret_cast<double *>(base + L.pv);
    int *rowb = reinterpret_cast<int *>(base + L.rowb);
    double *gcs = reinterpret_cast<double *>(base + L.gc);
    int n = 0;
    for (int r = 0; r < G + 2; r++) {
        rowb[r] = n;
        gcs[r] = gcoord[r];
        const int g = ext_row(r, G);
        // xq_is_row: one abscissa row for every group and cell (QDM: the quantile nodes)
        const R *xr = xq_is_row ? xq : xq + ((size_t)c * G + g) * Q, *yr = yq + ((size_t)c * G + g) * Q;
        double xf = __builtin_nan(""), xl = xf, yf = xf, yl = xf;
        bool fx = false, fy = false;
        const int row_begin = n;
        for (int k = 0; k < Q; k++) {
            const R xv = xr[k], yv = yr[k];
            if (xv == xv) { if (!fx) { xf = (double)xv; fx = true; } xl = (double)xv; }
            if (yv == yv) { if (!fy) { yf = (double)yv; fy = true; } yl = (double)yv; }
            if (xv == xv && yv == yv) {
                // the nodes of a row must ascend strictly: of exactly repeated abscissae (tied quantiles) the first is kept
                // (Qhull keeps one of them too, which one is not defined: SURVEY §9.10)
                if (n > row_begin && !((double)xv > px[n - 1])) continue;
                px[n] = (double)xv; py[n] = gcoord[r]; pv[n] = (double)yv;
                n++;
            }
        }
        meta[4 * r] = xf; meta[4 * r + 1] = xl; meta[4 * r + 2] = yf; meta[4 * r + 3] = yl;
    }
    Tri2D d;
    d.px = px; d.py = py;
    d.tv = reinterpret_cast<uint16_t *>(base + L.tv);
    d.tn = reinterpret_cast<uint16_t *>(base + L.tn);
    d.hnext = reinterpret_cast<uint16_t *>(base + L.hull);
    d.hprev = d.hnext + (size_t)(G + 2) * Q;
    d.htri = d.hprev + (size_t)(G + 2) * Q;
    d.n = n;
    d.ntri = 0;
    rowb[G + 2] = n;
    hdr[0] = n;
    uint16_t *ord = d.htri + (size_t)(G + 2) * Q;
    // insertion order: by (x, y) for short tables (monthly, seasonal: 28.7 vs 74 ms per 64800 cells); row by row — the storage
    // order, also a valid sweep — for tall ones (day-of-year: 367 rows make the x-sweep's front long; 0.58 vs 2.3 s per 8192)
    if (G + 2 <= 32) dt_sort_order(d, ord);
    else for (int i = 0; i < n; i++) ord[i] = (uint16_t)i;
    const int ntri = dt_build(d, ord);
    hdr[1] = ntri;
    uint16_t *vt = reinterpret_cast<uint16_t *>(base + L.vt);
    for (int i = 0; i < n; i++) vt[i] = 0;
    for (int t = 0; t < ntri; t++) { vt[d.tv[3 * t]] = (uint16_t)t; vt[d.tv[3 * t + 1]] = (uint16_t)t; vt[d.tv[3 * t + 2]] = (uint16_t)t; }
    if (cubic) {  // nodal gradients of the Clough-Tocher interpolant (the hull ring is free again: scratch of the adjacency)
        int *adjp = reinterpret_cast<int *>(base + L.adjp);
        uint16_t *adji = reinterpret_cast<uint16_t *>(base + L.adji);
        dt_adjacency(d, adjp, adji, d.hnext);
        dt_gradients(d, pv, adjp, adji, reinterpret_cast<double *>(base + L.grad));
    }
}

// numpy.interp (compiled_base.c) on the G + 2 row coordinates: clamped outside, slope * (x - xp[j]) + fp[j] inside.
// The bracket j of every time step (xp[j] <= x < xp[j+1]; -1 below the first row, -2 at or above the last, -3 NaN) is the
// same for all cells: k_lin2d_rows finds it once.
__global__ void k_lin2d_rows(const double *__restrict__ newg, const double *__restrict__ xp, int n, int64_t T, int *__restrict__ jrow)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const double x = newg[t];
    int j;
    if (x != x) j = -3;
    else if (x <= xp[0]) j = -1;
    else if (x >= xp[n - 1]) j = -2;
    else {
        int lo = 0, hi = n - 1;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (xp[mid] <= x) lo = mid; else hi = mid; }
        j = lo;
    }
    jrow[t] = j;
}

__device__ __forceinline__ double np_interp_at(int j, double x, const double *xp, const double *fp, int stride, int n)
{
    if (j == -3) return x;
    if (j == -1) return fp[0];
    if (j == -2) return fp[(size_t)(n - 1) * stride];
    const double f0 = fp[(size_t)j * stride], f1 = fp[(size_t)(j + 1) * stride];
    if (x == xp[j]) return f0;
    const double slope = (f1 - f0) / (xp[j + 1] - xp[j]);
    double res = slope * (x - xp[j]) + f0;
    if (res != res) {
        res = slope * (x - xp[j + 1]) + f1;
        if (res != res && f0 == f1) res = f0;
    }
    return res;
}

// trend != nullptr: the series is detrended with it before the mapping and retrended after (DQM: the caller passes the
// scaled series).  dynamic LDS: L.staged bytes when use_lds, else 0.
template <typename R, bool USE_LDS>
__global__ void k_lin2d_adjust(const R *__restrict__ sim, int64_t T, int64_t C, int64_t sc, const double *__restrict__ newg,
                               const int *__restrict__ jrow, int G, int Q, const unsigned char *__restrict__ scratch,
                               size_t stride, int extrap, int kind, const R *__restrict__ trend, R *__restrict__ scen,
                               int64_t osc, const double *__restrict__ lookup, int cubic)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const Lin2DLayout L = lin2d_layout(G, Q, cubic != 0);
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        const unsigned char *base;
        if constexpr (USE_LDS) {
            const uint32_t *src = reinterpret_cast<const uint32_t *>(scratch + (size_t)c * stride);
            uint32_t *dst = reinterpret_cast<uint32_t *>(smem);
            for (size_t i = threadIdx.x; i < L.staged / 4; i += blockDim.x) dst[i] = src[i];
            __syncthreads();
            base = smem;
        } else {
            base = scratch + (size_t)c * stride;
        }
        const uint16_t *vt = reinterpret_cast<const uint16_t *>(base + L.vt);
        const int *rowb = reinterpret_cast<const int *>(base + L.rowb);
        const int *hdr = reinterpret_cast<const int *>(base + L.hdr);
        const double *meta = reinterpret_cast<const double *>(base + L.meta);
        const double *gcoord = reinterpret_cast<const double *>(base + L.gc);
        Tri2D d;
        d.px = reinterpret_cast<const double *>(base + L.px);
        d.py = reinterpret_cast<const double *>(base + L.py);
        const double *pv = reinterpret_cast<const double *>(base + L.pv);
        const double *grad = reinterpret_cast<const double *>(base + L.grad);
        d.tv = const_cast<uint16_t *>(reinterpret_cast<const uint16_t *>(base + L.tv));
        d.tn = const_cast<uint16_t *>(reinterpret_cast<const uint16_t *>(base + L.tn));
        d.hnext = d.hprev = d.htri = nullptr;
        d.n = hdr[0];
        d.ntri = hdr[1];
        const R *x = sim + c * sc;
        const R *trc = trend ? trend + (size_t)c * T : nullptr;
        const double *lk = lookup ? lookup + (size_t)c * T : nullptr;  // the abscissa to look up when it is not the series itself (QDM: sim_q)
        R *o = scen + c * osc;
        for (int64_t t = threadIdx.x; t < T; t += blockDim.x) {
            R v = x[t];
            R tr = (R)0;
            if (trc) { tr = trc[t]; v = apply_corr<R>(v, invert<R>(tr, kind), kind); }
            const double g = newg[t];
            R a = nan_of<R>();
            const double vx = lk ? lk[t] : (double)v;
            if (vx == vx && g == g && d.n > 0) {  // (a cell without a single valid node stays NaN: utils.py:384-390)
                // start the walk at a triangle of the node nearest in (row, x): the row next to g, bisection on its abscissae
                const int jr = jrow[t];
                const int r0 = jr == -1 ? 0 : (jr == -2 ? G : jr), r1 = r0 + 1;
                int r = (g - gcoord[r0] <= gcoord[r1] - g) ? r0 : r1;
                if (rowb[r + 1] == rowb[r]) r = r == r0 ? r1 : r0;
                int hint = 0;
                if (rowb[r + 1] > rowb[r]) {
                    int lo = rowb[r], hi = rowb[r + 1] - 1;
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (d.px[mid] < vx) lo = mid + 1; else hi = mid; }
                    hint = vt[lo];
                }
                const int tri = dt_locate(d, vx, g, hint);
                if (tri >= 0) a = (R)(cubic ? dt_eval_ct(d, pv, grad, tri, vx, g) : dt_eval(d, pv, tri, vx, g));
                // nbutils._extrapolate_on_quantiles: below / above the group-interpolated first / last node
                const double lo = extrap == 0 ? np_interp_at(jr, g, gcoord, meta, 4, G + 2) : 0.0;
                const double hi = extrap == 0 ? np_interp_at(jr, g, gcoord, meta + 1, 4, G + 2) : 0.0;
                // (only for extrapolation="constant": with "nan" the reference leaves griddata's own NaN outside the hull and
                // does not call it, utils.py:397-399)
                if (extrap == 0) {
                    if (vx < lo) a = (R)np_interp_at(jr, g, gcoord, meta + 2, 4, G + 2);
                    if (vx > hi) a = (R)np_interp_at(jr, g, gcoord, meta + 3, 4, G + 2);
                }
            }
            R y = kind ? apply_corr<R>(v, a, kind) : a;  // kind 0: the interpolated factor itself (utils.interp_on_quantiles)
            if (trc) y = apply_corr<R>(y, tr, kind);
            o[t] = y;
        }
        __syncthreads();
    }
}

}  // namespace xs
