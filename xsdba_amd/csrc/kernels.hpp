// kernels.hpp — gfx950 kernels of the quantile-mapping hot path (generic "row" formulation).
//
// Layout inside the kernels is cell-major: series x[c*sc + t*st] with st == 1 on the fast paths (the
// host side transposes time-major cubes once on the way in / out, see api.hip).  One workgroup owns one
// (cell, group) row for training and one cell for adjusting; everything a row/cell needs is staged in
// LDS once (sort buffer or factor table), so HBM sees each input element once per kernel.
#pragma once
#include "device_math.hpp"

namespace xs {

// ---------------------------------------------------------------------------------------------
// block-wide helpers (wave = 64 lanes)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// red: LDS scratch of >= 2*(blockDim/64)+2 doubles.  Returns the block-wide (sum, count) to every thread.
__device__ __forceinline__ void block_sum_count(double &s, int &n, double *red)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    s = wave_sum(s);
    n = wave_sum(n);
    __syncthreads();
    if (lane == 0) { red[2 * wid] = s; red[2 * wid + 1] = (double)n; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ts = 0, tn = 0;
        for (int w = 0; w < nw; w++) { ts += red[2 * w]; tn += red[2 * w + 1]; }
        red[2 * nw] = ts; red[2 * nw + 1] = tn;
    }
    __syncthreads();
    s = red[2 * nw];
    n = (int)red[2 * nw + 1];
    __syncthreads();
}

// In-LDS bitonic sort of P (power of two) keys by the whole block.
template <typename U> __device__ __forceinline__ void block_bitonic_sort(U *keys, int P)
{
    const int half = P >> 1;
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < half; i += blockDim.x) {
                int l = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                int r = l | j;
                U a = keys[l], b = keys[r];
                bool up = (l & k) == 0;
                if ((a > b) == up) { keys[l] = b; keys[r] = a; }
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K1: rows -> sort -> type-7 quantiles (-> correction factors).
//   mode 0: out_hq[c][g][:] = quantiles of x0's rows                      (nbutils.quantile, a4)
//   mode 1: EQM train: hist_q, af = corr(hist_q, ref_q)                   (_adjustment.py:253-286)
//   mode 2: DQM train: rows normalised by their NaN-skipping mean first,  (_adjustment.py:150-190)
//           scaling = corr(mean(hist), mean(ref))
//   mode 3: out_hq[c] = np.nanquantile(row c, q[c])  (nbutils._vecquantiles :151-161; G must be 1)
// grid.x = n_cell * n_groups (block b -> cell b / G, group b % G); dynamic LDS:
//   [red: 2*(NT/64)+2 doubles][keys: P_max * sizeof(U)][rq: Q * sizeof(R)]
// ---------------------------------------------------------------------------------------------
template <typename R>
__global__ void k_train_rows(const R *__restrict__ x0, const R *__restrict__ x1, int64_t st, int64_t sc,
                             const int *__restrict__ row_ptr, const int *__restrict__ row_idx, int G,
                             const R *__restrict__ q, int Q, int kind, int mode, R *__restrict__ af,
                             R *__restrict__ out_hq, R *__restrict__ scaling, int P_max,
                             const int *__restrict__ cell_list, const int *__restrict__ n_list_dev)
{
    using U = typename KeyOf<R>::U;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nw = (blockDim.x + 63) >> 6;
    double *red = reinterpret_cast<double *>(smem);
    U *keys = reinterpret_cast<U *>(red + 2 * nw + 2);
    R *vals = reinterpret_cast<R *>(keys);
    R *rq = reinterpret_cast<R *>(keys + P_max);

    // full mode: one (cell, group) row per block.  subset mode (cell_list): the rows of the *n_list_dev flagged
    // cells are walked grid-stride, the count being read on the device so that the host never has to wait for it
    const int64_t total = n_list_dev ? (int64_t)(*n_list_dev) * G : (int64_t)gridDim.x;
    for (int64_t b = blockIdx.x; b < total; b += gridDim.x) {
    const int64_t ci = b / G;
    const int g = (int)(b - ci * G);
    const int64_t c = cell_list ? (int64_t)cell_list[ci] : ci;  // subset mode: redo the cells k_train_cell flagged
    const int r0 = row_ptr[g], L = row_ptr[g + 1] - r0;
    int P = 2;
    while (P < L) P <<= 1;
    const size_t obase = ((size_t)c * G + g) * (size_t)Q;

    R mu[2] = {nan_of<R>(), nan_of<R>()};
    const int npass = (mode == 0 || mode == 3) ? 1 : 2;
    for (int pass = 0; pass < npass; pass++) {
        const R *x = (pass == 0 ? x0 : x1) + c * sc;
        R inv = (R)0;
        if (mode == 2) {
            double s = 0; int n = 0;
            for (int i = threadIdx.x; i < L; i += blockDim.x) {
                R v = x[(int64_t)row_idx[r0 + i] * st];
                vals[i] = v;
                if (v == v) { s += (double)v; n++; }
            }
            block_sum_count(s, n, red);
            mu[pass] = n ? (R)(s / (double)n) : nan_of<R>();
            inv = invert<R>(mu[pass], kind);
        }
        int nvalid = 0;
        for (int i = threadIdx.x; i < P; i += blockDim.x) {
            U k = KeyOf<R>::MAXK;
            if (i < L) {
                R v = mode == 2 ? apply_corr<R>(vals[i], inv, kind) : x[(int64_t)row_idx[r0 + i] * st];
                k = to_key(v);
                nvalid += (v == v);
            }
            keys[i] = k;
        }
        double dummy = 0;
        block_sum_count(dummy, nvalid, red);  // also the barrier between key writes and the sort
        block_bitonic_sort<U>(keys, P);
        auto load = [&](int i) { return from_key(keys[i]); };
        if (mode == 3) {  // nbutils._vecquantiles: one quantile per row, q[c] (NaN q -> NaN)
            if (threadIdx.x == 0) out_hq[c] = vecquantile_sorted<R>(load, nvalid, q[c]);
            __syncthreads();
            continue;
        }
        for (int k = threadIdx.x; k < Q; k += blockDim.x) {
            R v = quantile_sorted<R>(load, nvalid, q[k]);
            if (mode == 0) out_hq[obase + k] = v;
            else if (pass == 0) rq[k] = v;  // ref_q stays on chip
            else {
                out_hq[obase + k] = v;
                af[obase + k] = get_corr<R>(v, rq[k], kind);
            }
        }
        __syncthreads();
    }
    if (mode == 2 && threadIdx.x == 0) scaling[(size_t)c * G + g] = get_corr<R>(mu[1], mu[0], kind);
    __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// table staging shared by the adjust kernels: copies a cell's [G][Q] (xq, yq) into LDS, compacts the
// NaN pairs of every row in place and fills the row metadata (nbutils._first_and_last_nonnull).
// xq_row_stride == 0 broadcasts one abscissa row to all groups (QDM: the quantile nodes).
// A table carved with shared_x keeps that ONE row instead of G copies (half the LDS); it can only hold cells whose
// factors are NaN-free (no per-row compaction) — tab_stage answers any_valid == -1 otherwise and the caller stages
// the cell into a full-layout table.
// ---------------------------------------------------------------------------------------------
template <typename R> struct TabLds {
    R *xs, *ys;
    int *rn;
    double *xfirst, *xlast;
    R *yfirst, *ylast;
    int *any;
    int xstride;  // Q, or 0 for the shared abscissa row
    double *spl;  // [3 Q] cubic-spline second derivatives of row 0 + work space (tab_spline)
};

// Row stride of a cell's [G][Q] tables in LDS.  The lanes of a wave search the rows of consecutive groups at the same offsets:
// float rows a multiple of 16 dwords long would put sixteen lanes on every LDS bank (nquantiles = 16: the adjustment 4.3 ms per
// 64 800 cells against 2.6 ms with 20 nodes) — they get four dwords of padding, which keeps the 16-byte staging.
template <typename R> __host__ __device__ inline int tab_row_stride(int G, int Q) { return (sizeof(R) == 4 && G > 1 && (Q & 15) == 0) ? Q + 4 : Q; }

template <typename R> __host__ __device__ inline size_t tab_lds_bytes(int G, int Q, bool shared_x = false)
{
    size_t n = (size_t)G * tab_row_stride<R>(G, Q);
    // (+ 16 bytes behind xs and ys each: tab_stage may shift a table by up to three elements, see there)
    return 2 * sizeof(double) * (size_t)G + 3 * sizeof(double) * (size_t)Q + sizeof(R) * (n + (shared_x ? (size_t)Q : n)) +
           2 * sizeof(R) * (size_t)G + sizeof(int) * ((size_t)G + 2) + 16 + 32;
}

template <typename R> __device__ __forceinline__ TabLds<R> tab_carve(unsigned char *p, int G, int Q, bool shared_x = false)
{
    TabLds<R> t;
    size_t n = (size_t)G * tab_row_stride<R>(G, Q);
    t.xstride = shared_x ? 0 : Q;
    t.xfirst = reinterpret_cast<double *>(p); p += sizeof(double) * (size_t)G;
    t.xlast = reinterpret_cast<double *>(p); p += sizeof(double) * (size_t)G;
    t.spl = reinterpret_cast<double *>(p); p += 3 * sizeof(double) * (size_t)Q;
    t.xs = reinterpret_cast<R *>(p); p += sizeof(R) * (shared_x ? (size_t)Q : n) + 16;
    t.ys = reinterpret_cast<R *>(p); p += sizeof(R) * n + 16;
    t.yfirst = reinterpret_cast<R *>(p); p += sizeof(R) * (size_t)G;
    t.ylast = reinterpret_cast<R *>(p); p += sizeof(R) * (size_t)G;
    t.rn = reinterpret_cast<int *>(p); p += sizeof(int) * (size_t)G;
    t.any = reinterpret_cast<int *>(p);
    return t;
}

template <typename R>
__device__ __forceinline__ CellTab<R> tab_stage(const TabLds<R> &t, const R *__restrict__ xq, int64_t xq_row_stride,
                                                const R *__restrict__ yq, int G, int Q)
{
    const int n = G * Q;
    if (threadIdx.x == 0) *t.any = 0;
    bool has_nan = false;
    if (t.xstride == 0) {  // one shared abscissa row (xq_row_stride == 0)
        for (int i = threadIdx.x; i < Q; i += blockDim.x) { const R xv = xq[i]; t.xs[i] = xv; has_nan |= xv != xv; }
        {
            constexpr int NB = 8;  // batched like the straight copy below
            for (int i0 = threadIdx.x; i0 < n; i0 += NB * blockDim.x) {
                R yv[NB];
#pragma unroll
                for (int k = 0; k < NB; k++) yv[k] = yq[min(i0 + k * (int)blockDim.x, n - 1)];
#pragma unroll
                for (int k = 0; k < NB; k++) {
                    const int i = i0 + k * (int)blockDim.x;
                    if (i < n) { t.ys[i] = yv[k]; has_nan |= yv[k] != yv[k]; }
                }
            }
        }
        CellTab<R> ct;
        ct.xs = t.xs; ct.ys = t.ys; ct.rn = t.rn; ct.xfirst = t.xfirst; ct.xlast = t.xlast;
        ct.yfirst = t.yfirst; ct.ylast = t.ylast; ct.G = G; ct.Q = Q; ct.xstride = 0;
        if (__syncthreads_or(has_nan ? 1 : 0)) { ct.any_valid = -1; return ct; }
        for (int g = threadIdx.x; g < G; g += blockDim.x) {
            const R *yr = t.ys + (size_t)g * Q;
            t.rn[g] = Q; t.xfirst[g] = (double)t.xs[0]; t.xlast[g] = (double)t.xs[Q - 1]; t.yfirst[g] = yr[0]; t.ylast[g] = yr[Q - 1];
        }
        __syncthreads();
        ct.any_valid = 1;
        return ct;
    }
    const int RS = tab_row_stride<R>(G, Q);  // row stride in LDS: Q, or Q + 4 (tab_row_stride)
    R *xs_ = t.xs, *ys_ = t.ys;  // where the tables end up (shifted by up to three elements in the unaligned 16-byte staging below)
    const bool vec16 = xq_row_stride == Q && (n & 3) == 0 &&
                       ((reinterpret_cast<uintptr_t>(xq) | reinterpret_cast<uintptr_t>(yq) | reinterpret_cast<uintptr_t>(t.xs) |
                         reinterpret_cast<uintptr_t>(t.ys)) & 15) == 0;
    if (sizeof(R) == 4 && vec16) {  // plain [G][Q] float tables, 16-byte aligned on both sides: four values per load and store
        const int n4 = n >> 2;
        const uint4 *x4 = reinterpret_cast<const uint4 *>(xq), *y4 = reinterpret_cast<const uint4 *>(yq);
        uint4 *dx = reinterpret_cast<uint4 *>(t.xs), *dy = reinterpret_cast<uint4 *>(t.ys);
        auto nan4 = [](const uint4 &v) {  // any NaN among four float bit patterns
            auto isn = [](uint32_t b) { return (b & 0x7FFFFFFFu) > 0x7F800000u; };
            return isn(v.x) || isn(v.y) || isn(v.z) || isn(v.w);
        };
        constexpr int NB4 = 3;  // (a 365 x 20 table is 1825 vectors: three per thread of a 768-thread block, all in flight)
        for (int i0 = threadIdx.x; i0 < n4; i0 += NB4 * blockDim.x) {
            uint4 xv[NB4], yv[NB4];
#pragma unroll
            for (int k = 0; k < NB4; k++) { const int i = min(i0 + k * (int)blockDim.x, n4 - 1); xv[k] = x4[i]; yv[k] = y4[i]; }
#pragma unroll
            for (int k = 0; k < NB4; k++) {
                const int i = i0 + k * (int)blockDim.x;
                if (i < n4) {
                    const int d = RS == Q ? i : i + i / (Q >> 2);  // (padded rows: Q is a multiple of 16, a vector lies inside one row)
                    dx[d] = xv[k]; dy[d] = yv[k]; has_nan |= nan4(xv[k]) || nan4(yv[k]);
                }
            }
        }
    } else if (sizeof(R) == 4 && xq_row_stride == Q && RS == Q && n >= 16 &&
               ((reinterpret_cast<uintptr_t>(xq) | reinterpret_cast<uintptr_t>(yq)) & 3) == 0 &&
               ((reinterpret_cast<uintptr_t>(xq) ^ reinterpret_cast<uintptr_t>(yq)) & 15) == 0) {
        // plain [G][Q] float tables that are NOT 16-byte aligned (15, 19, 21 ... nodes: a cell's table starts anywhere): the LDS
        // copies are shifted by up to three elements so that their addresses are congruent to the global ones modulo 16 — the
        // body then moves in 16-byte pieces like the aligned case, the up to three elements in front of the first whole piece
        // and behind the last one go one by one (the scalar copy of such a table cost 0.5 ms per 64 800 cells)
        xs_ = t.xs + (((reinterpret_cast<uintptr_t>(xq) - reinterpret_cast<uintptr_t>(t.xs)) & 15) >> 2);
        ys_ = t.ys + (((reinterpret_cast<uintptr_t>(yq) - reinterpret_cast<uintptr_t>(t.ys)) & 15) >> 2);
        const int head = (int)(((16 - (reinterpret_cast<uintptr_t>(xq) & 15)) & 15) >> 2), n4 = (n - head) >> 2, tail0 = head + 4 * n4;
        if ((int)threadIdx.x < head) {
            const R xv = xq[threadIdx.x], yv = yq[threadIdx.x];
            xs_[threadIdx.x] = xv; ys_[threadIdx.x] = yv; has_nan |= (xv != xv) || (yv != yv);
        }
        if ((int)threadIdx.x >= 4 && (int)threadIdx.x - 4 < n - tail0) {  // (lanes 4..6: present in every block size)
            const int i = tail0 + (int)threadIdx.x - 4;
            const R xv = xq[i], yv = yq[i];
            xs_[i] = xv; ys_[i] = yv; has_nan |= (xv != xv) || (yv != yv);
        }
        const uint4 *x4 = reinterpret_cast<const uint4 *>(xq + head), *y4 = reinterpret_cast<const uint4 *>(yq + head);
        uint4 *dx = reinterpret_cast<uint4 *>(xs_ + head), *dy = reinterpret_cast<uint4 *>(ys_ + head);
        auto nan4 = [](const uint4 &v) {
            auto isn = [](uint32_t b) { return (b & 0x7FFFFFFFu) > 0x7F800000u; };
            return isn(v.x) || isn(v.y) || isn(v.z) || isn(v.w);
        };
        constexpr int NB4 = 3;
        for (int i0 = threadIdx.x; i0 < n4; i0 += NB4 * blockDim.x) {
            uint4 xv[NB4], yv[NB4];
#pragma unroll
            for (int k = 0; k < NB4; k++) { const int i = min(i0 + k * (int)blockDim.x, n4 - 1); xv[k] = x4[i]; yv[k] = y4[i]; }
#pragma unroll
            for (int k = 0; k < NB4; k++) {
                const int i = i0 + k * (int)blockDim.x;
                if (i < n4) { dx[i] = xv[k]; dy[i] = yv[k]; has_nan |= nan4(xv[k]) || nan4(yv[k]); }
            }
        }
    } else if (xq_row_stride == Q) {  // both tables are plain [G][Q] arrays: straight copy, no index arithmetic
        // NB elements of each table are requested before the first is stored: one memory latency per batch
        constexpr int NB = 8;
        for (int i0 = threadIdx.x; i0 < n; i0 += NB * blockDim.x) {
            R xv[NB], yv[NB];
#pragma unroll
            for (int k = 0; k < NB; k++) { const int i = min(i0 + k * (int)blockDim.x, n - 1); xv[k] = xq[i]; yv[k] = yq[i]; }
#pragma unroll
            for (int k = 0; k < NB; k++) {
                const int i = i0 + k * (int)blockDim.x;
                if (i < n) { const int d = RS == Q ? i : i + (RS - Q) * (i / Q); t.xs[d] = xv[k]; t.ys[d] = yv[k]; has_nan |= (xv[k] != xv[k]) || (yv[k] != yv[k]); }
            }
        }
    } else {
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            int g = i / Q, k = i - g * Q;
            const R xv = xq[(size_t)g * xq_row_stride + k], yv = yq[i];
            t.xs[(size_t)g * RS + k] = xv;
            t.ys[(size_t)g * RS + k] = yv;
            has_nan |= (xv != xv) || (yv != yv);
        }
    }
    if (!__syncthreads_or(has_nan ? 1 : 0)) {
        // the usual case, no NaN anywhere in the cell's table: rows are already compact
        for (int g = threadIdx.x; g < G; g += blockDim.x) {
            const R *xr = xs_ + (size_t)g * RS, *yr = ys_ + (size_t)g * RS;
            t.rn[g] = Q; t.xfirst[g] = (double)xr[0]; t.xlast[g] = (double)xr[Q - 1]; t.yfirst[g] = yr[0]; t.ylast[g] = yr[Q - 1];
        }
        if (threadIdx.x == 0) *t.any = 1;
        __syncthreads();
        CellTab<R> ct;
        ct.xs = xs_; ct.ys = ys_; ct.rn = t.rn; ct.xfirst = t.xfirst; ct.xlast = t.xlast;
        ct.yfirst = t.yfirst; ct.ylast = t.ylast; ct.G = G; ct.Q = Q; ct.any_valid = 1; ct.xstride = RS; ct.ystride = RS;
        ct.dense = true;
        return ct;
    }
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        R *xr = xs_ + (size_t)g * RS, *yr = ys_ + (size_t)g * RS;
        int m = 0;
        bool fx = false, fy = false;
        double xf = __builtin_nan(""), xl = __builtin_nan("");
        R yf = nan_of<R>(), yl = nan_of<R>();
        for (int k = 0; k < Q; k++) {
            R xv = xr[k], yv = yr[k];
            if (xv == xv) { if (!fx) { xf = (double)xv; fx = true; } xl = (double)xv; }
            if (yv == yv) { if (!fy) { yf = yv; fy = true; } yl = yv; }
            if (xv == xv && yv == yv) { xr[m] = xv; yr[m] = yv; m++; }
        }
        t.rn[g] = m; t.xfirst[g] = xf; t.xlast[g] = xl; t.yfirst[g] = yf; t.ylast[g] = yl;
        if (m) atomicOr(t.any, 1);
    }
    __syncthreads();
    CellTab<R> ct;
    ct.xs = xs_; ct.ys = ys_; ct.rn = t.rn; ct.xfirst = t.xfirst; ct.xlast = t.xlast;
    ct.yfirst = t.yfirst; ct.ylast = t.ylast; ct.G = G; ct.Q = Q; ct.any_valid = *t.any; ct.xstride = RS; ct.ystride = RS;
    return ct;
}

// interp = cubic on a one-group table: the spline of the compacted row (after tab_stage; ends on a barrier)
template <typename R>
__device__ __forceinline__ void tab_spline(CellTab<R> &ct, const TabLds<R> &t, int interp)
{
    if (interp != 2 || ct.G != 1) return;
    if (threadIdx.x == 0) spline_not_a_knot<R>(ct.xs, ct.ys, ct.rn[0], t.spl, t.spl + ct.Q);
    __syncthreads();
    ct.spl = t.spl;
}

// ---------------------------------------------------------------------------------------------
// K2: qm_adjust (_adjustment.py:594-676) and the mapping stage of dqm_adjust (:748-773).
// One block per cell; the cell's (hist_q, af) table is staged in LDS once, then the block streams the
// series.  scaling/trend non-null -> DQM: x = (sim (+|*) scaling[g]) detrended by trend[t]
// (detrending.py:102-120), mapped, retrended.  trend is cell-major contiguous [cell][time].
// ---------------------------------------------------------------------------------------------
// TAB_LDS: the compacted table lives in LDS (the normal case); otherwise (tables larger than the LDS, e.g.
// float64 with 50 quantiles x 366 groups) each block compacts it into its own slice of a global scratch
// buffer (L2-resident) and the block walks the cells grid-stride.
// GROUPED (more than one group: 2-D nearest neighbour) and DQM (scaling + trend supplied) are template parameters so
// that the per-element loop carries no mode branches.  The series are cell-major contiguous rows (time stride 1).
// DQM without a trend array: the per-group regression coefficients (x_mean, y_mean, slope) of k_group_trend, [cell][G][3]
// float64, and the time axis; the kernel evaluates the trend of an element itself (same expression, same value)
struct TrendCoef {
    const double *coef = nullptr, *tnum = nullptr;
    int affine = 0;
    double t0 = 0.0, dt = 0.0;
};

template <typename R, bool TAB_LDS, bool GROUPED, bool DQM>
__global__ void k_qm_adjust(const R *__restrict__ sim, int64_t T, int64_t C, int64_t sc,
                            const R *__restrict__ af, const R *__restrict__ hist_q, int G, int Q,
                            const int *__restrict__ gidx, int interp, int extrap, int kind, R *__restrict__ scen,
                            int64_t osc, int *__restrict__ node, const R *__restrict__ scaling,
                            const R *__restrict__ trend, unsigned char *__restrict__ gscratch, size_t gstride, int tkind,
                            int skip0, TrendCoef tc = TrendCoef())
{
    // tkind: the kind of the detrending (PolyDetrend has one of its own, detrending.py:102-120); skip0: mult_skip_zeros —
    // where the trend is 0 the value is kept instead of divided (:112-113)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *tabmem = TAB_LDS ? smem : gscratch + (size_t)blockIdx.x * gstride;
    TabLds<R> tl = tab_carve<R>(tabmem, G, Q);
    double *cl = reinterpret_cast<double *>(smem + (TAB_LDS ? ((tab_lds_bytes<R>(G, Q) + 15) & ~(size_t)15) : 0));  // [G][3] when tc.coef
    const int Ti = (int)T;
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        if (DQM && tc.coef) {  // (tab_stage ends on a barrier)
            for (int i = threadIdx.x; i < 3 * G; i += blockDim.x) cl[i] = tc.coef[(size_t)c * 3 * G + i];
            // the cell's scaling factors ride behind the coefficients: read from LDS per element instead of gathered from L2
            float *sl = reinterpret_cast<float *>(cl + 3 * G);
            for (int i = threadIdx.x; i < G; i += blockDim.x) sl[i] = (float)scaling[(size_t)c * G + i];
        }
        CellTab<R> ct = tab_stage<R>(tl, hist_q + (size_t)c * G * Q, Q, af + (size_t)c * G * Q, G, Q);
        if constexpr (!GROUPED) tab_spline<R>(ct, tl, interp);
        const R *x = sim + c * sc;
        R *o = scen + c * osc;
        const R *scal = DQM ? ((tc.coef && sizeof(R) == 4) ? reinterpret_cast<const R *>(cl + 3 * G) : scaling + (size_t)c * G) : nullptr;
        const R *trc = (DQM && trend) ? trend + (size_t)c * T : nullptr;
        int *ndc = node ? node + (size_t)c * T : nullptr;
        // EU elements per thread and round: their table searches are independent chains of LDS reads
#ifndef XS_ADJ_EU
#define XS_ADJ_EU 4
#endif
        constexpr int EU = XS_ADJ_EU;
        for (int t0 = threadIdx.x; t0 < Ti; t0 += EU * blockDim.x) {
            R v[EU], tr[EU], a[EU];
            int g0[EU], nd[EU];
#pragma unroll
            for (int e = 0; e < EU; e++) {
                const int t = min(t0 + e * (int)blockDim.x, Ti - 1);
                v[e] = x[t];
                g0[e] = (GROUPED || DQM) ? gidx[t] : 0;
                tr[e] = (R)0;
                if constexpr (DQM) {
                    R s = g0[e] >= 0 ? scal[g0[e]] : nan_of<R>();
                    v[e] = apply_corr<R>(v[e], s, kind);
                    if (trc) tr[e] = trc[t];
                    else {
                        const int g = g0[e];
                        const double xt = tc.affine ? tc.t0 + tc.dt * (double)t : tc.tnum[t];
                        tr[e] = g >= 0 ? (R)(cl[3 * g + 1] + cl[3 * g + 2] * (xt - cl[3 * g])) : nan_of<R>();
                    }
                    if (!(skip0 && tr[e] == (R)0)) v[e] = apply_corr<R>(v[e], invert<R>(tr[e], tkind), tkind);
                }
            }
#pragma unroll
            for (int e = 0; e < EU; e++) {
                if constexpr (GROUPED) {
                    if (g0[e] < 0) { nd[e] = -1; a[e] = nan_of<R>(); }
                    else if (ct.dense) a[e] = interp2d_nearest_dense<R>(v[e], g0[e], ct, extrap, nd[e]);
                    else a[e] = interp2d_nearest<R, R>(v[e], g0[e], ct, extrap, nd[e]);
                } else {
                    a[e] = interp1d_eval<R>((double)v[e], false, ct.xs, ct.ys, ct.rn[0], ct.yfirst[0], ct.ylast[0], interp, extrap, nd[e], ct.spl);
                }
            }
#pragma unroll
            for (int e = 0; e < EU; e++) {
                const int t = t0 + e * (int)blockDim.x;
                if (t < Ti) {
                    R y = kind ? apply_corr<R>(v[e], a[e], kind) : a[e];
                    if constexpr (DQM) y = apply_corr<R>(y, tr[e], tkind);
                    o[t] = y;
                    if (ndc) ndc[t] = nd[e];
                }
            }
        }
        __syncthreads();  // everyone is done with the table before the next cell overwrites it
    }
}

// ---------------------------------------------------------------------------------------------
// K3: PolyDetrend(degree 0/1) trend per group (detrending.py:196-208) of the (optionally scaled) series.
// One block per cell: series in LDS -> NaN-skipping centred window means (partial at the ends) ->
// per-group least squares in float64 against time_num -> trend written cell-major.  The window means
// are parked in the trend buffer itself between the two phases.
// dynamic LDS: [coef: ncoef*G doubles][rcp: W+1 doubles][xs: T * sizeof(R)]
// degree 0 / 1 (ncoef = 3): closed-form mean / slope.  degree 2..POLY_MAX_DEG or preserve_mean (ncoef = degree + 4): least
// squares through the normal equations in the variable u = (x - x_mid) / x_halfrange of the group, which keeps them well
// conditioned (xarray's polyfit scales the columns of the Vandermonde matrix in x = ns since 1970 and goes through
// LAPACK's SVD; the fitted values agree to ~1e-9 relative, the conditioning error of that route), Horner in u.
// preserve_mean (detrending.py:204-205): the group's trend has its own mean over the group's time steps removed ('+') or
// divided out ('*'), in float64 before the cast.
// ---------------------------------------------------------------------------------------------
constexpr int POLY_MAX_DEG = 6;
constexpr int TREND_CH = 25;  // time steps per thread of k_group_trend's in-register window means (the host picks the block size from it)
constexpr int TREND_SCRATCH_D = 160;  // doubles of block scratch behind k_group_trend's coefficients (5 partial sums x 32 half-waves)

// least squares of degree d from the moments S_k = sum u^k (k <= 2d), B_k = sum u^k y (k <= d): c[0..d] ascending, NaN if singular
__device__ __forceinline__ void poly_solve(const double *S, const double *B, int d, double *c)
{
    double A[POLY_MAX_DEG + 1][POLY_MAX_DEG + 2];
    const int n = d + 1;
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) A[i][j] = S[i + j];
        A[i][n] = B[i];
    }
    bool ok = true;
    for (int k = 0; k < n; k++) {
        int piv = k;
        for (int i = k + 1; i < n; i++) if (fabs(A[i][k]) > fabs(A[piv][k])) piv = i;
        if (!(fabs(A[piv][k]) > 1e-300)) { ok = false; break; }
        if (piv != k) for (int j = k; j <= n; j++) { const double t = A[k][j]; A[k][j] = A[piv][j]; A[piv][j] = t; }
        for (int i = k + 1; i < n; i++) {
            const double f = A[i][k] / A[k][k];
            for (int j = k; j <= n; j++) A[i][j] -= f * A[k][j];
        }
    }
    for (int k = n - 1; k >= 0; k--) {
        double v = A[k][n];
        for (int j = k + 1; j < n; j++) v -= A[k][j] * c[j];
        c[k] = ok ? v / A[k][k] : __builtin_nan("");
    }
}

// M_LDS: the window means live in a second LDS array (T * sizeof(R) more) instead of the trend buffer — the
// normal case for float32; float64 series that do not fit keep them in global memory.
// MT64: the packed member table has 32 words per group (groups of 33 .. 64 members: 60 years of a day of the year) instead of 16 —
// an instantiation of its own, so that its 32 more registers do not cost the usual one (<= 32 members) its third block per CU
// XG (series too long for the LDS: hourly data): nothing of size T lives on chip — the (scaled) series is read from global
// memory wherever the LDS copy would be (its window sums re-derive the scaling per element), the window means go to the trend
// buffer; requires the trend buffer (trend_cm) even when only the coefficients are asked for.
template <typename R, bool M_LDS, bool MT64 = false, bool XG = false>
__global__ void k_group_trend(const R *__restrict__ sim, int64_t T, int64_t st, int64_t sc,
                              const R *__restrict__ scaling, int G, int kind, const int *__restrict__ gidx, int W,
                              const int *__restrict__ mem_ptr, const int *__restrict__ mem_idx,
                              const double *__restrict__ tnum, int degree, R *__restrict__ trend_cm, int lane_per_group,
                              int preserve_mean, int min_periods, int tn_affine, double tn0, double tnd,
                              const int *__restrict__ mem_t16, double *__restrict__ coef_out)
{
    // coef_out ([cell][G][3], degree <= 1 without preserve_mean): leave the regression coefficients instead of the trend —
    // the DQM adjust kernel evaluates it per element (no [cell][time] array written and read back)
    // mem_t16: the plan's packed member table ([16][G] words of two 16-bit time indexes: groups of <= 32 members; MT64: [32][G],
    // <= 64 members) or null
    // tn_affine: the time axis is tnum[t] == tn0 + tnd * t exactly for every t (host-checked: daily steps in ns since an
    // epoch are): the coordinate is computed instead of gathered from L2 in the regression and in the trend evaluation
    auto TN = [&](int64_t t) { return tn_affine ? tn0 + tnd * (double)t : tnum[t]; };
    // min_periods > 0: rolling-mean mode (RollingMeanDetrend, detrending.py:342-356: da.rolling(center=True, min_periods)
    // .mean()): the window means themselves are the result, NaN where fewer than min_periods values are in the window
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const bool general = degree >= 2 || preserve_mean;
    const int ncoef = general ? degree + 4 : 3;
    double *coef = reinterpret_cast<double *>(smem);
    double *part = coef + (size_t)ncoef * G;  // [TREND_SCRATCH_D] partial sums of the half-waves that share a group
    // (the scratch exists only when half-waves share groups — 2 G <= half-waves per block, the host sizes the LDS by the same rule —:
    //  1.3 KB more would cost the day-of-year case its third block per CU)
    double *rcp = part + (2 * G <= (int)(blockDim.x >> 5) ? TREND_SCRATCH_D : 0);  // [W+1] correctly rounded reciprocals of the possible window counts
    R *xs_l = reinterpret_cast<R *>(rcp + W + 1);
    const int64_t c = blockIdx.x;
    const R *x = sim + c * sc;
    R *tr_out = trend_cm + (size_t)c * T;
    for (int i = threadIdx.x; i <= W; i += blockDim.x) rcp[i] = i ? 1.0 / (double)i : 0.0;
    R *m = M_LDS ? xs_l + T : tr_out;
    // the cell's scaling factors wait in LDS; XG: behind the coefficient area (they are needed for as long as the series is read)
    R *scl_x = reinterpret_cast<R *>(rcp + W + 1);
    // the scaled series: the LDS copy, or (XG) straight from global memory
    struct XsView {
        R *l; const R *x; const R *scl; const int *gidx; int64_t st; int kind;
        __device__ __forceinline__ R operator[](int64_t t) const
        {
            if constexpr (!XG) return l[t];
            else {
                R w = x[t * st];
                if (scl) { const int g = gidx[t]; w = apply_corr<R>(w, g >= 0 ? scl[g] : nan_of<R>(), kind); }
                return w;
            }
        }
    };
    const XsView xs{xs_l, x, (XG && scaling) ? scl_x : nullptr, gidx, st, kind};
    // the cell's scaling factors wait in LDS (in the coefficient area, which is written much later): the load below then has one
    // global round trip per batch — value and group — instead of a dependent second one for the factor of that group
    R *scl = XG ? scl_x : reinterpret_cast<R *>(coef);
    if (scaling) {
        for (int i = threadIdx.x; i < G; i += blockDim.x) scl[i] = scaling[(size_t)c * G + i];
        __syncthreads();
    }
    if constexpr (!XG) {
        constexpr int NB = 8;  // NB series values (and their groups) are requested before the first is consumed
        for (int64_t t0 = threadIdx.x; t0 < T; t0 += (int64_t)NB * blockDim.x) {
            R v[NB];
            int gg[NB];
#pragma unroll
            for (int k = 0; k < NB; k++) {
                const int64_t t = t0 + (int64_t)k * blockDim.x, tc = t < T ? t : T - 1;
                v[k] = x[tc * st];
                gg[k] = scaling ? gidx[tc] : 0;
            }
#pragma unroll
            for (int k = 0; k < NB; k++) {
                const int64_t t = t0 + (int64_t)k * blockDim.x;
                if (t < T) {
                    R w = v[k];
                    if (scaling) w = apply_corr<R>(w, gg[k] >= 0 ? scl[gg[k]] : nan_of<R>(), kind);
                    xs_l[t] = w;
                }
            }
        }
    }
    __syncthreads();
    const int h = W / 2;
    // Window means.  Every thread owns a contiguous chunk of time steps and slides the window: one full window sum,
    // then one value in / one value out per step (float64 accumulation; exact for float32 data whose values within
    // a window span < 2^24, so it equals a fresh sum); s / n through a reciprocal table + one FMA residual
    // correction (Markstein) instead of the long IEEE division sequence.  When the series fits CH values per
    // thread the means are held in registers and written back over the series in place (no second array, no
    // global round trip); longer series park them in the trend buffer.
    constexpr int CH = TREND_CH;  // (odd: the threads of a wave read their chunks CH dwords apart — 24 put eight lanes on every LDS bank)
    const bool inplace = !XG && !M_LDS && W > 1 && T <= (int64_t)CH * blockDim.x;
    auto mean_of = [&](double s, int n) {
        double mv = __builtin_nan("");
        if (n && n >= min_periods) {
            const double dn = (double)n, rc = rcp[n];
            double q0 = s * rc;
            double r0 = __builtin_fma(-q0, dn, s);
            mv = __builtin_fma(r0, rc, q0);
        }
        return (R)mv;
    };
    // `big`: the largest magnitude that went through the running sum since it was last built from scratch.  The sum
    // carries a rounding error of about eps * big, so once big exceeds 1e6 * |s| (huge dynamic range: 1e-20 drizzle
    // right after a 1e2 storm left the window) the window is summed afresh.
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
    if (W == 1) {
        if constexpr (XG) {  // the "window mean" is the series itself: written out where the regression will read it
            for (int64_t t = threadIdx.x; t < T; t += blockDim.x) tr_out[t] = xs[t];
            m = tr_out;
        } else m = xs_l;
    } else if (inplace) {
        const int64_t t0 = (int64_t)threadIdx.x * CH;
        R mreg[CH];
        double s = 0, big = 0; int n = 0;
#pragma unroll
        for (int k = 0; k < CH; k++) {
            mreg[k] = nan_of<R>();
            if (t0 + k < T) { slide(t0 + k, k == 0, s, n, big); mreg[k] = mean_of(s, n); }
        }
        __syncthreads();  // every window has been read: the series can be overwritten by its window means
#pragma unroll
        for (int k = 0; k < CH; k++) if (t0 + k < T) xs_l[t0 + k] = mreg[k];
        m = xs_l;
    } else {
        const int64_t chunk = (T + blockDim.x - 1) / blockDim.x;
        const int64_t t0 = (int64_t)threadIdx.x * chunk, t1 = t0 + chunk < T ? t0 + chunk : T;
        double s = 0, big = 0; int n = 0;
        for (int64_t t = t0; t < t1; t++) { slide(t, t == t0, s, n, big); m[t] = mean_of(s, n); }
    }
    __syncthreads();
    if (min_periods > 0) {
        if (W == 1) { if (m != tr_out) for (int64_t t = threadIdx.x; t < T; t += blockDim.x) tr_out[t] = xs[t]; }  // a window of one step
        else if (m != tr_out) { for (int64_t t = threadIdx.x; t < T; t += blockDim.x) tr_out[t] = m[t]; }
        return;
    }
    if (general) {
        // general degree / preserve_mean: one lane per group, serial over its members (moments, solve, mean of the fit)
        for (int g = threadIdx.x; g < G; g += blockDim.x) {
            const int p0 = mem_ptr[g], p1 = mem_ptr[g + 1];
            double *cg = coef + (size_t)ncoef * g;
            double S[2 * POLY_MAX_DEG + 1], B[POLY_MAX_DEG + 1], cc[POLY_MAX_DEG + 1];
            for (int k = 0; k <= 2 * degree; k++) S[k] = 0.0;
            for (int k = 0; k <= degree; k++) B[k] = 0.0;
            const double xa = p1 > p0 ? TN(mem_idx[p0]) : 0.0, xb = p1 > p0 ? TN(mem_idx[p1 - 1]) : 0.0;
            const double xm = 0.5 * (xa + xb), hs = xb > xa ? 0.5 * (xb - xa) : 1.0;
            int n = 0;
            for (int p = p0; p < p1; p++) {
                const int t = mem_idx[p];
                const R v = m[t];
                if (v == v) {
                    const double u = (TN(t) - xm) / hs;
                    double pw = 1.0;
                    for (int k = 0; k <= 2 * degree; k++) { S[k] += pw; if (k <= degree) B[k] += pw * (double)v; pw *= u; }
                    n++;
                }
            }
            if (n > degree) poly_solve(S, B, degree, cc);
            else for (int k = 0; k <= degree; k++) cc[k] = __builtin_nan("");
            double adj = kind == '+' ? 0.0 : 1.0;
            if (preserve_mean) {
                double sm = 0.0;
                for (int p = p0; p < p1; p++) {
                    const double u = (TN(mem_idx[p]) - xm) / hs;
                    double y = cc[degree];
                    for (int k = degree - 1; k >= 0; k--) y = y * u + cc[k];
                    sm += y;
                }
                adj = p1 > p0 ? sm / (double)(p1 - p0) : __builtin_nan("");
            }
            cg[0] = xm; cg[1] = hs; cg[2] = adj;
            for (int k = 0; k <= degree; k++) cg[3 + k] = cc[k];
        }
        __syncthreads();
        for (int64_t t = threadIdx.x; t < T; t += blockDim.x) {
            const int g = gidx[t];
            R out = nan_of<R>();
            if (g >= 0) {
                const double *cg = coef + (size_t)ncoef * g;
                const double u = (TN(t) - cg[0]) / cg[1];
                double y = cg[3 + degree];
                for (int k = degree - 1; k >= 0; k--) y = y * u + cg[3 + k];
                if (preserve_mean) y = kind == '+' ? y - cg[2] : y / cg[2];
                out = (R)y;
            }
            tr_out[t] = out;
        }
        return;
    }
    // per-group least squares.  Many short groups (day-of-year: 365 groups of ~30 members): one LANE per group, a
    // serial loop over its members and no cross-lane reduction at all.
    if (lane_per_group && mem_t16 && tn_affine) {
        // groups of <= 32 (mem_tw = 16 words per group) or <= 64 members (32 words: 60 years of a day of the year) on an affine
        // time axis: the member indexes arrive as coalesced words and nothing else is gathered from global memory — both passes
        // of the regression read the window means from LDS, in member order
        auto regress = [&](auto nwc) {
            constexpr int NW = decltype(nwc)::value;
            for (int g = threadIdx.x; g < G; g += blockDim.x) {
                uint32_t w[NW];
#pragma unroll
                for (int u = 0; u < NW; u++) w[u] = (uint32_t)mem_t16[(size_t)u * G + g];
                double sx = 0, sy = 0;
                int n = 0;
#pragma unroll
                for (int y = 0; y < 2 * NW; y++) {
                    const uint32_t t = (w[y / 2] >> (16 * (y & 1))) & 0xFFFFu;
                    const R v = t != 0xFFFFu ? m[t] : nan_of<R>();
                    if (v == v) { sx += TN(t); sy += (double)v; n++; }
                }
                const double xb = n ? sx / n : 0.0, yb = n ? sy / n : __builtin_nan("");
                double slope = 0.0;
                if (degree >= 1 && n >= 2) {
                    double sxx = 0, sxy = 0;
#pragma unroll
                    for (int y = 0; y < 2 * NW; y++) {
                        const uint32_t t = (w[y / 2] >> (16 * (y & 1))) & 0xFFFFu;
                        const R v = t != 0xFFFFu ? m[t] : nan_of<R>();
                        if (v == v) { const double dx = TN(t) - xb; sxx += dx * dx; sxy += dx * ((double)v - yb); }
                    }
                    slope = sxx > 0 ? sxy / sxx : 0.0;
                }
                coef[3 * g] = xb; coef[3 * g + 1] = yb; coef[3 * g + 2] = slope;
            }
        };
        regress(std::integral_constant<int, MT64 ? 32 : 16>{});
    } else if (lane_per_group) {
        for (int g = threadIdx.x; g < G; g += blockDim.x) {
            const int p0 = mem_ptr[g], p1 = mem_ptr[g + 1];
            double sx = 0, sy = 0;
            int n = 0;
            // the member list, the window means and the time axis are gathers (global / L2 latency): NB members are
            // fetched before the first is consumed, then accumulated in member order (the sums stay deterministic)
            constexpr int NB = 8;
            for (int pb = p0; pb < p1; pb += NB) {
                int tt[NB];
                R vv[NB];
                double xx[NB];
#pragma unroll
                for (int k = 0; k < NB; k++) tt[k] = mem_idx[min(pb + k, p1 - 1)];
#pragma unroll
                for (int k = 0; k < NB; k++) { vv[k] = m[tt[k]]; xx[k] = TN(tt[k]); }
#pragma unroll
                for (int k = 0; k < NB; k++)
                    if (pb + k < p1 && vv[k] == vv[k]) { sx += xx[k]; sy += (double)vv[k]; n++; }
            }
            const double xb = n ? sx / n : 0.0, yb = n ? sy / n : __builtin_nan("");
            double slope = 0.0;
            if (degree >= 1 && n >= 2) {
                double sxx = 0, sxy = 0;
                for (int pb = p0; pb < p1; pb += NB) {
                    int tt[NB];
                    R vv[NB];
                    double xx[NB];
#pragma unroll
                    for (int k = 0; k < NB; k++) tt[k] = mem_idx[min(pb + k, p1 - 1)];
#pragma unroll
                    for (int k = 0; k < NB; k++) { vv[k] = m[tt[k]]; xx[k] = TN(tt[k]); }
#pragma unroll
                    for (int k = 0; k < NB; k++)
                        if (pb + k < p1 && vv[k] == vv[k]) { const double dx = xx[k] - xb; sxx += dx * dx; sxy += dx * ((double)vv[k] - yb); }
                }
                slope = sxx > 0 ? sxy / sxx : 0.0;
            }
            coef[3 * g] = xb; coef[3 * g + 1] = yb; coef[3 * g + 2] = slope;
        }
    } else {
        // few long groups: one half-wave (32 lanes) per group, members loaded in parallel, sums by xor-shuffles
        const int hw = threadIdx.x >> 5, nhw = blockDim.x >> 5, hl = threadIdx.x & 31;
        auto hsum = [](double v) {
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 32);
            return v;
        };
        const int nsplit = 2 * G <= nhw ? nhw / G : 1;  // half-waves per group
        if (nsplit > 1) {
            // fewer groups than half-waves (group="time": one group of T members): the half-waves of a group take its members
            // 32 * nsplit apart and add their partial sums in a fixed order — one half-wave alone walked all 10 950 members of
            // a 30-year series twice (4.2 ms per 16 200 cells, more than the training of the same cube)
            const int g = hw / nsplit, sp = hw - g * nsplit;
            const bool on = g < G;
            const int p0 = on ? mem_ptr[g] : 0, p1 = on ? mem_ptr[g + 1] : 0;
            double sx = 0, sy = 0, cn = 0;
            for (int p = p0 + sp * 32 + hl; p < p1; p += 32 * nsplit) {
                int t = mem_idx[p];
                R v = m[t];
                if (v == v) { sx += TN(t); sy += (double)v; cn += 1.0; }
            }
            sx = hsum(sx); sy = hsum(sy); cn = hsum(cn);
            if (hl == 0) { part[5 * hw] = sx; part[5 * hw + 1] = sy; part[5 * hw + 2] = cn; }
            __syncthreads();
            sx = sy = cn = 0;
            if (on) for (int k = 0; k < nsplit; k++) { sx += part[5 * (g * nsplit + k)]; sy += part[5 * (g * nsplit + k) + 1]; cn += part[5 * (g * nsplit + k) + 2]; }
            const int n = (int)cn;
            const double xb = n ? sx / n : 0.0, yb = n ? sy / n : __builtin_nan("");
            double sxx = 0, sxy = 0;
            if (degree >= 1 && n >= 2)
                for (int p = p0 + sp * 32 + hl; p < p1; p += 32 * nsplit) {
                    int t = mem_idx[p];
                    R v = m[t];
                    if (v == v) { double dx = TN(t) - xb; sxx += dx * dx; sxy += dx * ((double)v - yb); }
                }
            sxx = hsum(sxx); sxy = hsum(sxy);
            if (hl == 0) { part[5 * hw + 3] = sxx; part[5 * hw + 4] = sxy; }
            __syncthreads();
            if (on && sp == 0 && hl == 0) {
                double a = 0, b = 0;
                for (int k = 0; k < nsplit; k++) { a += part[5 * (g * nsplit + k) + 3]; b += part[5 * (g * nsplit + k) + 4]; }
                const double slope = (degree >= 1 && n >= 2 && a > 0) ? b / a : 0.0;
                coef[3 * g] = xb; coef[3 * g + 1] = yb; coef[3 * g + 2] = slope;
            }
        } else
        for (int g = hw; g < G; g += nhw) {
            const int p0 = mem_ptr[g], p1 = mem_ptr[g + 1];
            double sx = 0, sy = 0, cn = 0;
            for (int p = p0 + hl; p < p1; p += 32) {
                int t = mem_idx[p];
                R v = m[t];
                if (v == v) { sx += TN(t); sy += (double)v; cn += 1.0; }
            }
            sx = hsum(sx); sy = hsum(sy); cn = hsum(cn);
            const int n = (int)cn;
            const double xb = n ? sx / n : 0.0, yb = n ? sy / n : __builtin_nan("");
            double slope = 0.0;
            if (degree >= 1 && n >= 2) {
                double sxx = 0, sxy = 0;
                for (int p = p0 + hl; p < p1; p += 32) {
                    int t = mem_idx[p];
                    R v = m[t];
                    if (v == v) { double dx = TN(t) - xb; sxx += dx * dx; sxy += dx * ((double)v - yb); }
                }
                sxx = hsum(sxx); sxy = hsum(sxy);
                slope = sxx > 0 ? sxy / sxx : 0.0;
            }
            if (hl == 0) { coef[3 * g] = xb; coef[3 * g + 1] = yb; coef[3 * g + 2] = slope; }
        }
    }
    __syncthreads();
    if (coef_out) {
        for (int i = threadIdx.x; i < 3 * G; i += blockDim.x) coef_out[(size_t)c * 3 * G + i] = coef[i];
        return;
    }
    {
        constexpr int NB = 4;
        for (int64_t t0 = threadIdx.x; t0 < T; t0 += (int64_t)NB * blockDim.x) {
            int gg[NB];
            double xx[NB];
#pragma unroll
            for (int k = 0; k < NB; k++) { const int64_t t = t0 + (int64_t)k * blockDim.x, tc = t < T ? t : T - 1; gg[k] = gidx[tc]; xx[k] = TN(tc); }
#pragma unroll
            for (int k = 0; k < NB; k++) {
                const int64_t t = t0 + (int64_t)k * blockDim.x;
                const int g = gg[k];
                if (t < T) tr_out[t] = g >= 0 ? (R)(coef[3 * g + 1] + coef[3 * g + 2] * (xx[k] - coef[3 * g])) : nan_of<R>();
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K4: qdm_adjust (_adjustment.py:783-886): percent ranks of sim within its group (utils.rank, utils.py:573-638;
// average ties, NaN -> NaN, p = r/N rescaled with the extreme ranks) + interpolation of af on the quantile
// axis + apply.  af == nullptr -> ranks only (utils.rank through Grouper.apply).
// One block per cell; dynamic LDS: [table (if af; shared abscissa row)][gstat: 3*G doubles][xs: T * sizeof(R)]
// [gs: G * (sort_w + 1) keys]; cells whose factors hold NaN (and every cell when !TAB_LDS) use a full-layout table in the
// block's slice of gscratch.
// sort_w (0, 32 or 64): ranking rows of at most sort_w values (exact day-of-year groups: 30 members) are SORTED once —
// one row per sort_w lanes, bitonic network on the cross-lane path — and every element then finds its rank by two
// bisections in its row; sort_w == 0 ranks by counting over the row (any length).
// ---------------------------------------------------------------------------------------------
// XG (series too long for the LDS): the ranking series is read from global memory in place (cell-major input, st == 1: the
// host lays anything else out first); nothing of size T lives on chip.
template <typename R, bool TAB_LDS, bool XG = false>
__global__ void k_qdm_adjust(const R *__restrict__ sim, int64_t T, int64_t C, int64_t st, int64_t sc,
                             const R *__restrict__ af, const R *__restrict__ quantiles, int G, int Q,
                             const int *__restrict__ gidx, const int *__restrict__ grp_ptr,
                             const int *__restrict__ grp_idx, int interp, int extrap, int kind, R *__restrict__ scen,
                             double *__restrict__ sim_q, int64_t ost, int64_t osc, unsigned char *__restrict__ gscratch,
                             size_t gstride, const int *__restrict__ cell_list, const int *__restrict__ n_list_dev, int sort_w)
{
    using U = typename KeyOf<R>::U;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *p = smem;
    TabLds<R> tl, tlg;
    if (af) {
        tlg = tab_carve<R>(gscratch + (size_t)blockIdx.x * gstride, G, Q);
        if (TAB_LDS) {
            tl = tab_carve<R>(p, G, Q, true);
            p += (tab_lds_bytes<R>(G, Q, true) + 15) & ~(size_t)15;
        }
    }
    double *gstat = reinterpret_cast<double *>(p);  // per group: N valid, pmin, pmax of the percent-rank rescaling
    p += ((sizeof(double) * 3 * (size_t)G) + 15) & ~(size_t)15;
    R *xs = reinterpret_cast<R *>(p);
    if (!XG) p += (((size_t)T * sizeof(R)) + 15) & ~(size_t)15;
    U *gs = reinterpret_cast<U *>(p);  // sorted keys of every ranking row, row stride sort_w + 1 (odd: no bank conflicts)
    const int gstr = sort_w + 1;
  // cell_list (device list of *n_list_dev cell ids): redo only those cells (the fallback of k_qdm_cell)
  const int64_t n_cells = cell_list ? (int64_t)*n_list_dev : C;
  for (int64_t ci = blockIdx.x; ci < n_cells; ci += gridDim.x) {
    const int64_t c = cell_list ? (int64_t)cell_list[ci] : ci;
    CellTab<R> ct, ctg;
    bool tab_global = !TAB_LDS;
    if (af) {
        if (TAB_LDS) {
            ct = tab_stage<R>(tl, quantiles, 0, af + (size_t)c * G * Q, G, Q);
            tab_global = ct.any_valid < 0;
            if (!tab_global) tab_spline<R>(ct, tl, interp);
        }
        if (tab_global) { ctg = tab_stage<R>(tlg, quantiles, 0, af + (size_t)c * G * Q, G, Q); tab_spline<R>(ctg, tlg, interp); }
    }
    const R *x = sim + c * sc;
    if constexpr (XG) xs = const_cast<R *>(x);  // (read-only from here on)
    else {
        constexpr int NB = 8;  // the whole series is requested in batches of NB loads per thread
        for (int64_t t0 = threadIdx.x; t0 < T; t0 += (int64_t)NB * blockDim.x) {
            R v[NB];
#pragma unroll
            for (int k = 0; k < NB; k++) { const int64_t t = t0 + (int64_t)k * blockDim.x; v[k] = x[(t < T ? t : T - 1) * st]; }
#pragma unroll
            for (int k = 0; k < NB; k++) { const int64_t t = t0 + (int64_t)k * blockDim.x; if (t < T) xs[t] = v[k]; }
        }
    }
    __syncthreads();
    if (sort_w) {
        // sorted ranking rows: sort_w lanes per row, 64 / sort_w rows per wave; UNR independent rows per lane keep the
        // cross-lane latency of the bitonic network covered
        constexpr int UNR = 4;
        const int lane = threadIdx.x & 63, sub = lane / sort_w, hl = lane % sort_w, rpw = 64 / sort_w;
        const int per_round = (blockDim.x >> 6) * rpw * UNR;
        const unsigned long long rowmask = sort_w == 64 ? ~0ull : (((1ull << sort_w) - 1ull) << (sub * sort_w));
        for (int g0 = 0; g0 < G; g0 += per_round) {
            int g[UNR];
            U k[UNR];
#pragma unroll
            for (int u = 0; u < UNR; u++) {
                g[u] = g0 + (((int)(threadIdx.x >> 6)) * UNR + u) * rpw + sub;
                const bool gv = g[u] < G;
                const int p0 = gv ? grp_ptr[g[u]] : 0, len = gv ? grp_ptr[g[u] + 1] - p0 : 0;
                k[u] = KeyOf<R>::MAXK;
                if (hl < len) { const R v = xs[grp_idx[p0 + hl]]; if (v == v) k[u] = to_key(v + (R)0); }  // -0.0 ranks as +0.0
            }
            for (int kk = 2; kk <= sort_w; kk <<= 1)
                for (int j = kk >> 1; j > 0; j >>= 1) {
                    const bool up = (hl & kk) == 0, lower = (hl & j) == 0;
#pragma unroll
                    for (int u = 0; u < UNR; u++) {
                        const U o = __shfl_xor(k[u], j, 64);  // the partner stays inside the row: j < sort_w
                        k[u] = (lower == up) ? (k[u] < o ? k[u] : o) : (k[u] > o ? k[u] : o);
                    }
                }
#pragma unroll
            for (int u = 0; u < UNR; u++) {
                const unsigned long long valid = __ballot(k[u] != KeyOf<R>::MAXK);
                const int N = __popcll(valid & rowmask);
                const U kfirst = __shfl(k[u], sub * sort_w, 64), klast = __shfl(k[u], sub * sort_w + (N > 0 ? N - 1 : 0), 64);
                const int nmin = __popcll(__ballot(k[u] == kfirst && k[u] != KeyOf<R>::MAXK) & rowmask);
                const int nmax = __popcll(__ballot(k[u] == klast && k[u] != KeyOf<R>::MAXK) & rowmask);
                if (g[u] < G) {
                    gs[(size_t)g[u] * gstr + hl] = k[u];
                    if (hl == 0) {
                        const double Nd = (double)N;
                        gstat[3 * g[u]] = Nd;
                        gstat[3 * g[u] + 1] = (((double)nmin + 1.0) / 2.0) / Nd;
                        gstat[3 * g[u] + 2] = ((Nd - (double)nmax) + ((double)nmax + 1.0) / 2.0) / Nd;
                    }
                }
            }
        }
    } else
    // per-group statistics over the ranking row (grp_* = windowed rows when rank_window, else the exact groups)
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        int p0 = grp_ptr[g], p1 = grp_ptr[g + 1];
        int N = 0;
        R mn = nan_of<R>(), mx = nan_of<R>();
        for (int i = p0; i < p1; i++) {
            R v = xs[grp_idx[i]];
            if (v == v) { if (N == 0 || v < mn) mn = v; if (N == 0 || v > mx) mx = v; N++; }
        }
        int nmin = 0, nmax = 0;
        for (int i = p0; i < p1; i++) { R v = xs[grp_idx[i]]; nmin += (v == mn); nmax += (v == mx); }
        const double Nd = (double)N;
        gstat[3 * g] = Nd;
        gstat[3 * g + 1] = (((double)nmin + 1.0) / 2.0) / Nd;
        gstat[3 * g + 2] = ((Nd - (double)nmax) + ((double)nmax + 1.0) / 2.0) / Nd;
    }
    __syncthreads();
    // (a lambda so that the LDS table and the global one each get their own copy of the loop, with LDS loads in the
    // first; the table argument is unused when af == nullptr)
    auto elements = [&](const CellTab<R> &tab) __attribute__((always_inline)) {
    if (sort_w) {
        // two elements per thread and round: their four bisections (first key >= key(v), first key > key(v), fixed
        // step count over the row padded with MAXK) are independent chains of LDS reads
        constexpr int EU = 2;
        for (int64_t t0 = threadIdx.x; t0 < T; t0 += (int64_t)EU * blockDim.x) {
            R v[EU];
            int g[EU], lo[EU], hi[EU];
            bool ok[EU];
            U kv[EU];
            const U *row[EU];
#pragma unroll
            for (int e = 0; e < EU; e++) {
                const int64_t t = t0 + (int64_t)e * blockDim.x;
                const bool in = t < T;
                v[e] = in ? xs[t] : nan_of<R>();
                g[e] = in ? gidx[t] : -1;
                ok[e] = v[e] == v[e] && g[e] >= 0;
                row[e] = gs + (size_t)(ok[e] ? g[e] : 0) * gstr;
                kv[e] = ok[e] ? to_key(v[e] + (R)0) : (U)0;
                lo[e] = 0;
                hi[e] = 0;
            }
            for (int step = sort_w >> 1; step > 0; step >>= 1) {
#pragma unroll
                for (int e = 0; e < EU; e++) {
                    const U a = row[e][lo[e] + step - 1], b = row[e][hi[e] + step - 1];
                    lo[e] += a < kv[e] ? step : 0;
                    hi[e] += b <= kv[e] ? step : 0;
                }
            }
#pragma unroll
            for (int e = 0; e < EU; e++) {
                lo[e] += row[e][lo[e]] < kv[e] ? 1 : 0;
                hi[e] += row[e][hi[e]] <= kv[e] ? 1 : 0;
            }
            double sq[EU];
#pragma unroll
            for (int e = 0; e < EU; e++) {
                sq[e] = __builtin_nan("");
                if (ok[e]) {
                    const double N = gstat[3 * g[e]], pmin = gstat[3 * g[e] + 1], pmax = gstat[3 * g[e] + 2];
                    const double r = (double)lo[e] + (double)(hi[e] - lo[e] + 1) / 2.0;
                    const double pr = r / N;
                    sq[e] = interp == -1 ? r : pmax * (pr - pmin) / (pmax - pmin);  // interp -1 (ranks only): the average ranks themselves
                }
            }
#pragma unroll
            for (int e = 0; e < EU; e++) {
                const int64_t t = t0 + (int64_t)e * blockDim.x;
                if (t < T) {
                    if (sim_q) sim_q[c * osc + t * ost] = sq[e];
                    if (af) {
                        int nd_unused;
                        R a = celltab_eval<R>(tab, sq[e], true, G > 1 ? g[e] : 0, interp, extrap, nd_unused);
                        scen[c * osc + t * ost] = apply_corr<R>(v[e], a, kind);
                    }
                }
            }
        }
    } else
    for (int64_t t = threadIdx.x; t < T; t += blockDim.x) {
        R v = xs[t];
        int g = gidx[t];
        double sq = __builtin_nan("");
        if (v == v && g >= 0) {
            int less = 0, eq = 0;
            const int p0 = grp_ptr[g], p1 = grp_ptr[g + 1];
            for (int i = p0; i < p1; i++) { R u = xs[grp_idx[i]]; less += (u < v); eq += (u == v); }
            const double N = gstat[3 * g], pmin = gstat[3 * g + 1], pmax = gstat[3 * g + 2];
            double r = (double)less + (double)(eq + 1) / 2.0;
            double pr = r / N;
            sq = interp == -1 ? r : pmax * (pr - pmin) / (pmax - pmin);
        }
        if (sim_q) sim_q[c * osc + t * ost] = sq;
        if (af) {
            int nd_unused;
            R a = celltab_eval<R>(tab, sq, true, G > 1 ? g : 0, interp, extrap, nd_unused);
            scen[c * osc + t * ost] = apply_corr<R>(v, a, kind);
        }
    }
    };
    if (tab_global) elements(ctg);
    else elements(ct);
    __syncthreads();  // series / table are reused by the next cell
  }
}

// ---------------------------------------------------------------------------------------------
// max_tail_factor mask (_adjustment.py:648-658, 672-673): values above factor * (last raw hist quantile of their group)
// keep the (pre-processed) input instead of the adjusted value.  Elementwise, cell-major rows, in place on scen.
// ---------------------------------------------------------------------------------------------
template <typename R>
__global__ void k_tail_mask(const R *__restrict__ sim, int64_t T, int64_t C, int64_t sc, const R *__restrict__ hq_raw,
                            int G, int Q, const int *__restrict__ gidx, R factor, R *__restrict__ scen, int64_t osc)
{
    const int64_t c = blockIdx.y;
    const R *x = sim + c * sc;
    R *o = scen + c * osc;
    const R *last = hq_raw + (size_t)c * G * Q + (Q - 1);
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < (int)T; t += gridDim.x * blockDim.x) {
        const int g = G > 1 ? gidx[t] : 0;
        if (g < 0) continue;
        const R v = x[t], lim = factor * last[(size_t)g * Q];
        if (v > lim) o[t] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// layout kernels: time-major / arbitrary strides <-> cell-major contiguous [cell][time]
// ---------------------------------------------------------------------------------------------
template <typename E>
__global__ void k_transpose_tile(const E *__restrict__ src, int64_t rows, int64_t cols, int64_t src_rs,
                                 int64_t src_cs, E *__restrict__ dst, int64_t dst_rs, int64_t dst_cs)
{
    // dst[j*dst_rs' ...]: reads a 64x64 tile with src contiguous along `cols` (src_cs == 1) and writes it with
    // dst contiguous along `rows` (dst_rs == 1); LDS tile padded by one column to stay conflict-free.
    __shared__ E tile[64][65];
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 256 threads: 4 rows per pass
    for (int i = ty; i < 64; i += 4) {
        int64_t r = r0 + i, cc = c0 + tx;
        if (r < rows && cc < cols) tile[i][tx] = src[r * src_rs + cc * src_cs];
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        int64_t cc = c0 + i, r = r0 + tx;
        if (r < rows && cc < cols) dst[r * dst_rs + cc * dst_cs] = tile[tx][i];
    }
}

// The dense case of the layout change, src[rows][cols] -> dst[cols][rows], both contiguous: a tile of TR rows x TC columns per
// block of NT threads; a wave issues every load of its TR / (NT/64) rows before the first LDS store, and writes each column
// as TR / 64 pieces of 256 contiguous bytes.  Measured on 10 950 x 64 800 float32 (profiles/r03_transpose_tiles.txt):
// time-major -> cell-major 1.36 ms with <256, 64, 512> (the 64 x 64 tile above: 1.65-1.76 ms; the long run per cell matters
// because a cell's 43 800-byte row is not line-aligned); cell-major -> time-major 1.09 ms with <128, 128, 1024> (1.33 ms);
// a plain float4 copy of the same bytes takes 1.15-1.24 ms.
template <typename E, int TR, int TC, int NT>
__global__ __launch_bounds__(NT) void k_transpose_dense(const E *__restrict__ src, int64_t rows, int64_t cols, E *__restrict__ dst)
{
    extern __shared__ __align__(16) unsigned char xs_tr_smem[];
    E *tile = reinterpret_cast<E *>(xs_tr_smem);  // [TR][TC + 1]
    constexpr int LD = TC + 1, NW = NT / 64, PER = TR / NW, H = TC / 64, HT = TR / 64, PERC = TC / NW;
    static_assert(TR % 64 == 0 && TC % 64 == 0 && TR % NW == 0 && TC % NW == 0, "tile shape");
    const int64_t r0 = (int64_t)blockIdx.y * TR, c0 = (int64_t)blockIdx.x * TC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    E v[PER][H];
#pragma unroll
    for (int i = 0; i < PER; i++)
#pragma unroll
        for (int h = 0; h < H; h++) {
            const int64_t r = r0 + wave + i * NW, c = c0 + h * 64 + lane;
            v[i][h] = (r < rows && c < cols) ? src[r * cols + c] : E(0);
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
            const int64_t r = r0 + h * 64 + lane;
            if (r < rows && c < cols) dst[c * rows + r] = tile[(h * 64 + lane) * LD + j];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// synthetic cubes for bench.py / tests: counter-based, value = f(seed, array, cell, t), so any
// cell block can be regenerated independently on any rank.  generator 0: seasonal temperature-like
// (tests/test_utils.py:118-133 style); generator 1: gamma precipitation-like (testing/helpers.py:23-35
// shape/scale, Wilson-Hilferty transform of a normal deviate).  noleap calendar: doy = t mod 365.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ float u01(uint64_t h) { return ((float)(uint32_t)(h >> 40) + 0.5f) * (1.0f / 16777216.0f); }

template <typename R>
__global__ void k_synth(R *__restrict__ dst, int64_t T, int64_t C, int64_t st, int64_t sc, int generator, int array_id,
                        uint64_t seed, int64_t cell0, float nan_cell_frac, float nan_iid_frac)
{
    const int64_t n = T * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        // enumerate in storage order so that stores coalesce for either layout
        int64_t c, t;
        if (st == 1) { c = i / T; t = i - c * T; } else { t = i / C; c = i - t * C; }
        const uint64_t gc = (uint64_t)(cell0 + c);
        const uint64_t hc = mix64(seed * 0x100000001B3ull + gc);                       // per cell
        const uint64_t h = mix64(hc ^ mix64(((uint64_t)array_id << 40) + (uint64_t)t));  // per element
        float u = u01(h), u2 = u01(mix64(h));
        float v;
        if (generator == 0) {
            float off = 3.0f * sqrtf(-2.0f * logf(u01(hc))) * cospif(2.0f * u01(mix64(hc)));
            float amp = array_id == 0 ? 20.0f : 18.0f, base = array_id == 2 ? 281.0f : 280.0f;
            float noise = array_id == 2 ? 2.2f : 2.0f, tr = array_id == 0 ? 0.10f : (array_id == 1 ? 0.11f : 0.15f);
            int doy = (int)(t % 365) + 1;
            v = base - amp * cospif(2.0f * (float)doy / 365.0f) + noise * u + tr * (float)t / 365.0f + off;
        } else {
            float k = array_id == 0 ? 4.0f : (array_id == 1 ? 8.15f : 16.0f);
            float th = array_id == 0 ? 7.5f : (array_id == 1 ? 3.68f : 2.63f);
            float z = sqrtf(-2.0f * logf(u)) * cospif(2.0f * u2);
            float w = 1.0f - 1.0f / (9.0f * k) + z / (3.0f * sqrtf(k));
            v = fmaxf(k * th * w * w * w, 1e-3f);
        }
        if (nan_cell_frac > 0.f && u01(mix64(hc ^ 0xABCDEFull)) < nan_cell_frac) v = __builtin_nanf("");
        if (nan_iid_frac > 0.f && u01(mix64(h ^ 0x123457ull)) < nan_iid_frac) v = __builtin_nanf("");
        dst[c * sc + t * st] = (R)v;
    }
}

}  // namespace xs
