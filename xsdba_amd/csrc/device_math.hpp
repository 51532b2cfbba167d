// device_math.hpp — per-element numerics shared by the gfx950 kernels.
//
// Every routine restates one reference function (citations: /root/reference/src/xsdba/...) and is the
// device twin of the same-named routine in oracle/oracle_impl.inc; the library is built with
// -ffp-contract=off so that float expressions round exactly like the numpy/numba code they follow.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace xs {

template <typename R> struct KeyOf;
template <> struct KeyOf<float> {
    using U = uint32_t;
    static constexpr U MAXK = 0xFFFFFFFFu;
};
template <> struct KeyOf<double> {
    using U = uint64_t;
    static constexpr U MAXK = 0xFFFFFFFFFFFFFFFFull;
};

// Order-preserving bijection float -> unsigned (ascending, -0 < +0); every NaN maps to MAXK so that NaNs
// (and row padding) sort last exactly like numba's arr.sort() (nbutils.py:137).
__device__ __forceinline__ uint32_t to_key(float v)
{
    if (v != v) return 0xFFFFFFFFu;
    uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ uint64_t to_key(double v)
{
    if (v != v) return 0xFFFFFFFFFFFFFFFFull;
    uint64_t u = (uint64_t)__double_as_longlong(v);
    return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ float from_key(uint32_t k)
{
    uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(u);
}
__device__ __forceinline__ double from_key(uint64_t k)
{
    uint64_t u = (k & 0x8000000000000000ull) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __longlong_as_double((long long)u);
}

template <typename R> __device__ __forceinline__ R nan_of() { return (R)__builtin_nan(""); }

// utils.get_correction / apply_correction / invert (utils.py:130-177), evaluated in the data dtype
template <typename R> __device__ __forceinline__ R get_corr(R x, R y, int kind) { return kind == '+' ? y - x : y / x; }
template <typename R> __device__ __forceinline__ R apply_corr(R x, R f, int kind) { return kind == '+' ? x + f : x * f; }
template <typename R> __device__ __forceinline__ R invert(R x, int kind) { return kind == '+' ? -x : (R)1 / x; }

// nbutils._nan_quantile_1d on a row already sorted ascending with its n valid values first
// (nbutils.py:113-148 with _get_indexes :29-68 and _linear_interpolation :76-105, numba typing:
// virtual index in float64 from the dtype-rounded q; gamma cast to the data dtype; gamma >= 0.5
// branch evaluated in float64 and rounded once; NaN result -> arr[n-1]).
template <typename R, typename Load> __device__ __forceinline__ R quantile_sorted(Load a, int n, R q)
{
    if (n == 0) return nan_of<R>();
    double qk = (double)q;
    double vi = (double)n * qk + (1.0 + qk * (1.0 - 1.0 - 1.0)) - 1.0;
    R res;
    if (vi != vi || vi >= (double)(n - 1)) res = a(n - 1);
    else if (vi < 0.0) res = a(0);
    else {
        int lo = (int)floor(vi);
        R left = a(lo), right = a(lo + 1);
        R gamma = (R)(vi - (double)lo);
        R diff = right - left;
        if (gamma >= (R)0.5) res = (R)((double)right - (double)diff * (1.0 - (double)gamma));
        else res = left + diff * gamma;
        if (res != res) res = a(n - 1);
    }
    return res;
}

// nbutils._vecquantiles (nbutils.py:151-161): `np.nanquantile(arr, rnk)` as numba implements it
// (numba/np/arraymath.py, _collect_percentiles_inner): everything in float64; p = q*100; rank = 1 + (n-1)*(p/100);
// f = floor(rank); m = rank - f; val = a[f-1]*(1-m) + a[f]*m; q = 0 / 1 -> min / max; one value -> that value;
// no valid value or NaN q -> NaN.  The float64 result is stored in the data dtype.
template <typename R, typename Load> __device__ __forceinline__ R vecquantile_sorted(Load a, int n, R q)
{
    if (q != q || n == 0) return nan_of<R>();
    if (n == 1) return a(0);
    double p = (double)q * 100.0;
    if (p == 100.0) return a(n - 1);
    if (p == 0.0) return a(0);
    double rank = 1.0 + (double)(n - 1) * (p / 100.0);
    double f = floor(rank);
    double m = rank - f;
    int lo = (int)f - 1;
    lo = lo < 0 ? 0 : (lo > n - 2 ? n - 2 : lo);
    return (R)((double)a(lo) * (1.0 - m) + (double)a(lo + 1) * m);
}

// ---------------------------------------------------------------------------------------------
// Interpolation of the trained factors on quantiles.  A cell's table lives in LDS as G rows of
// compacted (x, y) pairs (NaN pairs removed in place, order kept) + per-row metadata; the cyclic
// extension rows 0 and G+1 (utils.py:284-314, 489-490) are the same storage seen through ext_row().
// ---------------------------------------------------------------------------------------------
template <typename R> struct CellTab {
    const R *xs;        // [G][xstride] compacted abscissae (hist_q), or ONE row shared by all groups (xstride == 0:
                        // the quantile nodes of QDM when the cell's factors hold no NaN)
    const R *ys;        // [G][ystride] compacted factors
    const int *rn;      // [G] valid pairs per row
    const double *xfirst, *xlast;  // [G] first/last non-NaN abscissa (NaN if none) — nbutils.py:380-389
    const R *yfirst, *ylast;       // [G] first/last non-NaN factor
    int G, Q, any_valid, xstride;
    int ystride = 0;    // row stride of ys (0: Q); tab_stage pads rows whose length is a multiple of 16 (tab_row_stride)
    __device__ __forceinline__ size_t yrow(int g) const { return (size_t)g * (size_t)(ystride ? ystride : Q); }
    bool dense = false;  // no NaN anywhere in the cell's table: every row holds Q pairs, xfirst / xlast are its end abscissae
    const double *spl = nullptr;  // [n] second derivatives of the not-a-knot cubic spline of row 0 (interp = cubic, one group)
};

__device__ __forceinline__ int ext_row(int rr, int G) { return rr == 0 ? G - 1 : (rr == G + 1 ? 0 : rr - 1); }

// utils._interp_on_quantiles_1D (utils.py:350-377) with scipy.interpolate.interp1d 'nearest'
// (x_bds = x/2 summed pairwise in the table dtype, searchsorted side='left': ties go DOWN) or 'linear'
// (slope in the table dtype; evaluation in float64 when the abscissa is float64, the QDM path).
// Second derivatives M[0..n) of the cubic spline through (xs, ys) with not-a-knot end conditions — what
// scipy.interpolate.interp1d(kind="cubic") builds (make_interp_spline(k=3), its default boundary condition) — by the
// Thomas algorithm on the interior unknowns after eliminating M_0 and M_{n-1}; float64; w: 2 n doubles of work space.
// n < 4 (scipy refuses) or repeated abscissae (scipy raises) give NaN.  One thread.
template <typename R>
__device__ inline void spline_not_a_knot(const R *xs, const R *ys, int n, double *M, double *w)
{
    if (n < 4) { for (int i = 0; i < n; i++) M[i] = __builtin_nan(""); return; }
    double *cp = w, *dp = w + n;  // modified upper diagonal / right-hand side
    auto h = [&](int i) { return (double)xs[i + 1] - (double)xs[i]; };
    auto rhs = [&](int i) { return 6.0 * (((double)ys[i + 1] - (double)ys[i]) / h(i) - ((double)ys[i] - (double)ys[i - 1]) / h(i - 1)); };
    // unknowns M_1 .. M_{n-2}; row i: a_i M_{i-1} + b_i M_i + c_i M_{i+1} = r_i with the end rows folded in:
    //   M_0 = ((h0 + h1) M_1 - h0 M_2) / h1,  M_{n-1} = ((h_{n-3} + h_{n-2}) M_{n-2} - h_{n-2} M_{n-3}) / h_{n-3}
    for (int i = 1; i <= n - 2; i++) {
        double a = h(i - 1), b = 2.0 * (h(i - 1) + h(i)), c = h(i);
        if (i == 1) { const double h0 = h(0), h1 = h(1); b += h0 * (h0 + h1) / h1; c -= h0 * h0 / h1; a = 0.0; }
        if (i == n - 2) { const double hl = h(n - 2), hp = h(n - 3); b += hl * (hp + hl) / hp; a -= hl * hl / hp; c = 0.0; }
        const double r = rhs(i);
        if (i == 1) { cp[i] = c / b; dp[i] = r / b; }
        else { const double den = b - a * cp[i - 1]; cp[i] = c / den; dp[i] = (r - a * dp[i - 1]) / den; }
    }
    M[n - 2] = dp[n - 2];
    for (int i = n - 3; i >= 1; i--) M[i] = dp[i] - cp[i] * M[i + 1];
    M[0] = ((h(0) + h(1)) * M[1] - h(0) * M[2]) / h(1);
    M[n - 1] = ((h(n - 3) + h(n - 2)) * M[n - 2] - h(n - 2) * M[n - 3]) / h(n - 3);
}

template <typename R>
__device__ __forceinline__ R interp1d_eval(double x, bool x_is_double, const R *xs, const R *ys, int n, R yfirst,
                                           R ylast, int interp, int extrap, int &node, const double *spl = nullptr)
{
    node = -1;
    if (x != x || n == 0) return nan_of<R>();
    if (x < (double)xs[0]) return extrap == 0 ? yfirst : nan_of<R>();
    if (x > (double)xs[n - 1]) return extrap == 0 ? ylast : nan_of<R>();
    if (interp == 2) {  // cubic: the spline piece of the interval that holds x, in float64
        if (!spl || n < 4) return nan_of<R>();
        int lo = 0, hi = n - 1;  // xs[lo] <= x <= xs[hi]
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((double)xs[mid] <= x) lo = mid; else hi = mid; }
        const double x0 = (double)xs[lo], x1 = (double)xs[hi], hh = x1 - x0, a = x1 - x, b = x - x0;
        const double v = spl[lo] * a * a * a / (6.0 * hh) + spl[hi] * b * b * b / (6.0 * hh) +
                         ((double)ys[lo] / hh - spl[lo] * hh / 6.0) * a + ((double)ys[hi] / hh - spl[hi] * hh / 6.0) * b;
        return (R)v;
    }
    if (interp == 0) {
        int lo = 0, hi = n - 1;
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            R b = xs[mid] / (R)2.0 + xs[mid + 1] / (R)2.0;
            if ((double)b < x) lo = mid + 1; else hi = mid;
        }
        node = lo;
        return ys[lo];
    }
    if (n == 1) return nan_of<R>();
    int lo = 0, hi = n;
    while (lo < hi) { int mid = (lo + hi) >> 1; if ((double)xs[mid] < x) lo = mid + 1; else hi = mid; }
    int idx = lo < 1 ? 1 : (lo > n - 1 ? n - 1 : lo);
    R slope = (ys[idx] - ys[idx - 1]) / (xs[idx] - xs[idx - 1]);
    if (x_is_double) return (R)((double)slope * (x - (double)xs[idx - 1]) + (double)ys[idx - 1]);
    return slope * ((R)x - xs[idx - 1]) + ys[idx - 1];
}

// XF: the type the abscissa arrives in (float for a float32 series: the bisection then compares floats, which orders
// exactly like the float64 comparison of the same values; the squared distances are always float64)
template <typename R, typename XF> __device__ __forceinline__ double row_nearest(XF x, const R *xs, int n, int *which)
{
    int lo = 0, hi = n;
    while (lo < hi) { int mid = (lo + hi) >> 1; if ((XF)xs[mid] < x) lo = mid + 1; else hi = mid; }
    double best = __builtin_inf();
    int bi = -1;
    if (lo > 0) { double d = (double)x - (double)xs[lo - 1]; best = d * d; bi = lo - 1; }
    if (lo < n) { double d = (double)x - (double)xs[lo]; if (d * d < best) { best = d * d; bi = lo; } }
    *which = bi;
    return best;
}

// utils._interp_on_quantiles_2d with method="nearest" (utils.py:380-400: scipy griddata -> exact 2-D
// Euclidean nearest neighbour over (xq, group coordinate) in float64) followed by
// nbutils._extrapolate_on_quantiles (nbutils.py:397-416), for an integer group coordinate r = g0+1.
// Rows are visited outward (r, r-1, r+1, r-2, ...) while dr^2 < best; exact distance ties keep the
// first visited node (cKDTree's choice at ties is implementation-defined).
// row_nearest on a complete row (n >= 1 pairs, no compaction): the lower bound in a fixed number of steps
template <typename R, typename XF> __device__ __forceinline__ double row_nearest_fixed(XF x, const R *xs, int n, int *which)
{
    int lo = 0;
    for (int m = n; m > 1;) {
        const int half = m >> 1;
        lo = (XF)xs[lo + half - 1] < x ? lo + half : lo;
        m -= half;
    }
    lo += (XF)xs[lo] < x ? 1 : 0;  // first place whose abscissa is >= x, n if none
    double best = __builtin_inf();
    int bi = -1;
    if (lo > 0) { double d = (double)x - (double)xs[lo - 1]; best = d * d; bi = lo - 1; }
    if (lo < n) { double d = (double)x - (double)xs[lo]; if (d * d < best) { best = d * d; bi = lo; } }
    *which = bi;
    return best;
}

// The same with a guess `lo` of the lower bound (the one found in a neighbouring row: the rows of consecutive windowed groups are
// nearly the same numbers): when the guess brackets x — xs[lo - 1] < x <= xs[lo], which makes it THE first place >= x of a sorted
// row — the two reads of the test are the two reads the distances need; otherwise the fixed bisection runs.  `lo` is updated.
template <typename R, typename XF> __device__ __forceinline__ double row_nearest_hint(XF x, const R *xs, int n, int &lo, int *which)
{
    const int h = min(max(lo, 0), n);
    const XF below = h > 0 ? (XF)xs[h - 1] : (XF)0, at = h < n ? (XF)xs[h] : (XF)0;
    if ((h == 0 || below < x) && (h == n || !(at < x))) {
        double best = __builtin_inf();
        int bi = -1;
        if (h > 0) { double d = (double)x - (double)below; best = d * d; bi = h - 1; }
        if (h < n) { double d = (double)x - (double)at; if (d * d < best) { best = d * d; bi = h; } }
        *which = bi;
        lo = h;
        return best;
    }
    // (the bisection of row_nearest_fixed, keeping its lower bound for the next row)
    int l2 = 0;
    for (int m = n; m > 1;) {
        const int half = m >> 1;
        l2 = (XF)xs[l2 + half - 1] < x ? l2 + half : l2;
        m -= half;
    }
    l2 += (XF)xs[l2] < x ? 1 : 0;
    double best = __builtin_inf();
    int bi = -1;
    if (l2 > 0) { double d = (double)x - (double)xs[l2 - 1]; best = d * d; bi = l2 - 1; }
    if (l2 < n) { double d = (double)x - (double)xs[l2]; if (d * d < best) { best = d * d; bi = l2; } }
    *which = bi;
    lo = l2;
    return best;
}

// rows r-1, r+1, r-2, ... while dr^2 < best (the tail of interp2d_nearest: the element's own row has been searched)
// lo_hint (dense tables): the lower bound found in the element's own row
template <typename R, typename XF, bool DENSE = false>
__device__ __forceinline__ R nearest_outward(XF x, int r, const CellTab<R> &ct, double best, R val, int &node, int lo_hint = -1)
{
    const int G = ct.G, Q = ct.Q;
    int lo_dn = lo_hint, lo_up = lo_hint;  // the guesses travel outward with their side
    for (int dr = 1; dr <= G + 1; dr++) {
        double d2r = (double)dr * (double)dr;
        if (d2r >= best) break;
        for (int s = 0; s < 2; s++) {
            int rr = s ? r + dr : r - dr;
            if (rr < 0 || rr > G + 1) continue;
            int g = ext_row(rr, G);
            int w;
            double d2;
            if (DENSE) d2 = (lo_hint >= 0 ? row_nearest_hint<R, XF>(x, ct.xs + (size_t)g * ct.xstride, Q, s ? lo_up : lo_dn, &w)
                                           : row_nearest_fixed<R, XF>(x, ct.xs + (size_t)g * ct.xstride, Q, &w)) + d2r;
            else {
                int n = ct.rn[g];
                if (n == 0) continue;
                d2 = row_nearest<R, XF>(x, ct.xs + (size_t)g * ct.xstride, n, &w) + d2r;
            }
            if (d2 < best) { best = d2; val = ct.ys[ct.yrow(g) + w]; node = rr * Q + w; }
        }
    }
    return val;
}

template <typename R, typename XF>
__device__ __forceinline__ R interp2d_nearest(XF x, int g0, const CellTab<R> &ct, int extrap, int &node)
{
    node = -1;
    if (x != x || !ct.any_valid) return nan_of<R>();
    const int Q = ct.Q, r = g0 + 1;
    if ((double)x < ct.xfirst[g0]) return extrap == 0 ? ct.yfirst[g0] : nan_of<R>();
    if ((double)x > ct.xlast[g0]) return extrap == 0 ? ct.ylast[g0] : nan_of<R>();
    double best = __builtin_inf();
    R val = nan_of<R>();
    {   // the element's own row first, outside the outward loop: its nearest node is almost always closer than
        // one row step (d^2 < 1), which ends the search right here
        const int n = ct.rn[g0];
        if (n) {
            int w;
            best = row_nearest<R, XF>(x, ct.xs + (size_t)g0 * ct.xstride, n, &w);
            val = ct.ys[ct.yrow(g0) + w];
            node = r * Q + w;
        }
        if (best <= 1.0) return val;
    }
    return nearest_outward<R, XF>(x, r, ct, best, val, node);
}

// The same on a table without NaN (ct.dense; the series and the table share their type): the row is complete, its end
// abscissae are the extrapolation bounds (compared as they are: float against float orders like their doubles), the search
// is a lower bound in a fixed number of steps (no data-dependent trip count, no divergence), and the nearer of the two
// neighbours is decided on the exact float64 differences — the squares the general routine compares are exact products of
// those differences, so the decision, ties included (the lower node), is the same.
template <typename R>
__device__ __forceinline__ R interp2d_nearest_dense(R x, int g0, const CellTab<R> &ct, int extrap, int &node)
{
    node = -1;
    if (x != x) return nan_of<R>();
    const int Q = ct.Q, r = g0 + 1;
    const R *xr = ct.xs + (size_t)g0 * ct.xstride, *yr = ct.ys + ct.yrow(g0);
    if (x < xr[0]) return extrap == 0 ? yr[0] : nan_of<R>();
    if (x > xr[Q - 1]) return extrap == 0 ? yr[Q - 1] : nan_of<R>();
    int lo = 0;
    for (int n = Q; n > 1;) {  // first place whose abscissa is >= x (it exists: x <= xr[Q - 1])
        const int half = n >> 1;
        lo = xr[lo + half - 1] < x ? lo + half : lo;
        n -= half;
    }
    lo += xr[lo] < x ? 1 : 0;
    const double d1 = (double)xr[lo] - (double)x;
    int w = lo;
    double d = d1;
    if (lo > 0) {
        const double d0 = (double)x - (double)xr[lo - 1];
        if (!(d1 < d0)) { w = lo - 1; d = d0; }
    }
    const R val = yr[w];
    node = r * Q + w;
    const double best = d * d;
    if (best <= 1.0) return val;
    return nearest_outward<R, R, true>(x, r, ct, best, val, node, lo);
}

template <typename R>
__device__ __forceinline__ R celltab_eval(const CellTab<R> &ct, double x, bool x_is_double, int g0, int interp,
                                          int extrap, int &node)
{
    if (ct.G == 1)
        return interp1d_eval<R>(x, x_is_double, ct.xs, ct.ys, ct.rn[0], ct.yfirst[0], ct.ylast[0], interp, extrap, node, ct.spl);
    if (g0 < 0) { node = -1; return nan_of<R>(); }
    return interp2d_nearest<R, double>(x, g0, ct, extrap, node);
}

}  // namespace xs
