/*
 * xsdba_oracle.c — CPU restatement ("oracle") of xsdba's empirical-quantile-mapping hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker / the reported CPU baseline.
 * The product path (xsdba_amd + libxsdba_hip.so) never links, imports or falls back to it.
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle_golden.py) against golden
 * vectors generated in the build container by executing the reference's own kernel source
 * (/root/reference/src/xsdba/nbutils.py via stub numba/xarray modules; utils.py functions
 * AST-extracted; real scipy interp1d/griddata) composed as _adjustment.py does — see
 * tests/golden/make_golden.py — and against the reference tests' known answers
 * (tests/test_utils.py:68-113,149-194; tests/test_nbutils.py:23-34; tests/test_adjustment.py:947-989).
 * Not restated (returns -3): grouped interp="linear"/"cubic", which the reference routes through
 * scipy.griddata's Qhull triangulation (utils.py:391-396).
 *
 * Rows added for SURVEY §8f (end of this file): tail_mask, jitter, adapt_freq, group_means / scaling_adjust, loci_train /
 * loci_adjust.  Their reference bodies are xarray orchestration (and unseeded random draws for jitter / adapt_freq) that
 * cannot be executed in the build container, so there are no reference-generated vectors for them: PARITY UNPINNED by
 * golden vectors; they are held to the assertions of the reference's own tests instead (tests/test_oracle_golden.py,
 * tests/test_processing_gpu.py, tests/test_api_gpu.py cite them).
 *
 * Conventions shared with include/xsdba_hip.h: series are addressed as x[c*stride_c + t*stride_t]
 * (element strides); tables are [cell][group][quantile] contiguous; group_of_time[t] in [0,G) is the
 * 0-based group of the window CENTRE (coordinate g+1), -1 = no group; dtype 0 = float32, 1 = float64;
 * interp 0 = nearest, 1 = linear; extrap 0 = constant, 1 = nan; kind '+' or '*'.
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    int G, W;
    int64_t T;
    const int32_t *group_of_time;
    int64_t *row_ptr;  /* [G+1] windowed rows: time index of every in-range window slot, with multiplicity */
    int32_t *row_idx;
    int64_t *mem_ptr;  /* [G+1] the times whose own group is g (window centres), in time order */
    int32_t *mem_idx;
    int64_t max_row;
} group_plan;

/* base.py:232-272: rolling(center=True, W).construct -> slot j of centre t' is x[t' + j - W//2];
 * groupby on the group of t'. */
static int plan_build(group_plan *pl, const int32_t *gidx, int64_t T, int G, int W)
{
    memset(pl, 0, sizeof(*pl));
    pl->G = G; pl->W = W; pl->T = T; pl->group_of_time = gidx;
    pl->row_ptr = (int64_t *)calloc((size_t)G + 1, sizeof(int64_t));
    pl->mem_ptr = (int64_t *)calloc((size_t)G + 1, sizeof(int64_t));
    if (!pl->row_ptr || !pl->mem_ptr) return -2;
    int h = W / 2;
    for (int64_t t = 0; t < T; t++) {
        int g = gidx[t];
        if (g < 0) continue;
        if (g >= G) return -1;
        int64_t lo = t - h, hi = t - h + W - 1;
        if (lo < 0) lo = 0;
        if (hi > T - 1) hi = T - 1;
        pl->row_ptr[g + 1] += hi - lo + 1;
        pl->mem_ptr[g + 1] += 1;
    }
    for (int g = 0; g < G; g++) {
        if (pl->row_ptr[g + 1] > pl->max_row) pl->max_row = pl->row_ptr[g + 1];
        pl->row_ptr[g + 1] += pl->row_ptr[g];
        pl->mem_ptr[g + 1] += pl->mem_ptr[g];
    }
    pl->row_idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(pl->row_ptr[G] + 1));
    pl->mem_idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(pl->mem_ptr[G] + 1));
    int64_t *rp = (int64_t *)malloc(sizeof(int64_t) * (size_t)G * 2), *mp = rp + G;
    if (!pl->row_idx || !pl->mem_idx || !rp) return -2;
    for (int g = 0; g < G; g++) { rp[g] = pl->row_ptr[g]; mp[g] = pl->mem_ptr[g]; }
    for (int64_t t = 0; t < T; t++) {
        int g = gidx[t];
        if (g < 0) continue;
        pl->mem_idx[mp[g]++] = (int32_t)t;
        for (int j = 0; j < W; j++) {
            int64_t tt = t + j - h;
            if (tt >= 0 && tt < T) pl->row_idx[rp[g]++] = (int32_t)tt;
        }
    }
    free(rp);
    return 0;
}

static void plan_free(group_plan *pl)
{
    free(pl->row_ptr); free(pl->row_idx); free(pl->mem_ptr); free(pl->mem_idx);
}

#define REAL float
#define SFX f32
#include "oracle_impl.inc"
#undef REAL
#undef SFX
#define REAL double
#define SFX f64
#include "oracle_impl.inc"
#undef REAL
#undef SFX

#define API __attribute__((visibility("default")))

API int xsdba_oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

API void xsdba_oracle_set_num_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

API int xsdba_oracle_quantile_groups(const void *x, int dtype, int64_t T, int64_t C, int64_t st, int64_t sc,
                                     const int32_t *gidx, int G, int W, const void *q, int Q, void *out)
{
    group_plan pl;
    int rc = plan_build(&pl, gidx, T, G, W);
    if (!rc)
        rc = dtype == 0 ? quantile_groups_f32((const float *)x, T, C, st, sc, &pl, (const float *)q, Q, (float *)out)
                        : quantile_groups_f64((const double *)x, T, C, st, sc, &pl, (const double *)q, Q, (double *)out);
    plan_free(&pl);
    return rc;
}

API int xsdba_oracle_eqm_train(const void *ref, const void *hist, int dtype, int64_t T, int64_t C, int64_t st,
                               int64_t sc, const int32_t *gidx, int G, int W, const void *q, int Q, char kind,
                               void *af, void *hist_q)
{
    group_plan pl;
    int rc = plan_build(&pl, gidx, T, G, W);
    if (!rc)
        rc = dtype == 0 ? train_f32((const float *)ref, (const float *)hist, T, C, st, sc, &pl, (const float *)q, Q,
                                    kind, (float *)af, (float *)hist_q, NULL)
                        : train_f64((const double *)ref, (const double *)hist, T, C, st, sc, &pl, (const double *)q,
                                    Q, kind, (double *)af, (double *)hist_q, NULL);
    plan_free(&pl);
    return rc;
}

API int xsdba_oracle_dqm_train(const void *ref, const void *hist, int dtype, int64_t T, int64_t C, int64_t st,
                               int64_t sc, const int32_t *gidx, int G, int W, const void *q, int Q, char kind,
                               void *af, void *hist_q, void *scaling)
{
    group_plan pl;
    int rc = plan_build(&pl, gidx, T, G, W);
    if (!rc)
        rc = dtype == 0 ? train_f32((const float *)ref, (const float *)hist, T, C, st, sc, &pl, (const float *)q, Q,
                                    kind, (float *)af, (float *)hist_q, (float *)scaling)
                        : train_f64((const double *)ref, (const double *)hist, T, C, st, sc, &pl, (const double *)q,
                                    Q, kind, (double *)af, (double *)hist_q, (double *)scaling);
    plan_free(&pl);
    return rc;
}

/* kind == 0: plain utils.interp_on_quantiles (returns the interpolated yq instead of sim (+|*) yq) */
API int xsdba_oracle_qm_adjust(const void *sim, int dtype, int64_t T, int64_t C, int64_t st, int64_t sc,
                               const void *af, const void *hist_q, int G, int Q, const int32_t *gidx, int interp,
                               int extrap, char kind, void *scen, int64_t ost, int64_t osc, int32_t *node_index)
{
    return dtype == 0 ? qm_adjust_f32((const float *)sim, T, C, st, sc, (const float *)af, (const float *)hist_q, G,
                                      Q, gidx, interp, extrap, kind, (float *)scen, ost, osc, node_index)
                      : qm_adjust_f64((const double *)sim, T, C, st, sc, (const double *)af, (const double *)hist_q,
                                      G, Q, gidx, interp, extrap, kind, (double *)scen, ost, osc, node_index);
}

API int xsdba_oracle_dqm_adjust(const void *sim, int dtype, int64_t T, int64_t C, int64_t st, int64_t sc,
                                const void *af, const void *hist_q, const void *scaling, int G, int Q,
                                const int32_t *gidx, int W, const double *time_num, int degree, int interp,
                                int extrap, char kind, void *scen, void *trend, int64_t ost, int64_t osc)
{
    group_plan pl;
    int rc = plan_build(&pl, gidx, T, G, W);
    if (!rc)
        rc = dtype == 0 ? dqm_adjust_f32((const float *)sim, T, C, st, sc, (const float *)af, (const float *)hist_q,
                                         (const float *)scaling, G, Q, &pl, time_num, degree, interp, extrap, kind,
                                         (float *)scen, (float *)trend, ost, osc)
                        : dqm_adjust_f64((const double *)sim, T, C, st, sc, (const double *)af,
                                         (const double *)hist_q, (const double *)scaling, G, Q, &pl, time_num, degree,
                                         interp, extrap, kind, (double *)scen, (double *)trend, ost, osc);
    plan_free(&pl);
    return rc;
}

API int xsdba_oracle_qdm_adjust(const void *sim, int dtype, int64_t T, int64_t C, int64_t st, int64_t sc,
                                const void *af, const void *quantiles, int G, int Q, const int32_t *gidx, int W,
                                int rank_window, int interp, int extrap, char kind, void *scen, double *sim_q,
                                int64_t ost, int64_t osc)
{
    group_plan pl;
    int rc = plan_build(&pl, gidx, T, G, W);
    if (!rc)
        rc = dtype == 0 ? qdm_adjust_f32((const float *)sim, T, C, st, sc, (const float *)af,
                                         (const float *)quantiles, G, Q, &pl, rank_window, interp, extrap, kind,
                                         (float *)scen, sim_q, ost, osc)
                        : qdm_adjust_f64((const double *)sim, T, C, st, sc, (const double *)af,
                                         (const double *)quantiles, G, Q, &pl, rank_window, interp, extrap, kind,
                                         (double *)scen, sim_q, ost, osc);
    plan_free(&pl);
    return rc;
}

/* the max_tail_factor mask of qm_adjust / dqm_adjust / qdm_adjust (_adjustment.py:648-658 + 672-673): last_quantile =
 * hist_q_raw.isel(quantiles=-1) broadcast on the group of each time step (utils.broadcast, "nearest" = the step's own
 * group), mask = sim > max_tail_factor * last_quantile (python float * float32 array -> float32 product),
 * scen = scen.where(~mask, sim).  In place on scen. */
API int xsdba_oracle_tail_mask(const void *sim, int dtype, int64_t T, int64_t C, int64_t st, int64_t sc,
                               const void *hq_raw, int G, int Q, const int32_t *gidx, double factor, void *scen,
                               int64_t ost, int64_t osc)
{
    for (int64_t c = 0; c < C; c++)
        for (int64_t t = 0; t < T; t++) {
            int g = G > 1 ? gidx[t] : 0;
            if (g < 0 || g >= G) continue;
            size_t iq = ((size_t)c * G + g) * Q + (Q - 1);
            if (dtype == 0) {
                float v = ((const float *)sim)[c * sc + t * st], lim = (float)factor * ((const float *)hq_raw)[iq];
                if (v > lim) ((float *)scen)[c * osc + t * ost] = v;
            } else {
                double v = ((const double *)sim)[c * sc + t * st], lim = factor * ((const double *)hq_raw)[iq];
                if (v > lim) ((double *)scen)[c * osc + t * ost] = v;
            }
        }
    return 0;
}

/* utils.rank(pct=True) through Grouper.apply for every cell: out[c*osc + t*ost] float64 */
API int xsdba_oracle_rank_groups(const void *x, int dtype, int64_t T, int64_t C, int64_t st, int64_t sc,
                                 const int32_t *gidx, int G, int W, int rank_window, double *out, int64_t ost,
                                 int64_t osc)
{
    group_plan pl;
    int rc = plan_build(&pl, gidx, T, G, W);
    if (rc) { plan_free(&pl); return rc; }
    size_t nb = (size_t)(pl.max_row > T ? pl.max_row : T);
#pragma omp parallel
    {
        void *xc = malloc((dtype ? 8 : 4) * ((size_t)T + nb));
        double *sq = (double *)malloc(sizeof(double) * (size_t)T);
#pragma omp for schedule(dynamic, 4)
        for (int64_t c = 0; c < C; c++) {
            if (dtype == 0) {
                float *xf = (float *)xc;
                for (int64_t t = 0; t < T; t++) xf[t] = ((const float *)x)[c * sc + t * st];
                group_pct_rank_f32(xf, T, &pl, rank_window, sq, xf + T);
            } else {
                double *xd = (double *)xc;
                for (int64_t t = 0; t < T; t++) xd[t] = ((const double *)x)[c * sc + t * st];
                group_pct_rank_f64(xd, T, &pl, rank_window, sq, xd + T);
            }
            for (int64_t t = 0; t < T; t++) out[c * osc + t * ost] = sq[t];
        }
        free(xc); free(sq);
    }
    plan_free(&pl);
    return 0;
}

/* PolyDetrend(degree).fit(x).ds.trend (detrending.py:59-88, 165-208): out[c*osc + t*ost] */
API int xsdba_oracle_polydetrend_trend(const void *x, int dtype, int64_t T, int64_t C, int64_t st, int64_t sc,
                                       const int32_t *gidx, int G, int W, const double *time_num, int degree,
                                       void *trend, int64_t ost, int64_t osc)
{
    if (degree < 0 || degree > 1) return -4;
    group_plan pl;
    int rc = plan_build(&pl, gidx, T, G, W);
    if (rc) { plan_free(&pl); return rc; }
#pragma omp parallel
    {
        void *buf = malloc((dtype ? 8 : 4) * (size_t)T * 3);
#pragma omp for schedule(dynamic, 4)
        for (int64_t c = 0; c < C; c++) {
            if (dtype == 0) {
                float *xc = (float *)buf, *tr = xc + T, *m = tr + T;
                for (int64_t t = 0; t < T; t++) xc[t] = ((const float *)x)[c * sc + t * st];
                group_trend_f32(xc, T, &pl, time_num, degree, tr, m);
                for (int64_t t = 0; t < T; t++) ((float *)trend)[c * osc + t * ost] = tr[t];
            } else {
                double *xc = (double *)buf, *tr = xc + T, *m = tr + T;
                for (int64_t t = 0; t < T; t++) xc[t] = ((const double *)x)[c * sc + t * st];
                group_trend_f64(xc, T, &pl, time_num, degree, tr, m);
                for (int64_t t = 0; t < T; t++) ((double *)trend)[c * osc + t * ost] = tr[t];
            }
        }
        free(buf);
    }
    plan_free(&pl);
    return 0;
}

/* nbutils._vecquantiles (nbutils.py:151-161): np.nanquantile(row, rnk) per row as numba evaluates it
 * (numba/np/arraymath.py _collect_percentiles_inner; numba is not importable here, so this row is restated from
 * numba's published algorithm: PARITY UNPINNED against the reference for this function; it is checked against
 * numpy's nanquantile, which agrees to float64 rounding). */
static double vecq_sorted(const double *a, int64_t n, double q)
{
    if (isnan(q) || n == 0) return NAN;
    if (n == 1) return a[0];
    double p = q * 100.0;
    if (p == 100.0) return a[n - 1];
    if (p == 0.0) return a[0];
    double rank = 1.0 + (double)(n - 1) * (p / 100.0);
    double f = floor(rank), m = rank - f;
    int64_t lo = (int64_t)f - 1;
    if (lo < 0) lo = 0;
    if (lo > n - 2) lo = n - 2;
    return a[lo] * (1.0 - m) + a[lo + 1] * m;
}

API int xsdba_oracle_vecquantiles(const void *x, int dtype, int64_t T, int64_t C, int64_t st, int64_t sc,
                                  const void *rnk, void *out)
{
#pragma omp parallel
    {
        double *buf = (double *)malloc(sizeof(double) * (size_t)(T > 0 ? T : 1));
#pragma omp for schedule(dynamic, 16)
        for (int64_t c = 0; c < C; c++) {
            int64_t n = 0;
            for (int64_t t = 0; t < T; t++) {
                double v = dtype == 0 ? (double)((const float *)x)[c * sc + t * st] : ((const double *)x)[c * sc + t * st];
                if (!isnan(v)) buf[n++] = v;
            }
            quick_sort_f64(buf, n);
            double q = dtype == 0 ? (double)((const float *)rnk)[c] : ((const double *)rnk)[c];
            double r = vecq_sorted(buf, n, q);
            if (dtype == 0) ((float *)out)[c] = (float)r; else ((double *)out)[c] = r;
        }
        free(buf);
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------------------------
 * Pre-processing rows.  The reference draws from numpy's unseeded global generator
 * (processing.py:234, _processing.py:122), so only the distribution of its output is defined.  This restatement and
 * the device kernels (xsdba_amd/csrc/preproc.hpp) share one counter-based generator, which pins them to each other
 * bit for bit; the reference's own properties are checked statistically in tests/.
 *   h(cell, t, stream) = mix64(mix64(seed ^ (cell + 1) * 0x9E3779B97F4A7C15) ^ (stream << 40) ^ t)
 *   u53(h) = (h >> 11) * 2^-53;  streams: 1 jitter lower, 2 jitter upper, 3 adapt_freq tie-break, 4 adapt_freq fill.
 * --------------------------------------------------------------------------------------------------------- */
static uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static uint64_t rng_hash(uint64_t seed, int64_t cell, int64_t t, int stream)
{
    return mix64(mix64(seed ^ ((uint64_t)(cell + 1) * 0x9E3779B97F4A7C15ull)) ^ ((uint64_t)stream << 40) ^ (uint64_t)t);
}
static double u53(uint64_t h) { return (double)(h >> 11) * (1.0 / 9007199254740992.0); }

/* processing.jitter (processing.py:178-257): x < lower -> U(nextafter(minimum, +inf), lower); x >= upper ->
 * U(upper, maximum) with maximum stepped down one ulp for dtypes narrower than float64; np.random.uniform(low, high)
 * = low + (high - low) * random_sample() in float64, cast to the data dtype.  NaN lower / upper = off. */
API int xsdba_oracle_jitter(const void *x, int dtype, int64_t T, int64_t C, int64_t st, int64_t sc, double lower,
                            double minimum, double upper, double maximum, uint64_t seed, int64_t cell0, void *out,
                            int64_t ost, int64_t osc)
{
    if (!isnan(upper) && isnan(maximum)) return -1;
    double m0 = isnan(minimum) ? 0.0 : minimum;
    double jmin = dtype == 0 ? (double)nextafterf((float)m0, INFINITY) : nextafter(m0, INFINITY);
    double jmax = dtype == 0 ? (double)nextafterf((float)maximum, -INFINITY) : maximum;
    for (int64_t c = 0; c < C; c++)
        for (int64_t t = 0; t < T; t++) {
            double v = dtype == 0 ? (double)((const float *)x)[c * sc + t * st] : ((const double *)x)[c * sc + t * st];
            double r = v;
            if (!isnan(v)) {
                if (!isnan(lower) && v < lower) r = jmin + (lower - jmin) * u53(rng_hash(seed, cell0 + c, t, 1));
                if (!isnan(upper) && v >= upper) r = upper + (jmax - upper) * u53(rng_hash(seed, cell0 + c, t, 2));
            }
            if (dtype == 0) ((float *)out)[c * osc + t * ost] = (r == v) ? ((const float *)x)[c * sc + t * st] : (float)r;
            else ((double *)out)[c * osc + t * ost] = r;
        }
    return 0;
}

typedef struct { double v; uint64_t tag; } af_item;  /* tag: 16 random bits << 32 | slot in the block (any block length) */
static int af_cmp(const void *a, const void *b)
{
    const af_item *x = (const af_item *)a, *y = (const af_item *)b;
    if (x->v < y->v) return -1;
    if (x->v > y->v) return 1;
    return x->tag < y->tag ? -1 : (x->tag > y->tag ? 1 : 0);
}

/* _processing._adapt_freq (_processing.py:75-142) on every (cell, group) block, dims [time, window] pooled; the time
 * series keeps the window-centre element of each block (base.py:438-439).  ref != NULL: training branch (P0_ref,
 * P0_hist, pth, dP0 are outputs); ref == NULL: adjust branch, the three are inputs (:76-83, 93-96).
 *   P0 = ecdf(x, thresh) = #(x <= thresh) / #notnull (utils.py:87-105)                         [:90-92]
 *   dP0 = where(P0_hist == 0, nan, (P0_hist - P0_ref) / P0_hist)                                [:94]
 *   pth = vecquantiles(ref, P0_hist).where(dP0 > 0): np.nanquantile in float64, cast to dtype   [:103-105]
 *   rnk = rank(sim, pct=True, use_random_tiebreak=True): a random order inside tie groups,
 *         k/N rescaled mx*(rnk-mn)/(mx-mn) (utils.py:622-635)                                   [:109]
 *   keep = (rnk < P0_ref/P0_hist*P0_sim) | (rnk > P0_sim) | isnull(sim)                         [:112]
 *   sim_ad = sim where keep or no adaptation needed, else (pth - thresh) * U[0,1).astype(dtype) + thresh  [:115-122]
 * Tie order: (value, 16 random bits of stream 3 mixed with the group, slot in the block). */
API int xsdba_oracle_adapt_freq(const void *sim, const void *ref, int dtype, int64_t T, int64_t C, int64_t st,
                                int64_t sc, const int32_t *gidx, int G, int W, double thresh, uint64_t seed,
                                int64_t cell0, double *P0_ref, double *P0_hist, void *pth, double *dP0_out,
                                void *sim_ad, int64_t ost, int64_t osc)
{
    group_plan pl;
    int rc = plan_build(&pl, gidx, T, G, W);
    if (rc) { plan_free(&pl); return rc; }
    if (!ref && W != 1) { plan_free(&pl); return -1; }
#define LOADV(p, c, t) (dtype == 0 ? (double)((const float *)(p))[(c) * sc + (t) * st] : ((const double *)(p))[(c) * sc + (t) * st])
    af_item *items = (af_item *)malloc(sizeof(af_item) * (size_t)(pl.max_row + 1));
    double *rbuf = (double *)malloc(sizeof(double) * (size_t)(pl.max_row + 1));
    for (int64_t c = 0; c < C; c++) {
        for (int64_t t = 0; t < T; t++) {  /* sim_ad starts as sim */
            if (dtype == 0) ((float *)sim_ad)[c * osc + t * ost] = ((const float *)sim)[c * sc + t * st];
            else ((double *)sim_ad)[c * osc + t * ost] = ((const double *)sim)[c * sc + t * st];
        }
        for (int g = 0; g < G; g++) {
            int64_t p0 = pl.row_ptr[g], L = pl.row_ptr[g + 1] - p0;
            size_t og = (size_t)c * G + g;
            int64_t N = 0, n0 = 0;
            for (int64_t i = 0; i < L; i++) {
                double v = LOADV(sim, c, pl.row_idx[p0 + i]);
                if (!isnan(v)) { N++; if (v <= thresh) n0++; }
            }
            double P0_sim = N ? (double)n0 / (double)N : NAN, p0r, p0h, pthv;
            if (ref) {
                int64_t nr = 0, c0r = 0;
                for (int64_t i = 0; i < L; i++) {
                    double v = LOADV(ref, c, pl.row_idx[p0 + i]);
                    if (!isnan(v)) { rbuf[nr++] = v; if (v <= thresh) c0r++; }
                }
                p0r = nr ? (double)c0r / (double)nr : NAN;
                p0h = P0_sim;
                double d = p0h == 0.0 ? NAN : (p0h - p0r) / p0h;
                pthv = NAN;
                if (d > 0.0) {
                    quick_sort_f64(rbuf, nr);
                    pthv = vecq_sorted(rbuf, nr, p0h);
                    if (dtype == 0) pthv = (double)(float)pthv;
                }
                P0_ref[og] = p0r; P0_hist[og] = p0h;
                if (dtype == 0) ((float *)pth)[og] = (float)pthv; else ((double *)pth)[og] = pthv;
            } else {
                p0r = P0_ref[og]; p0h = P0_hist[og];
                pthv = dtype == 0 ? (double)((const float *)pth)[og] : ((const double *)pth)[og];
            }
            double dP0 = p0h == 0.0 ? NAN : (p0h - p0r) / p0h;
            if (dP0_out) dP0_out[og] = dP0;
            if (!(dP0 > 0.0)) continue;
            uint64_t gm = mix64((uint64_t)(g + 1) * 0xD1B54A32D192ED03ull);
            int64_t m = 0;
            for (int64_t i = 0; i < L; i++) {
                int64_t tt = pl.row_idx[p0 + i];
                double v = LOADV(sim, c, tt);
                if (isnan(v)) continue;
                items[m].v = v;
                items[m].tag = ((uint64_t)(mix64(rng_hash(seed, cell0 + c, tt, 3) ^ gm) >> 48) << 32) | (uint64_t)(uint32_t)i;
                m++;
            }
            qsort(items, (size_t)m, sizeof(af_item), af_cmp);
            double lowb = (p0r / p0h) * P0_sim;
            for (int64_t p = 0; p < m; p++) {
                int64_t tt = pl.row_idx[p0 + (int64_t)(items[p].tag & 0xFFFFFFFFu)];
                if (gidx[tt] != g) continue;
                double rk = (double)(p + 1) / (double)N, mn = 1.0 / (double)N, mx = (double)N / (double)N;
                double rnk = mx * (rk - mn) / (mx - mn);
                int keep = (rnk < lowb) || (rnk > P0_sim);
                if (keep) continue;
                double u = u53(rng_hash(seed, cell0 + c, tt, 4));
                if (dtype == 0) ((float *)sim_ad)[c * osc + tt * ost] = ((float)pthv - (float)thresh) * (float)u + (float)thresh;
                else ((double *)sim_ad)[c * osc + tt * ost] = (pthv - thresh) * u + thresh;
            }
        }
    }
#undef LOADV
    free(items); free(rbuf);
    plan_free(&pl);
    return 0;
}

/* Scaling: scaling_train (_adjustment.py:941-958) needs the mean of every (cell, group) block with the
 * dims [time, window] pooled (xarray .mean(), skipna); scaling_adjust (:961-974) applies af[group of t] (utils.broadcast
 * "nearest" :180-248, apply_correction :149-165). */
API int xsdba_oracle_group_means(const void *x, int dtype, int64_t T, int64_t C, int64_t st, int64_t sc, const int32_t *gidx,
                                 int G, int W, void *out)
{
    group_plan pl;
    int rc = plan_build(&pl, gidx, T, G, W);
    if (rc) { plan_free(&pl); return rc; }
    for (int64_t c = 0; c < C; c++)
        for (int g = 0; g < G; g++) {
            double s = 0;
            int64_t n = 0;
            for (int64_t i = pl.row_ptr[g]; i < pl.row_ptr[g + 1]; i++) {
                int64_t t = pl.row_idx[i];
                double v = dtype == 0 ? (double)((const float *)x)[c * sc + t * st] : ((const double *)x)[c * sc + t * st];
                if (!isnan(v)) { s += v; n++; }
            }
            double m = n ? s / (double)n : NAN;
            if (dtype == 0) ((float *)out)[(size_t)c * G + g] = (float)m; else ((double *)out)[(size_t)c * G + g] = m;
        }
    plan_free(&pl);
    return 0;
}

/* utils.broadcast with interp="linear" (utils.py:222-246): the group table, cyclically extended (add_cyclic_bounds :284-314),
 * is interpolated at the fractional group index of every time step by xarray interp -> scipy interp1d "linear":
 * slope = (y_hi - y_lo) / (x_hi - x_lo) with y in the table dtype and unit node spacing, y = slope * (x - x_lo) + y_lo in
 * float64.  lo / hi / frac come from the caller (0-based cyclic group indexes, frac = x - x_lo). */
static double group_factor(const void *tab, int dtype, size_t row, int glo, const int32_t *ghi, const double *frac, int64_t t)
{
    if (dtype == 0) {
        const float *f = (const float *)tab + row;
        if (!frac) return (double)f[glo];
        return (double)(float)(f[ghi[t]] - f[glo]) * frac[t] + (double)f[glo];
    }
    const double *f = (const double *)tab + row;
    if (!frac) return f[glo];
    return (f[ghi[t]] - f[glo]) * frac[t] + f[glo];
}

API int xsdba_oracle_scaling_adjust(const void *sim, int dtype, int64_t T, int64_t C, int64_t st, int64_t sc, const void *af,
                                    const int32_t *gidx, int G, const int32_t *ghi, const double *frac, char kind, void *scen,
                                    int64_t ost, int64_t osc)
{
    for (int64_t c = 0; c < C; c++)
        for (int64_t t = 0; t < T; t++) {
            int g = G > 1 ? gidx[t] : 0;
            double a = g >= 0 ? group_factor(af, dtype, (size_t)c * G, g, ghi, frac, t) : NAN;
            if (dtype == 0) {
                float v = ((const float *)sim)[c * sc + t * st];
                float y = g < 0 ? NAN : (!frac ? (kind == '+' ? v + (float)a : v * (float)a) : (float)(kind == '+' ? (double)v + a : (double)v * a));
                ((float *)scen)[c * osc + t * ost] = y;
            } else {
                double v = ((const double *)sim)[c * sc + t * st];
                ((double *)scen)[c * osc + t * ost] = g < 0 ? NAN : (kind == '+' ? v + a : v * a);
            }
        }
    return 0;
}

/* LOCI (_adjustment.py:889-939).  loci_train: s_thresh = map_cdf(x=hist, y=ref, y_value=thresh) per block
 * (utils.py:35-44: q = searchsorted([-inf, sorted y], thresh, "right") / count(~isnan) = (#(y <= thresh) + 1) / (n + 1);
 * np.nanquantile(x, q), linear method), sth = its nearest-group broadcast, ws = hist where hist >= sth, wo = ref where
 * ref >= thresh (masks on the time series), ms / mo = block means (dims [time, window]), af = (mo - thresh) / (ms - s_thresh). */
API int xsdba_oracle_loci_train(const void *ref, const void *hist, int dtype, int64_t T, int64_t C, int64_t st, int64_t sc,
                                const int32_t *gidx, int G, int W, double thresh, void *af, void *hist_thresh)
{
    group_plan pl;
    int rc = plan_build(&pl, gidx, T, G, W);
    if (rc) { plan_free(&pl); return rc; }
#define LD(p, c, t) (dtype == 0 ? (double)((const float *)(p))[(c) * sc + (t) * st] : ((const double *)(p))[(c) * sc + (t) * st])
    double *buf = (double *)malloc(sizeof(double) * (size_t)(pl.max_row + 1));
    double *sth = (double *)malloc(sizeof(double) * (size_t)G);
    for (int64_t c = 0; c < C; c++) {
        for (int g = 0; g < G; g++) {
            int64_t nle = 0, nr = 0, nh = 0;
            for (int64_t i = pl.row_ptr[g]; i < pl.row_ptr[g + 1]; i++) {
                double v = LD(ref, c, pl.row_idx[i]);
                if (!isnan(v)) { nr++; if (v <= thresh) nle++; }
                double h = LD(hist, c, pl.row_idx[i]);
                if (!isnan(h)) buf[nh++] = h;
            }
            double q = ((double)nle + 1.0) / ((double)nr + 1.0), res = NAN;
            if (nh > 0) {
                quick_sort_f64(buf, nh);
                double vi = (double)(nh - 1) * q;
                int64_t lo = (int64_t)floor(vi);
                if (lo < 0) lo = 0;
                if (lo > nh - 1) lo = nh - 1;
                int64_t hi = lo + 1 > nh - 1 ? nh - 1 : lo + 1;
                res = buf[lo] + (buf[hi] - buf[lo]) * (vi - (double)lo);
            }
            if (dtype == 0) res = (double)(float)res;
            sth[g] = res;
            if (dtype == 0) ((float *)hist_thresh)[(size_t)c * G + g] = (float)res; else ((double *)hist_thresh)[(size_t)c * G + g] = res;
        }
        for (int g = 0; g < G; g++) {
            double so = 0, ss = 0;
            int64_t no = 0, ns = 0;
            for (int64_t i = pl.row_ptr[g]; i < pl.row_ptr[g + 1]; i++) {
                int64_t tt = pl.row_idx[i];
                double vo = LD(ref, c, tt), vs = LD(hist, c, tt);
                if (vo >= thresh) { so += vo; no++; }
                int gt = gidx ? gidx[tt] : 0;
                if (gt >= 0 && vs >= sth[gt]) { ss += vs; ns++; }
            }
            double mo = no ? so / (double)no : NAN, ms = ns ? ss / (double)ns : NAN;
            if (dtype == 0) ((float *)af)[(size_t)c * G + g] = ((float)mo - (float)thresh) / ((float)ms - (float)sth[g]);
            else ((double *)af)[(size_t)c * G + g] = (mo - thresh) / (ms - sth[g]);
        }
    }
#undef LD
    free(buf); free(sth);
    plan_free(&pl);
    return 0;
}

/* loci_adjust (_adjustment.py:922-939): scen = (factor * (sim - sth) + thresh).clip(min=0), nearest-group broadcast */
API int xsdba_oracle_loci_adjust(const void *sim, int dtype, int64_t T, int64_t C, int64_t st, int64_t sc, const void *af,
                                 const void *hist_thresh, const int32_t *gidx, int G, const int32_t *ghi, const double *frac,
                                 double thresh, void *scen, int64_t ost, int64_t osc)
{
    for (int64_t c = 0; c < C; c++)
        for (int64_t t = 0; t < T; t++) {
            int g = G > 1 ? gidx[t] : 0;
            if (dtype == 0) {
                float y = NAN;
                if (g >= 0 && !frac) {
                    y = ((const float *)af)[(size_t)c * G + g] * (((const float *)sim)[c * sc + t * st] - ((const float *)hist_thresh)[(size_t)c * G + g]) + (float)thresh;
                    if (y < 0.0f) y = 0.0f;
                } else if (g >= 0) {
                    double fa = group_factor(af, 0, (size_t)c * G, g, ghi, frac, t), th = group_factor(hist_thresh, 0, (size_t)c * G, g, ghi, frac, t);
                    double yd = fa * ((double)((const float *)sim)[c * sc + t * st] - th) + thresh;
                    if (yd < 0.0) yd = 0.0;
                    y = (float)yd;
                }
                ((float *)scen)[c * osc + t * ost] = y;
            } else {
                double y = NAN;
                if (g >= 0) {
                    double fa = group_factor(af, 1, (size_t)c * G, g, ghi, frac, t), th = group_factor(hist_thresh, 1, (size_t)c * G, g, ghi, frac, t);
                    y = fa * (((const double *)sim)[c * sc + t * st] - th) + thresh;
                    if (y < 0.0) y = 0.0;
                }
                ((double *)scen)[c * osc + t * ost] = y;
            }
        }
    return 0;
}
