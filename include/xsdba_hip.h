/*
 * xsdba_hip.h — C ABI of libxsdba_hip.so: the MI355X (gfx950) implementation of xsdba's
 * empirical-quantile-mapping hot path (EQM / DQM / QDM train + adjust over grouped time windows).
 *
 * This is the drop-in boundary.  The reference (Ouranosinc/xsdba, pure Python + numba) has no FFI of
 * its own; each entry point below replaces one Python-level compute function that the reference's
 * TrainAdjust classes call, and is what a ctypes binding inside xsdba would bind (INTEGRATION.md shows
 * that stub).  Citations are file:line under /root/reference/src/xsdba/.
 *
 * Conventions
 *   - plain C symbols, `int` status: 0 ok; <0 argument / capability error (XSDBA_E_*); >0 a hipError_t.
 *     The message of the last failure on the calling thread: xsdba_hip_last_error().  Nothing throws.
 *   - every data pointer may be a HOST pointer or a DEVICE pointer (hipMalloc / xsdba_hip_malloc /
 *     any other allocator on the context's device); the library detects which (hipPointerGetAttributes).
 *     Host buffers are staged through device scratch owned by the context and results are copied back
 *     before the call returns.  With device pointers the call only enqueues work on the context's
 *     stream (call xsdba_hip_synchronize to wait); nothing is allocated after the first call of a
 *     given shape.
 *   - series blocks are addressed as x[c*stride_c + t*stride_t] (ELEMENT strides): stride_t == 1 is the
 *     cell-major [cell][time] layout, stride_c == 1 the time-major [time][cell] layout of an
 *     xarray (time, lat, lon) cube.  Host series must be dense (cover exactly n_cell*n_time elements).
 *   - trained tables are [cell][group][quantile] contiguous, `scaling` is [cell][group].
 *   - dtype: XSDBA_F32 or XSDBA_F64 for every floating array of the call (sim_q / time_num / newg are
 *     always float64 as in the reference).
 *   - group_of_time[t] in [0, n_groups): 0-based group of time step t (the reference's group coordinate
 *     is group_of_time[t] + 1: month 1..12, dayofyear 1..366, base.py:207-230); -1 = belongs to no group.
 *     `window` is Grouper.window (base.py:128-175): the row of group g holds x[t' + j - window/2] for every
 *     t' with group_of_time[t'] == g and j in [0, window), slots outside the series being NaN padding
 *     (xarray rolling(center=True).construct, base.py:261-262).
 *   - kind: '+' additive or '*' multiplicative (utils.py:130-177).
 *   - one context per (process, device); calls on one context are serialised by the caller, contexts on
 *     different devices are independent (one process per GPU under torch.distributed / mpirun).
 */
#ifndef XSDBA_HIP_H
#define XSDBA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XSDBA_HIP_ABI_VERSION 1

enum { XSDBA_F32 = 0, XSDBA_F64 = 1 };
enum { XSDBA_INTERP_NEAREST = 0, XSDBA_INTERP_LINEAR = 1, XSDBA_INTERP_CUBIC = 2 /* one group only: scipy interp1d(kind="cubic"), utils.py:370-376 */ };
enum { XSDBA_EXTRAP_CONSTANT = 0, XSDBA_EXTRAP_NAN = 1 };
/* `degree` of xsdba_hip_dqm_adjust: the trend was fitted by the caller and `trend` is an INPUT; detrended / retrended with
 * the kind of the call, or with the detrending object's own kind (detrending.py:102-120), the last one with
 * mult_skip_zeros (values whose trend is 0 are not divided, :112-113) */
enum { XSDBA_TREND_GIVEN = -1, XSDBA_TREND_GIVEN_ADD = -2, XSDBA_TREND_GIVEN_MUL = -3, XSDBA_TREND_GIVEN_MUL_SKIP_ZEROS = -4 };

enum {
    XSDBA_E_ARG = -1,          /* bad argument (null pointer, size, dtype, kind, group index out of range) */
    XSDBA_E_NOMEM = -2,        /* host allocation failed */
    XSDBA_E_UNSUPPORTED = -3,  /* an option this entry point does not serve, e.g. interp = LINEAR on a grouped table through
                                * xsdba_hip_qm_adjust (the griddata interpolants have their own entries: _qm_adjust_linear2d / _cubic2d) */
    XSDBA_E_DEGREE = -4,       /* PolyDetrend degree outside 0..6 */
    XSDBA_E_ROW_TOO_LONG = -5, /* a group row beyond 2^30 values, or a kernel whose LDS need exceeds the device */
    XSDBA_E_NO_DEVICE = -6     /* no usable gfx950 device / kernel image */
};

typedef struct xsdba_hip_ctx xsdba_hip_ctx;

/* ---- runtime ------------------------------------------------------------------------------- */
int xsdba_hip_abi_version(void);
int xsdba_hip_device_count(int *count);
int xsdba_hip_create(int device, xsdba_hip_ctx **ctx);
int xsdba_hip_destroy(xsdba_hip_ctx *ctx);
/* The per-cell sort kernels (train_cell.hpp: stable counting passes on wave-private counter rows) rely on ONE undocumented
 * hardware property: a returning LDS atomic serves the lanes of an instruction that hit the same word in ascending lane order.
 * xsdba_hip_create probes it (~0.5 ms, csrc/order_probe.hpp; XSDBA_HIP_ATOMIC_ORDER = probe | assume | broken); *ordered = 0
 * means the probe failed and this context computes everything those kernels would have served with the generic ones
 * (k_train_rows / k_qdm_adjust: same results, slower) — never a wrong quantile. */
int xsdba_hip_lane_ordered_atomics(xsdba_hip_ctx *ctx, int *ordered);
int xsdba_hip_last_error(char *buf, size_t buflen);
int xsdba_hip_synchronize(xsdba_hip_ctx *ctx);
/* the hipStream_t every kernel of this context is launched on (for event timing / stream ordering) */
int xsdba_hip_get_stream(xsdba_hip_ctx *ctx, void **stream);
/* Device buffers owned by the context.  A buffer of >= 1 MiB given back with xsdba_hip_free is parked in a per-context pool
 * (at most XSDBA_HIP_POOL_MB MiB — default: a quarter of THIS context's device memory, read at xsdba_hip_create —; 0 = hipFree at once) and serves
 * the next xsdba_hip_malloc of exactly that size without a hipMalloc / stream synchronisation: a train + adjust loop asks for the same tables and series every pass.
 * Reuse is ordered on the context's stream, like everything else the library does with the buffer; work that OTHER streams
 * queued on it must have been waited for before the free (that was already so when free synchronised this stream only).
 * The pool is emptied when a hipMalloc fails and at xsdba_hip_destroy. */
int xsdba_hip_malloc(xsdba_hip_ctx *ctx, size_t bytes, void **dptr);
int xsdba_hip_free(xsdba_hip_ctx *ctx, void *dptr);
int xsdba_hip_memcpy_h2d(xsdba_hip_ctx *ctx, void *dst, const void *src, size_t bytes);
int xsdba_hip_memcpy_d2h(xsdba_hip_ctx *ctx, void *dst, const void *src, size_t bytes);
int xsdba_hip_memset(xsdba_hip_ctx *ctx, void *dst, int value, size_t bytes);

/* ---- f4: streaming blocks of cells through the GPU (the role dask chunks play for the reference,
 * base.py:640-702).  A context is one HIP stream; copies, kernels and read-backs of different blocks overlap when they
 * are queued on different contexts of the same device and ordered by events:
 *   host_register / host_unregister  pin a caller-owned host buffer in place, so that memcpy_async from / to it is a
 *                                    true asynchronous DMA at PCIe rate (hipHostRegister);
 *   memcpy_async                     hipMemcpyAsync(hipMemcpyDefault) on the context's stream, returns immediately;
 *   event_record(ctx, slot)          records event `slot` (0..15) of `ctx` on its stream;
 *   stream_wait_event(w, src, slot)  work queued on w's stream after this call starts only when event `slot` of `src`
 *                                    (as last recorded) has completed; a never-recorded event does not block;
 *   event_synchronize(ctx, slot)     the HOST waits until event `slot` of `ctx` (as last recorded) has completed — what
 *                                    lets a pipeline unpin the host block it read back two steps ago without draining the
 *                                    streams (hipEventSynchronize); a never-recorded event returns at once. */
int xsdba_hip_host_register(xsdba_hip_ctx *ctx, void *host_ptr, size_t bytes);
int xsdba_hip_host_unregister(xsdba_hip_ctx *ctx, void *host_ptr);
int xsdba_hip_memcpy_async(xsdba_hip_ctx *ctx, void *dst, const void *src, size_t bytes);
int xsdba_hip_event_record(xsdba_hip_ctx *ctx, int slot);
int xsdba_hip_stream_wait_event(xsdba_hip_ctx *waiter, xsdba_hip_ctx *src, int slot);
int xsdba_hip_event_synchronize(xsdba_hip_ctx *ctx, int slot);

/* ---- e: multi-GPU.  Cells shard embarrassingly: one process per GPU, rank r owns a contiguous block of
 * the flattened cell index (the role dask's chunks over lat / lon play for the reference: map_blocks, base.py:640-650,
 * 715), nothing is exchanged during train / adjust, and the ONE collective is the final gather of the per-rank results —
 * RCCL point-to-point over xGMI, straight from / into the device buffers on the context's stream (no staging copy).
 * librccl.so is dlopen'ed on first use: a single-GPU user never needs it.
 *   shard_plan(n_cell, n_ranks, rank)  -> first cell and number of cells of `rank` (blocks differ by at most one cell and
 *                                         tile [0, n_cell) exactly); pure arithmetic, no context, no GPU;
 *   comm_unique_id(id)                 rank 0 fills the XSDBA_COMM_ID_BYTES-byte token; the CALLER hands the bytes to the
 *                                         other ranks by whatever rendezvous it has (a file, MPI, torch.distributed ...);
 *   comm_init(ctx, id, rank, n_ranks)  collective: every rank calls it with the same token (ncclCommInitRank);
 *   comm_info(ctx, &rank, &n_ranks)    what RCCL itself reports for the communicator (-1, 0 before comm_init);
 *   gather(ctx, send, n_local, row_elems, dtype, recv, n_cell, root)
 *                                      every rank sends its [n_local, row_elems] device block (n_local as in shard_plan)
 *                                      to `root`, which receives them at their places of the [n_cell, row_elems] device
 *                                      buffer `recv` (ignored elsewhere); asynchronous on the context's stream;
 *   comm_destroy(ctx)                  (also done by xsdba_hip_destroy). */
#define XSDBA_COMM_ID_BYTES 128
int xsdba_hip_shard_plan(int64_t n_cell, int n_ranks, int rank, int64_t *first_cell, int64_t *n_local);
int xsdba_hip_comm_unique_id(void *id);
int xsdba_hip_comm_init(xsdba_hip_ctx *ctx, const void *id, int rank, int n_ranks);
int xsdba_hip_comm_info(xsdba_hip_ctx *ctx, int *rank, int *n_ranks);
int xsdba_hip_gather(xsdba_hip_ctx *ctx, const void *send, int64_t n_local, int64_t row_elems, int dtype, void *recv,
                     int64_t n_cell, int root);
int xsdba_hip_comm_destroy(xsdba_hip_ctx *ctx);

/* Per-kernel timing with HIP events on the context's stream (bench.py's roofline leg).
 * enable != 0 brackets every kernel launch with an event pair; xsdba_hip_profile_read synchronises the
 * stream and returns, for up to `cap` distinct kernels, name / launch count / total milliseconds
 * (names are static strings); *n = number of entries.  reset != 0 clears the table afterwards. */
int xsdba_hip_profile_enable(xsdba_hip_ctx *ctx, int enable);
int xsdba_hip_profile_read(xsdba_hip_ctx *ctx, const char **names, int64_t *counts, double *total_ms, int cap,
                           int *n, int reset);
/* whole-region timing: record events on the context's stream around a region */
int xsdba_hip_timer_start(xsdba_hip_ctx *ctx);
int xsdba_hip_timer_stop(xsdba_hip_ctx *ctx, double *elapsed_ms);

/* ---- a4: nbutils.quantile through Grouper.apply (nbutils.py:113-148, 198-271; base.py:347-457) ----
 * out[c][g][k] = type-7 NaN-skipping quantile q[k] of the row of group g of cell c. */
int xsdba_hip_quantile_groups(xsdba_hip_ctx *ctx, const void *x, int dtype, int64_t n_time, int64_t n_cell,
                              int64_t stride_t, int64_t stride_c, const int32_t *group_of_time, int32_t n_groups,
                              int32_t window, const void *q, int32_t nq, void *out);

/* ---- a5: nbutils.vecquantiles / _vecquantiles (nbutils.py:151-195): a different quantile for every row:
 * out[c] = np.nanquantile(x[c, :], rnk[c]) as numba evaluates it (float64 linear interpolation between the
 * closest ranks); NaN rnk or no valid value -> NaN.  Used by adapt_freq (_processing.py:107). */
int xsdba_hip_vecquantiles(xsdba_hip_ctx *ctx, const void *x, int dtype, int64_t n_time, int64_t n_cell,
                           int64_t stride_t, int64_t stride_c, const void *rnk, void *out);

/* ---- a9: _adjustment.eqm_train (_adjustment.py:193-286), called by EmpiricalQuantileMapping._train and
 * QuantileDeltaMapping (adjustment.py:485-495, 674).  af = ref_q - hist_q or ref_q / hist_q. */
int xsdba_hip_eqm_train(xsdba_hip_ctx *ctx, const void *ref, const void *hist, int dtype, int64_t n_time,
                        int64_t n_cell, int64_t stride_t, int64_t stride_c, const int32_t *group_of_time,
                        int32_t n_groups, int32_t window, const void *q, int32_t nq, char kind, void *af,
                        void *hist_q);

/* ---- a10: _adjustment.dqm_train (_adjustment.py:95-190; adjustment.py:615-625): quantiles of the
 * group-mean-normalised series, plus scaling = mean(ref) - mean(hist) or mean(ref)/mean(hist). */
int xsdba_hip_dqm_train(xsdba_hip_ctx *ctx, const void *ref, const void *hist, int dtype, int64_t n_time,
                        int64_t n_cell, int64_t stride_t, int64_t stride_c, const int32_t *group_of_time,
                        int32_t n_groups, int32_t window, const void *q, int32_t nq, char kind, void *af,
                        void *hist_q, void *scaling);

/* ---- a11-a14: _adjustment.qm_adjust (_adjustment.py:594-676; adjustment.py:519-528) =
 * utils.interp_on_quantiles(sim, hist_q, af) (utils.py:408-513) + apply_correction.
 * n_groups == 1 is the ungrouped 1-D path (scipy interp1d semantics, utils.py:350-377; nearest or linear);
 * n_groups > 1 is the grouped path on the cyclically extended table (utils.py:284-314, 380-400) with the
 * true 2-D nearest neighbour and nbutils._extrapolate_on_quantiles (nbutils.py:397-416); interp must be
 * NEAREST there.  kind == 0 returns the interpolated factor itself.  node_index (nullable, int32
 * [cell][time]): flat index r*nq+k of the chosen node on the (n_groups+2)-row extended table, -1 when
 * the value came from extrapolation / NaN (parity instrumentation).
 * Layouts: float32 (time, cell) series on the device for BOTH sim and scen (stride_t = out_stride_t = n_cell, stride_c =
 * out_stride_c = 1: a (time, lat, lon) array as xarray holds it, base.py:653-656) are mapped in place, without a layout pass
 * (k_qm_adjust_tm, csrc/adjust_tm.hpp; grouped tables, every time step in a group, no node_index); other combinations of
 * strides go through the tiled layout kernel. */
int xsdba_hip_qm_adjust(xsdba_hip_ctx *ctx, const void *sim, int dtype, int64_t n_time, int64_t n_cell,
                        int64_t stride_t, int64_t stride_c, const void *af, const void *hist_q, int32_t n_groups,
                        int32_t nq, const int32_t *group_of_time, int interp, int extrap, char kind, void *scen,
                        int64_t out_stride_t, int64_t out_stride_c, int32_t *node_index);

/* ---- a12-a14 with grouped interp="linear": qm_adjust (_adjustment.py:594-676) through utils.interp_on_quantiles ->
 * _interp_on_quantiles_2d (utils.py:380-400, 476-513): scipy.griddata(method="linear") = Delaunay triangulation of the
 * (hist_q value, group coordinate) nodes of the cell's table wrapped with its cyclic rows (add_cyclic_bounds :284-314) and
 * the piecewise-linear interpolant on it, then nbutils._extrapolate_on_quantiles (nbutils.py:397-416).
 * group_coord [n_groups + 2]: coordinates of the wrapped rows (month: 0..13); new_group [n_time]: fractional group
 * coordinate of every time step (Grouper.get_index(interp=True), base.py:308-320).  trend (nullable, contiguous
 * [cell][time]): the series is detrended with it before the mapping and retrended after (the DQM variant; the caller
 * passes the scaled series).  Tied quantiles: of exactly repeated nodes one is kept (Qhull's own choice is undefined).
 * lookup (nullable, contiguous float64 [cell][time]): the abscissa looked up in the table when it is not the series itself,
 * and hist_q_is_row != 0: hist_q is ONE row [nq] shared by every group and cell — together the QDM form (_adjustment.py:
 * 873-881: af interpolated at (sim_q, group) over the regular (quantile, group) grid; every cell of that grid is a
 * cocircular quadruple, so which diagonal splits it is the triangulator's choice, here as in Qhull). */
int xsdba_hip_qm_adjust_linear2d(xsdba_hip_ctx *ctx, const void *sim, int dtype, int64_t n_time, int64_t n_cell,
                                 int64_t stride_t, int64_t stride_c, const void *af, const void *hist_q, int32_t n_groups,
                                 int32_t nq, const double *group_coord, const double *new_group, int extrap, char kind,
                                 const void *trend, const double *lookup, int hist_q_is_row, void *scen, int64_t out_stride_t,
                                 int64_t out_stride_c);

/* ---- the same call chain with grouped interp="cubic" (utils.py:391-396: scipy.griddata(method="cubic") =
 * CloughTocher2DInterpolator(tol=1e-6, maxiter=400) on the same nodes): nodal gradients by Gauss-Seidel sweeps over the
 * triangulation's edges, the C1 Clough-Tocher element on every triangle, then nbutils._extrapolate_on_quantiles as above.
 * Arguments as for xsdba_hip_qm_adjust_linear2d. */
int xsdba_hip_qm_adjust_cubic2d(xsdba_hip_ctx *ctx, const void *sim, int dtype, int64_t n_time, int64_t n_cell,
                                int64_t stride_t, int64_t stride_c, const void *af, const void *hist_q, int32_t n_groups,
                                int32_t nq, const double *group_coord, const double *new_group, int extrap, char kind,
                                const void *trend, const double *lookup, int hist_q_is_row, void *scen, int64_t out_stride_t,
                                int64_t out_stride_c);

/* ---- a15-a17: _adjustment.dqm_adjust (_adjustment.py:679-780; adjustment.py:659-668): nearest
 * broadcast of `scaling`, PolyDetrend(degree) fitted per group on the window means against `time_num`
 * (detrending.py:165-208; float64 abscissa, e.g. ns since 1970 as xarray.polyfit uses), quantile
 * mapping of the detrended series, retrend.  trend (nullable) has the layout of scen; with trend == NULL and degree <= 1
 * no trend array exists at all: the fit leaves (x_mean, y_mean, slope) per group and the mapping kernel evaluates the line
 * per element — what DetrendedQuantileMapping._adjust needs, which keeps `.scen` only (adjustment.py:659-668).
 * degree == XSDBA_TREND_GIVEN: `trend` is read instead of fitted (contiguous [cell][time]; time_num may be null) — the
 * pooled fit of Grouper(add_dims=...), where one trend from the mean over the pooled members (detrending.py:196-199)
 * serves all of them. */
int xsdba_hip_dqm_adjust(xsdba_hip_ctx *ctx, const void *sim, int dtype, int64_t n_time, int64_t n_cell,
                         int64_t stride_t, int64_t stride_c, const void *af, const void *hist_q,
                         const void *scaling, int32_t n_groups, int32_t nq, const int32_t *group_of_time,
                         int32_t window, const double *time_num, int32_t degree, int interp, int extrap, char kind,
                         void *scen, void *trend, int64_t out_stride_t, int64_t out_stride_c);

/* ---- a18-a19: _adjustment.qdm_adjust (_adjustment.py:783-886; adjustment.py:728-737): sim_q =
 * utils.rank(pct=True) within the exact group (rank_window == 0) or the windowed row (rank_window != 0)
 * (utils.py:573-638), af interpolated on the quantile axis, applied to sim.  sim_q (nullable) is float64
 * with the layout of scen. */
int xsdba_hip_qdm_adjust(xsdba_hip_ctx *ctx, const void *sim, int dtype, int64_t n_time, int64_t n_cell,
                         int64_t stride_t, int64_t stride_c, const void *af, const void *quantiles,
                         int32_t n_groups, int32_t nq, const int32_t *group_of_time, int32_t window,
                         int rank_window, int interp, int extrap, char kind, void *scen, double *sim_q,
                         int64_t out_stride_t, int64_t out_stride_c);

/* ---- a18: utils.rank through Grouper.apply (utils.py:573-638): float64 out[c*osc + t*ost].  rank_window bit 0: rank
 * within the windowed row instead of the exact group; bit 1 (value 2): the average ranks themselves (pct=False) instead
 * of the percent ranks rescaled to [0, 1] (pct=True) */
int xsdba_hip_rank_groups(xsdba_hip_ctx *ctx, const void *x, int dtype, int64_t n_time, int64_t n_cell,
                          int64_t stride_t, int64_t stride_c, const int32_t *group_of_time, int32_t n_groups,
                          int32_t window, int rank_window, double *out, int64_t out_stride_t, int64_t out_stride_c);

/* ---- f2: the max_tail_factor mask of qm_adjust / dqm_adjust / qdm_adjust (_adjustment.py:648-658 + 672-673,
 * :734-746 + 775-776, :846-857 + 882-883): scen[c][t] <- sim[c][t] wherever
 * sim[c][t] > max_tail_factor * hist_q_raw[c][group_of_time[t]][nq-1] (the product in the data dtype, as numpy's
 * weak scalar promotion does; NaN or a time step in no group never masks).  scen is read and written in place
 * (strides out_stride_*); hist_q_raw is [cell][group][quantile]. */
int xsdba_hip_tail_mask(xsdba_hip_ctx *ctx, const void *sim, int dtype, int64_t n_time, int64_t n_cell,
                        int64_t stride_t, int64_t stride_c, const void *hist_q_raw, int32_t n_groups, int32_t nq,
                        const int32_t *group_of_time, double max_tail_factor, void *scen, int64_t out_stride_t,
                        int64_t out_stride_c);

/* ---- f1: pre-processing in front of the quantile kernels.  Random draws are a pure function of
 * (seed, first_cell + cell, time step, stream) — see xsdba_amd/csrc/preproc.hpp — so results are reproducible and do
 * not depend on how cells are sharded over GPUs (pass the shard's first global cell index as first_cell).
 *
 * processing.jitter (processing.py:178-257): values < lower become U(nextafter(minimum), lower), values >= upper
 * become U(upper, maximum); lower / upper = NaN switches that end off (minimum NaN = 0). */
int xsdba_hip_jitter(xsdba_hip_ctx *ctx, const void *x, int dtype, int64_t n_time, int64_t n_cell, int64_t stride_t,
                     int64_t stride_c, double lower, double minimum, double upper, double maximum, uint64_t seed,
                     int64_t first_cell, void *out, int64_t out_stride_t, int64_t out_stride_c);

/* _processing._adapt_freq (_processing.py:20-142) for every (cell, group) block (dims [time, window] pooled).
 * ref != NULL (training, _adjustment.py:64-65): P0_ref, P0_hist (float64 [cell][group]), pth (data dtype
 * [cell][group]) and dP0 (nullable) are OUTPUTS.  ref == NULL (adjusting, _adjustment.py:639-645, window must be 1):
 * P0_ref, P0_hist, pth are INPUTS from the training.  sim_ad [cell][time] (strides out_stride_*) receives sim with the
 * frequency-adapted values. */
int xsdba_hip_adapt_freq(xsdba_hip_ctx *ctx, const void *sim, const void *ref, int dtype, int64_t n_time,
                         int64_t n_cell, int64_t stride_t, int64_t stride_c, const int32_t *group_of_time,
                         int32_t n_groups, int32_t window, double thresh, uint64_t seed, int64_t first_cell,
                         double *P0_ref, double *P0_hist, void *pth, double *dP0, void *sim_ad,
                         int64_t out_stride_t, int64_t out_stride_c);

/* ---- f3: Scaling (adjustment.py:985-1050, _adjustment.py:941-974).  group_means: the NaN-skipping mean of every
 * (cell, group) block, dims [time, window] pooled (scaling_train :953-958 takes two of them and their correction);
 * scaling_adjust: scen = sim (+|*) af[cell][group_of_time[t]]  (utils.broadcast with interp="nearest" :180-248, then
 * apply_correction); time steps in no group give NaN. */
int xsdba_hip_group_means(xsdba_hip_ctx *ctx, const void *x, int dtype, int64_t n_time, int64_t n_cell, int64_t stride_t,
                          int64_t stride_c, const int32_t *group_of_time, int32_t n_groups, int32_t window,
                          void *out /* [cell][group] */);
int xsdba_hip_scaling_adjust(xsdba_hip_ctx *ctx, const void *sim, int dtype, int64_t n_time, int64_t n_cell,
                             int64_t stride_t, int64_t stride_c, const void *af /* [cell][group] */,
                             const int32_t *group_of_time, int32_t n_groups, const int32_t *group_hi, const double *frac,
                             char kind, void *scen, int64_t out_stride_t, int64_t out_stride_c);
/* group_hi / frac (both NULL: nearest-group broadcast): utils.broadcast with interp="linear" (utils.py:222-246).  The
 * caller passes, per time step, the lower neighbour in group_of_time, the upper one in group_hi (cyclic, 0-based) and
 * frac = the fractional group index of Grouper.get_index(interp=True) minus the lower node (base.py:308-320); the factor is
 * (y_hi - y_lo) * frac + y_lo as scipy's interp1d evaluates it, applied in float64 and rounded once. */

/* ---- f3: LOCI (adjustment.py:933-1009, _adjustment.py:889-939).  loci_train: per (cell, group) block (dims
 * [time, window] pooled) hist_thresh = the value of hist with the same CDF as thresh in ref (utils.map_cdf :40-84:
 * q = (#(ref <= thresh) + 1) / (#valid + 1), np.nanquantile(hist, q)); af = (mean(ref >= thresh) - thresh) /
 * (mean(hist >= hist_thresh of the element's own group) - hist_thresh).  loci_adjust: scen = max(af * (sim -
 * hist_thresh) + thresh, 0) with the factors of each time step's group (nearest-group broadcast). */
int xsdba_hip_loci_train(xsdba_hip_ctx *ctx, const void *ref, const void *hist, int dtype, int64_t n_time, int64_t n_cell,
                         int64_t stride_t, int64_t stride_c, const int32_t *group_of_time, int32_t n_groups,
                         int32_t window, double thresh, void *af /* [cell][group] */, void *hist_thresh /* [cell][group] */);
int xsdba_hip_loci_adjust(xsdba_hip_ctx *ctx, const void *sim, int dtype, int64_t n_time, int64_t n_cell, int64_t stride_t,
                          int64_t stride_c, const void *af, const void *hist_thresh, const int32_t *group_of_time,
                          int32_t n_groups, const int32_t *group_hi, const double *frac, double thresh, void *scen,
                          int64_t out_stride_t, int64_t out_stride_c);

/* ---- a16: PolyDetrend(degree).fit(x).ds.trend (detrending.py:59-88, 165-208) */
int xsdba_hip_polydetrend_trend(xsdba_hip_ctx *ctx, const void *x, int dtype, int64_t n_time, int64_t n_cell,
                                int64_t stride_t, int64_t stride_c, const int32_t *group_of_time, int32_t n_groups,
                                int32_t window, const double *time_num, int32_t degree, void *trend,
                                int64_t out_stride_t, int64_t out_stride_c);

/* the same with the remaining options of PolyDetrend: degree 0..6 (least squares per group in the scaled variable
 * (x - x_mid) / x_halfrange; xarray.polyfit scales the Vandermonde columns and uses LAPACK's SVD — the fitted values agree
 * to the conditioning error of that route, ~1e-9 relative) and preserve_mean (detrending.py:204-205: the trend of every
 * group has its mean removed ('+') or divided out ('*')) */
int xsdba_hip_polydetrend_fit(xsdba_hip_ctx *ctx, const void *x, int dtype, int64_t n_time, int64_t n_cell, int64_t stride_t,
                              int64_t stride_c, const int32_t *group_of_time, int32_t n_groups, int32_t window,
                              const double *time_num, int32_t degree, int preserve_mean, char kind, void *trend,
                              int64_t out_stride_t, int64_t out_stride_c);

/* ---- a16: RollingMeanDetrend's trend on the whole series (detrending.py:342-356: da.rolling(center=True, min_periods,
 * time=window).mean()): NaN-skipping mean of the centred window of every time step ([t - window/2, t - window/2 + window - 1],
 * xarray's alignment for even windows too), NaN where fewer than min_periods values are present */
int xsdba_hip_rolling_mean(xsdba_hip_ctx *ctx, const void *x, int dtype, int64_t n_time, int64_t n_cell, int64_t stride_t,
                           int64_t stride_c, int32_t window, int32_t min_periods, void *out, int64_t out_stride_t,
                           int64_t out_stride_c);

/* ---- a7 / a16: utils.apply_correction of a series by another series (utils.py:149-177; BaseDetrend._detrend / _retrend,
 * detrending.py:102-120): out = x (+|*) f, or x (+|*) invert(f) when `inverted`; skip_zeros: out = x where f == 0
 * (mult_skip_zeros).  f has the layout (f_stride_t, f_stride_c); f_stride_c == 0 broadcasts one series over all cells. */
int xsdba_hip_series_correction(xsdba_hip_ctx *ctx, const void *x, int dtype, int64_t n_time, int64_t n_cell, int64_t stride_t,
                                int64_t stride_c, const void *f, int64_t f_stride_t, int64_t f_stride_c, char kind,
                                int inverted, int skip_zeros, void *out, int64_t out_stride_t, int64_t out_stride_c);

/* ---- bench / test utility: fill a DEVICE series block with the synthetic cubes of SURVEY §8d.
 * generator 0 = seasonal temperature-like, 1 = gamma precipitation-like; array_id 0/1/2 = ref/hist/sim;
 * value = f(seed, array_id, cell_offset + c, t): any cell block can be regenerated independently.
 * nan_cell_frac: fraction of cells that are entirely NaN; nan_iid_frac: i.i.d. NaN elsewhere. */
int xsdba_hip_synth_series(xsdba_hip_ctx *ctx, void *dst, int dtype, int64_t n_time, int64_t n_cell, int64_t stride_t,
                           int64_t stride_c, int generator, int array_id, uint64_t seed, int64_t cell_offset,
                           float nan_cell_frac, float nan_iid_frac);

#ifdef __cplusplus
}
#endif
#endif /* XSDBA_HIP_H */
