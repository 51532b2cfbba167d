// lin2d.hpp — grouped interp="linear" of the quantile-mapping adjustment (qm_adjust with a monthly / seasonal /
// day-of-year table, _adjustment.py:594-676 -> utils.interp_on_quantiles :408-513 -> _interp_on_quantiles_2d :380-400):
// the reference hands the (G + 2) * Q nodes (hist_q value, group coordinate) of a cell — its table wrapped with the
// cyclic rows of add_cyclic_bounds (:284-314) — to scipy.griddata(method="linear"), i.e. a Delaunay triangulation and the
// piecewise-linear interpolant on it, evaluated at (sim value, fractional group coordinate of the time step), then
// nbutils._extrapolate_on_quantiles (nbutils.py:397-416) replaces what lies below / above the (group-interpolated) first /
// last node by the (group-interpolated) first / last factor, or NaN.
//
//   k_lin2d_build_rows: one WORKGROUP per cell builds the cell's node list, its Delaunay triangulation (delaunay.hpp: row strips
//                   + pockets + arbitrated parallel Lawson flips) and the strip lists (the edges crossing every strip between two
//                   rows, left to right); in LDS for small tables, in global scratch for day-of-year ones (round 4)
//   k_lin2d_build : the one-THREAD-per-cell sweep of rounds 1-3 (XSDBA_HIP_LIN2D_BUILD=sweep; also the fallback inside the above)
//   k_lin2d_adjust: one BLOCK per cell stages the triangulation in LDS (when it fits) and maps the series: an element is located
//                   by a bisection over the edges of its strip (or, without strip lists, by a walk through the triangulation)
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
    size_t grad, adjp, adji, grec;  // cubic only: nodal gradients (staged), vertex adjacency in CSR form, 64-byte node records (build only)
    size_t plane;              // linear only (staged): the interpolant of every triangle as a plane, hdr[3] says whether they stand
    size_t sofs, sedge, stri;  // strip lists (staged): first entry of every strip, the crossing edges left to right, the triangle right of each
    size_t lock, work, cand;  // k_lin2d_build_rows only: a bid per triangle, the two work lists + their bitmaps, the flips wanted in a round
};

// a flip wanted in a round of k_lin2d_build_rows: its bid and the triangles it writes
struct Lin2DCand { uint32_t bid; uint16_t t, i, sites[4]; };
static_assert(sizeof(Lin2DCand) == 16, "one 16-byte record per wanted flip");

// tables up to this many nodes get 64-byte node records for the gradient sweeps (k_lin2d_grad): larger ones — day-of-year tables —
// are launched a few thousand cells at a time, where the sweeps wait for single reads and the records would only shrink the chunks
constexpr size_t LIN2D_REC_MAX_NODES = 2048;

// entries of a cell's strip lists (an edge counts once per strip it crosses: about 2.2 per node on tables like the bench's)
__host__ __device__ inline size_t lin2d_strip_cap(int G, int Q) { return (size_t)4 * (G + 2) * Q; }

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
    // strip lists (k_lin2d_build_rows; hdr[2] says whether the cell has them): between two consecutive rows no node lies, so the
    // edges crossing that strip are ordered left to right and exactly one triangle lies between two neighbours — a query is located
    // by a bisection over its strip's edges instead of a walk through the triangulation
    L.sofs = o; o += sizeof(int) * (size_t)(G + 2);
    o = (o + 15) & ~(size_t)15;
    L.sedge = o; o += 16 * lin2d_strip_cap(G, Q);
    L.stri = o; o += sizeof(uint16_t) * lin2d_strip_cap(G, Q);
    o = (o + 7) & ~(size_t)7;
    // planes (k_lin2d_build_rows): v = p[2] + p[0] * (x - x_ref) + p[1] * y on triangle t, p = plane + 3 t; plane[6 N] = x_ref
    L.plane = o; if (!cubic) o += sizeof(double) * (6 * N + 1);
    L.staged = o;
    L.hull = o; o += sizeof(uint16_t) * 4 * N;  // hull ring (3N) + insertion order (N): build only
    o = (o + 7) & ~(size_t)7;
    L.adjp = o; if (cubic) o += sizeof(int) * (N + 2);
    L.adji = o; if (cubic) o += sizeof(uint16_t) * (6 * N + 8);
    o = (o + 63) & ~(size_t)63;
    L.grec = o; if (cubic && N <= LIN2D_REC_MAX_NODES) o += 64 * N;  // (small tables only: see k_lin2d_grad)
    o = (o + 15) & ~(size_t)15;
    L.lock = o; o += sizeof(uint32_t) * 2 * N;
    L.work = o; o += sizeof(uint16_t) * 4 * N + 2 * (((2 * N + 31) / 32) * 4);
    o = (o + 15) & ~(size_t)15;
    L.cand = o; o += sizeof(Lin2DCand) * 2 * N;
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
    hdr[2] = 0;  // no strip lists: the adjustment walks
    hdr[3] = 0;  // no planes: it evaluates the barycentric form
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

// The same slice as k_lin2d_build leaves, built by ONE WORKGROUP per cell (delaunay.hpp: dtp_*): a lane per table row compacts
// the nodes, a lane per strip zips it, lane 0 scans the two pockets, then rounds of parallel Lawson flips — (A) every edge whose
// triangles changed in the previous round is tested, a wanted flip is recorded with the triangles it writes and bids for them
// (atomic min of a round-stamped edge number), (B) the flips that hold all their triangles are applied.  Tables the row
// construction does not take (a row of one node, fewer than two rows) are swept by lane 0 (dt_build).  cubic: the gradients are
// k_lin2d_grad's.  IN_LDS (tables of a few hundred nodes: monthly, seasonal): coordinates, triangles and the flip bookkeeping live
// in LDS (64 bytes per node) and the triangles are copied to the slice at the end — in global memory a round costs a chain of L2
// round trips per tested edge, 4.3 ms per cell; dynamic LDS: lin2d_rows_lds(G, Q, IN_LDS) bytes.
__host__ __device__ inline size_t lin2d_rows_lds(int G, int Q, bool in_lds)
{
    const size_t N = (size_t)(G + 2) * Q;
    return (size_t)(G + 3) * 24 + 64 + (in_lds ? N * 64 + 2 * (((2 * N + 31) / 32) * 4) : 0);
}

template <typename R, bool IN_LDS>
__global__ void k_lin2d_build_rows(const R *__restrict__ xq, const R *__restrict__ yq, const double *__restrict__ gcoord, int G, int Q,
                                   int64_t C, unsigned char *__restrict__ scratch, size_t stride, int xq_is_row, int cubic, int strips)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int NR = G + 2, tid = threadIdx.x, NT = blockDim.x;
    int *rowb = reinterpret_cast<int *>(smem);                       // [NR + 1]
    int *sbase = rowb + (NR + 1);                                    // [NR + 1]
    uint16_t *ne = reinterpret_cast<uint16_t *>(sbase + (NR + 1));   // [NR] each
    uint16_t *stk = ne + NR, *own_t = stk + NR, *own_s = own_t + NR;
    int *sofs = reinterpret_cast<int *>(smem + (size_t)(G + 3) * 16);  // [NR + 1] strip counts, then offsets
    uint16_t *start_t = reinterpret_cast<uint16_t *>(sofs + (NR + 1)), *start_i = start_t + NR;  // [NR] each: the left hull edge of a strip
    int *sh = reinterpret_cast<int *>(smem + (size_t)(G + 3) * 24);  // [0] m, [1] ntri, [2] flips wanted in the round, [3] length of the next list, [4] strips fine
    const Lin2DLayout L = lin2d_layout(G, Q, cubic != 0);
    const size_t N = (size_t)NR * Q;
    // IN_LDS working set: px | py (8 N each) | tv | tn (12 N each) | up | dn (2 N each), then over them the two work lists (4 N each;
    // the sweep's hull ring + order: 8 N) | lock (8 N) | wanted flips (N / 2 records) | the lists' bitmaps
    unsigned char *wk = smem + (size_t)(G + 3) * 24 + 64;
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        unsigned char *base = scratch + (size_t)c * stride;
        int *hdr = reinterpret_cast<int *>(base + L.hdr);
        double *meta = reinterpret_cast<double *>(base + L.meta);
        double *px = reinterpret_cast<double *>(base + L.px), *py = reinterpret_cast<double *>(base + L.py);
        double *pv = reinterpret_cast<double *>(base + L.pv);
        int *rowb_g = reinterpret_cast<int *>(base + L.rowb);
        double *gcs = reinterpret_cast<double *>(base + L.gc);
        // ---- the nodes: a lane per row counts, lane 0 sums, the lane writes (k_lin2d_build's rule, row by row)
        auto row_nodes = [&](int r, int n0, bool write) {
            const int g = ext_row(r, G);
            const R *xr = xq_is_row ? xq : xq + ((size_t)c * G + g) * Q, *yr = yq + ((size_t)c * G + g) * Q;
            double xf = __builtin_nan(""), xl = xf, yf = xf, yl = xf, last = 0.0;
            bool fx = false, fy = false;
            int n = n0;
            for (int k = 0; k < Q; k++) {
                const R xv = xr[k], yv = yr[k];
                if (xv == xv) { if (!fx) { xf = (double)xv; fx = true; } xl = (double)xv; }
                if (yv == yv) { if (!fy) { yf = (double)yv; fy = true; } yl = (double)yv; }
                if (xv == xv && yv == yv) {
                    if (n > n0 && !((double)xv > last)) continue;
                    last = (double)xv;
                    if (write) {
                        px[n] = (double)xv; py[n] = gcoord[r]; pv[n] = (double)yv;
                        if (IN_LDS) { reinterpret_cast<double *>(wk)[n] = (double)xv; reinterpret_cast<double *>(wk + 8 * N)[n] = gcoord[r]; }
                    }
                    n++;
                }
            }
            if (write) { meta[4 * r] = xf; meta[4 * r + 1] = xl; meta[4 * r + 2] = yf; meta[4 * r + 3] = yl; gcs[r] = gcoord[r]; }
            return n - n0;
        };
        for (int r = tid; r < NR; r += NT) sbase[r] = row_nodes(r, 0, false);
        __syncthreads();
        if (tid == 0) {
            int n = 0;
            for (int r = 0; r < NR; r++) { rowb[r] = n; n += sbase[r]; }
            rowb[NR] = n;
        }
        __syncthreads();
        const int n = rowb[NR];
        for (int r = tid; r < NR; r += NT) { row_nodes(r, rowb[r], true); rowb_g[r] = rowb[r]; }
        if (tid == 0) { rowb_g[NR] = n; hdr[0] = n; }
        Tri2D d;
        uint16_t *tv_g = reinterpret_cast<uint16_t *>(base + L.tv), *tn_g = reinterpret_cast<uint16_t *>(base + L.tn);
        d.px = IN_LDS ? reinterpret_cast<const double *>(wk) : px;
        d.py = IN_LDS ? reinterpret_cast<const double *>(wk + 8 * N) : py;
        d.tv = IN_LDS ? reinterpret_cast<uint16_t *>(wk + 16 * N) : tv_g;
        d.tn = IN_LDS ? reinterpret_cast<uint16_t *>(wk + 28 * N) : tn_g;
        d.hnext = IN_LDS ? reinterpret_cast<uint16_t *>(wk + 40 * N) : reinterpret_cast<uint16_t *>(base + L.hull);
        d.hprev = d.hnext + N;
        d.htri = d.hprev + N;
        d.n = n;
        d.ntri = 0;
        uint16_t *up = d.hnext, *dn = d.hprev;  // (the hull ring of the sweep is not needed by the row construction)
        // work lists of the rounds (the triangles that changed) + a bit per triangle each; in LDS they take the place of up / dn
        uint16_t *list0 = IN_LDS ? reinterpret_cast<uint16_t *>(wk + 40 * N) : reinterpret_cast<uint16_t *>(base + L.work);
        const int BW = (int)((2 * N + 31) / 32);
        uint32_t *bits0 = IN_LDS ? reinterpret_cast<uint32_t *>(wk + 64 * N) : reinterpret_cast<uint32_t *>(base + L.work + 8 * N);
        uint32_t *lock = IN_LDS ? reinterpret_cast<uint32_t *>(wk + 48 * N) : reinterpret_cast<uint32_t *>(base + L.lock);
        Lin2DCand *cand = IN_LDS ? reinterpret_cast<Lin2DCand *>(wk + 56 * N) : reinterpret_cast<Lin2DCand *>(base + L.cand);
        const int CAP = IN_LDS ? (int)(N / 2) : (int)(2 * N);
        const int NTRI_MAX = (int)(2 * N);
        for (int i = tid; i < n; i += NT) { up[i] = DT_NONE; dn[i] = DT_NONE; }
        for (int i = tid; i < NTRI_MAX; i += NT) lock[i] = 0xFFFFFFFFu;
        if (tid == 0) sh[0] = dtp_plan(rowb, NR, ne, sbase);
        __syncthreads();
        const int m = sh[0];
        int ntri = 0;
        if (m < 0) {
            if (tid == 0) {  // the sweep (k_lin2d_build)
                uint16_t *ord = d.htri + N;
                if (NR <= 32) dt_sort_order(d, ord);
                else for (int i = 0; i < n; i++) ord[i] = (uint16_t)i;
                sh[1] = dt_build(d, ord);
            }
            __syncthreads();
            ntri = sh[1];
        } else {
            for (int s2 = tid; s2 + 1 < m; s2 += NT) dtp_zip(d, rowb, ne, sbase, s2, up, dn);
            __syncthreads();
            for (int t = tid; t < sbase[m - 1]; t += NT) dtp_link(d, t, up, dn);
            __syncthreads();
            if (tid == 0) sh[1] = dtp_pockets(d, rowb, ne, m, sbase, stk, own_t, own_s);
            __syncthreads();
            ntri = sh[1];
            const int maxr = min(8 * n + 64, 16000);
            uint16_t *cur = list0, *nxt = list0 + 2 * N;
            uint32_t *bcur = bits0, *bnxt = bits0 + BW;
            for (int t = tid; t < ntri; t += NT) cur[t] = (uint16_t)t;  // (up / dn are dead: the link phase is behind a barrier)
            for (int w = tid; w < BW; w += NT) { bcur[w] = 0xFFFFFFFFu; bnxt[w] = 0u; }
            int ncur = ntri;
            bool settled = false;  // (block-uniform) a round found no edge left to flip: the triangulation is Delaunay
            for (int r = 1; r <= maxr; r++) {
                if (tid == 0) { sh[2] = 0; sh[3] = 0; }
                __syncthreads();
                // (A) test the edges of the triangles on the list (an edge between two of them from the lower-numbered one), record, bid
                for (int k = tid; k < ncur; k += NT) {
                    const int t = cur[k];
#pragma unroll
                    for (int i = 0; i < 3; i++) {
                        const int u = d.tn[3 * t + i];
                        if (u == DT_NONE || (u < t && ((bcur[u >> 5] >> (u & 31)) & 1u))) continue;
                        int sites[4];
                        if (!dtp_flip_sites(d, t, i, sites)) continue;
                        const int kc = atomicAdd(&sh[2], 1);
                        if (kc >= CAP) {  // (no room: the triangle stays on the list)
                            if (!((atomicOr(&bnxt[t >> 5], 1u << (t & 31)) >> (t & 31)) & 1u)) nxt[atomicAdd(&sh[3], 1)] = (uint16_t)t;
                            continue;
                        }
                        const uint32_t bid = ((uint32_t)(16383 - r) << 18) | (uint32_t)(3 * t + i);
                        Lin2DCand cd;
                        cd.bid = bid; cd.t = (uint16_t)t; cd.i = (uint16_t)i;
#pragma unroll
                        for (int k2 = 0; k2 < 4; k2++) {
                            cd.sites[k2] = (uint16_t)sites[k2];
                            if (sites[k2] != DT_NONE) atomicMin(&lock[sites[k2]], bid);
                        }
                        cand[kc] = cd;
                    }
                }
                __syncthreads();
                const int nc = min(sh[2], CAP);
                if (nc == 0) { settled = true; break; }  // (block-uniform)
                // (B) the flips that hold every triangle they write; their triangles (a loser: its first) go on the next list
                for (int k = tid; k < nc; k += NT) {
                    const Lin2DCand cd = cand[k];
                    bool mine = true;
#pragma unroll
                    for (int k2 = 0; k2 < 4; k2++) mine = mine && (cd.sites[k2] == DT_NONE || lock[cd.sites[k2]] == cd.bid);
                    const int t = cd.t, u = cd.sites[1];
                    if (mine) {
                        dtp_flip_apply(d, t, cd.i);
                        if (!((atomicOr(&bnxt[u >> 5], 1u << (u & 31)) >> (u & 31)) & 1u)) nxt[atomicAdd(&sh[3], 1)] = (uint16_t)u;
                    }
                    if (!((atomicOr(&bnxt[t >> 5], 1u << (t & 31)) >> (t & 31)) & 1u)) nxt[atomicAdd(&sh[3], 1)] = (uint16_t)t;
                }
                __syncthreads();
                ncur = sh[3];
                for (int w = tid; w < BW; w += NT) bcur[w] = 0u;  // (the old list's bits: the arrays swap)
                { uint16_t *tl = cur; cur = nxt; nxt = tl; }
                { uint32_t *tb = bcur; bcur = bnxt; bnxt = tb; }
                __syncthreads();
            }
            if (!settled) {
                // the rounds ran out with flips still wanted (never seen; a table far denser along its rows than they are apart
                // could): the triangulation in hand is not known to be Delaunay — the one-thread sweep builds the cell from scratch
                __syncthreads();
                if (tid == 0) {
                    uint16_t *ord = d.htri + N;
                    if (NR <= 32) dt_sort_order(d, ord);
                    else for (int i = 0; i < n; i++) ord[i] = (uint16_t)i;
                    sh[1] = dt_build(d, ord);
                }
                __syncthreads();
                ntri = sh[1];
            }
        }
        __syncthreads();
        if (tid == 0) hdr[1] = ntri;
        if (IN_LDS)
            for (int i = tid; i < 3 * ntri; i += NT) { tv_g[i] = d.tv[i]; tn_g[i] = d.tn[i]; }
        uint16_t *vt = reinterpret_cast<uint16_t *>(base + L.vt);
        for (int i = tid; i < n; i += NT) vt[i] = 0;
        __syncthreads();
        for (int t = tid; t < ntri; t += NT) { vt[d.tv[3 * t]] = (uint16_t)t; vt[d.tv[3 * t + 1]] = (uint16_t)t; vt[d.tv[3 * t + 2]] = (uint16_t)t; }
        // ---- linear: the interpolant of every triangle as a plane around x_ref (the barycentric form costs the adjustment twelve
        // LDS reads and two divisions per element; the plane three reads)
        if (!cubic) {
            double *plane = reinterpret_cast<double *>(base + L.plane);
            const double xref = n > 0 ? d.px[0] : 0.0;
            for (int t = tid; t < ntri; t += NT) {
                const int a = d.tv[3 * t], b = d.tv[3 * t + 1], c2 = d.tv[3 * t + 2];
                const double xa = d.px[a], ya = d.py[a], va = pv[a];
                const double bx = d.px[b] - xa, by = d.py[b] - ya, bv = pv[b] - va;
                const double cx = d.px[c2] - xa, cy = d.py[c2] - ya, cv = pv[c2] - va;
                const double det = bx * cy - cx * by;
                const double gx = (bv * cy - cv * by) / det, gy = (cv * bx - bv * cx) / det;
                plane[3 * t] = gx; plane[3 * t + 1] = gy; plane[3 * t + 2] = va - gx * (xa - xref) - gy * ya;
            }
            if (tid == 0) { plane[6 * N] = xref; hdr[3] = 1; }
        } else if (tid == 0) hdr[3] = 0;
        // ---- strip lists (strips: the coordinates are float-exact, host-checked): count the edges crossing every strip (a lane per
        // edge), note the strip's left hull edge, then a lane per strip walks from it to the right hull edge — the crossing edges
        // come out in order, each with the triangle on its right
        __syncthreads();
        if (strips && ntri > 0) {
            auto row_of = [&](int v) { int lo = 0, hi = NR - 1; while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (rowb[mid] <= v) lo = mid; else hi = mid - 1; } return lo; };
            for (int s2 = tid; s2 < NR; s2 += NT) { sofs[s2] = 0; start_t[s2] = DT_NONE; }
            if (tid == 0) sh[4] = 1;
            __syncthreads();
            for (int e = tid; e < 3 * ntri; e += NT) {
                const int t = e / 3, i = e - 3 * t;
                const int u = d.tn[3 * t + i];
                if (u != DT_NONE && u < t) continue;  // (every edge once)
                const int a = d.tv[3 * t + (i + 1) % 3], b = d.tv[3 * t + (i + 2) % 3];
                // (a row keeps its nodes together: an empty row has rowb[r] == rowb[r + 1] and is skipped by the bisection)
                const int ra = row_of(a), rb = row_of(b);
                if (ra == rb) continue;
                const int lo = min(ra, rb), hi = max(ra, rb);
                for (int s2 = lo; s2 < hi; s2++) atomicAdd(&sofs[s2], 1);
                if (u == DT_NONE && ra > rb)  // counter-clockwise a -> b runs downwards: the hull's left side
                    for (int s2 = lo; s2 < hi; s2++) { start_t[s2] = (uint16_t)t; start_i[s2] = (uint16_t)i; }
            }
            __syncthreads();
            if (tid == 0) {
                int acc = 0;
                for (int s2 = 0; s2 < NR; s2++) { const int c2 = sofs[s2]; sofs[s2] = acc; acc += c2; }
                sofs[NR] = acc;
                if ((size_t)acc > lin2d_strip_cap(G, Q)) sh[4] = 0;
            }
            __syncthreads();
            int *sofs_g = reinterpret_cast<int *>(base + L.sofs);
            float4 *sedge = reinterpret_cast<float4 *>(base + L.sedge);
            uint16_t *stri = reinterpret_cast<uint16_t *>(base + L.stri);
            if (sh[4]) {
                for (int s2 = tid; s2 < NR; s2 += NT) sofs_g[s2] = sofs[s2];
                for (int s2 = tid; s2 + 1 < NR; s2 += NT) {
                    int k = sofs[s2];
                    const int kend = sofs[s2 + 1];
                    int t = start_t[s2];
                    if (t == DT_NONE) { if (k != kend) sh[4] = 0; continue; }
                    const int i0 = start_i[s2];
                    int hi_v = d.tv[3 * t + (i0 + 1) % 3], lo_v = d.tv[3 * t + (i0 + 2) % 3];
                    const double yhi = gcoord[s2 + 1];
                    for (int guard = 0; guard <= kend - sofs[s2] && k < kend; guard++) {
                        sedge[k] = make_float4((float)d.px[lo_v], (float)d.py[lo_v], (float)d.px[hi_v], (float)d.py[hi_v]);
                        stri[k] = (uint16_t)t;
                        k++;
                        const int v0 = d.tv[3 * t], v1 = d.tv[3 * t + 1], v2 = d.tv[3 * t + 2];
                        const int cth = (v0 != lo_v && v0 != hi_v) ? v0 : ((v1 != lo_v && v1 != hi_v) ? v1 : v2);
                        // the third corner lies on or beyond one of the two rows: the way out is the edge from it to the far end
                        int opp;
                        if (d.py[cth] >= yhi) { opp = hi_v; hi_v = cth; } else { opp = lo_v; lo_v = cth; }
                        const int slot = v0 == opp ? 0 : (v1 == opp ? 1 : 2);
                        const int u = d.tn[3 * t + slot];
                        if (u == DT_NONE) {  // the hull's right side
                            if (k < kend) {
                                sedge[k] = make_float4((float)d.px[lo_v], (float)d.py[lo_v], (float)d.px[hi_v], (float)d.py[hi_v]);
                                stri[k] = DT_NONE;
                                k++;
                            }
                            t = DT_NONE;
                            break;
                        }
                        t = u;
                    }
                    if (k != kend || t != DT_NONE) sh[4] = 0;  // (never seen: the cell then walks)
                }
            }
            __syncthreads();
            if (tid == 0) hdr[2] = sh[4];
        } else if (tid == 0) hdr[2] = 0;
        __syncthreads();  // the row tables in LDS are reused by the next cell
    }
}

// cubic: the nodal gradients of the Clough-Tocher interpolant on a finished triangulation, a thread per cell (Gauss-Seidel sweeps
// in node order: sequential by definition)
__global__ void k_lin2d_grad(int G, int Q, int64_t C, unsigned char *__restrict__ scratch, size_t stride, int use_rec)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const Lin2DLayout L = lin2d_layout(G, Q, true);
    unsigned char *base = scratch + (size_t)c * stride;
    const int *hdr = reinterpret_cast<const int *>(base + L.hdr);
    const size_t N = (size_t)(G + 2) * Q;
    Tri2D d;
    d.px = reinterpret_cast<const double *>(base + L.px);
    d.py = reinterpret_cast<const double *>(base + L.py);
    d.tv = reinterpret_cast<uint16_t *>(base + L.tv);
    d.tn = reinterpret_cast<uint16_t *>(base + L.tn);
    d.hnext = reinterpret_cast<uint16_t *>(base + L.hull);
    d.hprev = d.hnext + N;
    d.htri = d.hprev + N;
    d.n = hdr[0];
    d.ntri = hdr[1];
    int *adjp = reinterpret_cast<int *>(base + L.adjp);
    uint16_t *adji = reinterpret_cast<uint16_t *>(base + L.adji);
    dt_adjacency(d, adjp, adji, d.hnext);
    if (!use_rec || (size_t)(G + 2) * Q > LIN2D_REC_MAX_NODES) {  // (few cells: the kernel waits for single reads, not for the L2 — the records' two extra passes would only add to it)
        dt_gradients(d, reinterpret_cast<const double *>(base + L.pv), adjp, adji, reinterpret_cast<double *>(base + L.grad));
        return;
    }
    // dt_gradients (delaunay.hpp) on 64-byte node records (x, y, value, gradient): with a thread per cell every read of a lane is a
    // sector of its own, and the kernel ran at the L2's bandwidth — coordinates, value and gradient of a neighbour in four arrays
    // were four or five sectors, the record is one.  Same sweeps, same order, same expressions.
    const int n = d.n;
    const double *pv = reinterpret_cast<const double *>(base + L.pv);
    double *rec = reinterpret_cast<double *>(base + L.grec);
    for (int i = 0; i < n; i++) { double *r = rec + 8 * (size_t)i; r[0] = d.px[i]; r[1] = d.py[i]; r[2] = pv[i]; r[3] = 0.0; r[4] = 0.0; }
    for (int iter = 0; iter < 400; iter++) {
        double err = 0.0;
        for (int ip = 0; ip < n; ip++) {
            const int j0 = adjp[ip], j1 = adjp[ip + 1];
            if (j1 == j0) continue;  // a node outside the triangulation
            double *me = rec + 8 * (size_t)ip;
            double Q0 = 0.0, Q1 = 0.0, Q3 = 0.0, s0 = 0.0, s1 = 0.0;
            const double x1 = me[0], y1 = me[1], f1 = me[2];
            for (int j = j0; j < j1; j++) {
                const double *nb = rec + 8 * (size_t)adji[j];
                const double ex = nb[0] - x1, ey = nb[1] - y1;
                const double Ln = sqrt(ex * ex + ey * ey), L3 = Ln * Ln * Ln;
                const double f2 = nb[2];
                const double df2 = -ex * nb[3] - ey * nb[4];
                Q0 += 4.0 * ex * ex / L3;
                Q1 += 4.0 * ex * ey / L3;
                Q3 += 4.0 * ey * ey / L3;
                s0 += (6.0 * (f1 - f2) - 2.0 * df2) * ex / L3;
                s1 += (6.0 * (f1 - f2) - 2.0 * df2) * ey / L3;
            }
            const double det = Q0 * Q3 - Q1 * Q1;
            const double r0 = (Q3 * s0 - Q1 * s1) / det, r1 = (-Q1 * s0 + Q0 * s1) / det;
            double change = fmax(fabs(me[3] + r0), fabs(me[4] + r1));
            me[3] = -r0;
            me[4] = -r1;
            change /= fmax(1.0, fmax(fabs(r0), fabs(r1)));
            err = fmax(err, change);
        }
        if (err < 1e-6) break;
    }
    double *grad = reinterpret_cast<double *>(base + L.grad);
    for (int i = 0; i < n; i++) { grad[2 * i] = rec[8 * (size_t)i + 3]; grad[2 * i + 1] = rec[8 * (size_t)i + 4]; }
}

// bytes of a cell's slice the adjust kernel stages in LDS: everything up to the planes (linear), which only the global path reads
__host__ __device__ inline size_t lin2d_lds_staged(const Lin2DLayout &L, bool cubic) { return cubic ? L.staged : L.plane; }

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
// scaled series).  dynamic LDS: lin2d_lds_staged() bytes when use_lds, else 0.
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
            for (size_t i = threadIdx.x; i < lin2d_lds_staged(L, cubic != 0) / 4; i += blockDim.x) dst[i] = src[i];
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
        // (the planes serve the tables that stay in global memory — day-of-year: one sector per element instead of ten; staged in LDS
        //  with the rest they cost a monthly table a block per CU: 10.5 -> 12.6 ms per 64 800 cells)
        const bool has_strips = hdr[2] != 0, has_plane = !USE_LDS && !cubic && hdr[3] != 0;
        const double *plane = reinterpret_cast<const double *>(base + L.plane);
        const double xref = has_plane ? plane[6 * (size_t)(G + 2) * Q] : 0.0;
        const int *sofs = reinterpret_cast<const int *>(base + L.sofs);
        const float4 *sedge = reinterpret_cast<const float4 *>(base + L.sedge);
        const uint16_t *stri = reinterpret_cast<const uint16_t *>(base + L.stri);
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
                const int jr = jrow[t];
                int tri;
                if (has_strips && jr >= 0) {
                    // the strip between rows jr and jr + 1: its crossing edges run left to right, (x_lo, y_lo, x_hi, y_hi) each; the
                    // first edge the query lies strictly left of ends the bisection, the triangle right of the one before holds it
                    const int s0 = sofs[jr], s1 = sofs[jr + 1];
                    auto side = [&](int k) {  // > 0: the query is left of edge k (dt_orient_q with the lower end first)
                        const float4 e4 = sedge[k];
                        return ((double)e4.z - (double)e4.x) * (g - (double)e4.y) - ((double)e4.w - (double)e4.y) * (vx - (double)e4.x);
                    };
                    auto beyond = [&](int k, double o) {  // the tolerance of dt_locate at a hull edge
                        const float4 e4 = sedge[k];
                        const double scale = (fabs((double)e4.z - (double)e4.x) + fabs((double)e4.w - (double)e4.y)) *
                                             (fabs(vx - (double)e4.x) + fabs(g - (double)e4.y) + 1.0);
                        return fabs(o) > 2.3e-14 * scale;
                    };
                    int lo = s0, hi = s1;
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (side(mid) > 0.0) hi = mid; else lo = mid + 1; }
                    tri = -1;
                    if (s1 > s0) {
                        if (lo == s0) { if (!beyond(s0, side(s0))) tri = stri[s0]; }  // left of the hull's left side: by a rounding only?
                        else {
                            tri = stri[lo - 1];
                            if (tri == DT_NONE) tri = (lo - 2 >= s0 && !beyond(lo - 1, side(lo - 1))) ? (int)stri[lo - 2] : -1;
                        }
                    }
                } else {
                    // walk, from a triangle of the node nearest in (row, x): the row next to g, bisection on its abscissae
                    const int r0 = jr == -1 ? 0 : (jr == -2 ? G : jr), r1 = r0 + 1;
                    int r = (g - gcoord[r0] <= gcoord[r1] - g) ? r0 : r1;
                    if (rowb[r + 1] == rowb[r]) r = r == r0 ? r1 : r0;
                    int hint = 0;
                    if (rowb[r + 1] > rowb[r]) {
                        int lo = rowb[r], hi = rowb[r + 1] - 1;
                        while (lo < hi) { const int mid = (lo + hi) >> 1; if (d.px[mid] < vx) lo = mid + 1; else hi = mid; }
                        hint = vt[lo];
                    }
                    tri = dt_locate(d, vx, g, hint);
                }
                if (tri >= 0)
                    a = (R)(cubic ? dt_eval_ct(d, pv, grad, tri, vx, g)
                                  : (has_plane ? plane[3 * tri + 2] + plane[3 * tri] * (vx - xref) + plane[3 * tri + 1] * g : dt_eval(d, pv, tri, vx, g)));
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
