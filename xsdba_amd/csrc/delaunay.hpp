// delaunay.hpp — 2-D Delaunay triangulation + piecewise-linear interpolation of the (quantile value, group coordinate)
// nodes of ONE cell's trained table: what scipy.interpolate.griddata(method="linear") does for the reference's grouped
// interp="linear" (utils._interp_on_quantiles_2d, utils.py:380-400: Qhull Delaunay + LinearNDInterpolator; call site
// utils.py:476-513 with the cyclic rows of add_cyclic_bounds :284-314).  The Delaunay triangulation of points in general
// position is unique, so any exact construction gives Qhull's triangles; cocircular quadruples (where Qhull's choice is
// arbitrary) differ at most in which diagonal splits the quad.
//
// Construction: the nodes are inserted in (x, y) order, so every new node lies outside the hull of the earlier ones:
// sweep-hull insertion (fan to the visible hull edges) followed by Lawson flips.  One thread builds one cell; all loops are bounded.  Plain C++ (no HIP constructs beyond the
// XS_HD qualifier) so that the same source is exercised on the CPU by tests/test_delaunay_host.py against scipy.
#pragma once
#include <math.h>
#include <stdint.h>

#ifndef XS_HD
#define XS_HD __host__ __device__ __forceinline__
#endif

namespace xs {

constexpr uint16_t DT_NONE = 0xFFFFu;

// storage of one cell's triangulation (caller-provided; n nodes, at most 2 n triangles)
struct Tri2D {
    const double *px, *py;  // [n] node coordinates, no exact duplicates (stored row by row; inserted in (x, y) order)
    uint16_t *tv;           // [3 * 2n] triangle vertices, counter-clockwise
    uint16_t *tn;           // [3 * 2n] neighbour across the edge opposite each vertex (DT_NONE on the hull)
    uint16_t *hnext, *hprev, *htri;  // [n] hull as a counter-clockwise ring of vertices; htri[v]: triangle on edge v -> hnext[v]
    int n, ntri;
};

XS_HD double dt_orient(const Tri2D &d, int a, int b, int c)
{
    return (d.px[b] - d.px[a]) * (d.py[c] - d.py[a]) - (d.py[b] - d.py[a]) * (d.px[c] - d.px[a]);
}

XS_HD double dt_orient_q(const Tri2D &d, int a, int b, double qx, double qy)
{
    return (d.px[b] - d.px[a]) * (qy - d.py[a]) - (d.py[b] - d.py[a]) * (qx - d.px[a]);
}

// > 0: node q inside the circumcircle of the counter-clockwise triangle (a, b, c)
XS_HD double dt_incircle(const Tri2D &d, int a, int b, int c, int q)
{
    const double adx = d.px[a] - d.px[q], ady = d.py[a] - d.py[q];
    const double bdx = d.px[b] - d.px[q], bdy = d.py[b] - d.py[q];
    const double cdx = d.px[c] - d.px[q], cdy = d.py[c] - d.py[q];
    return (adx * adx + ady * ady) * (bdx * cdy - cdx * bdy) + (bdx * bdx + bdy * bdy) * (cdx * ady - adx * cdy) +
           (cdx * cdx + cdy * cdy) * (adx * bdy - bdx * ady);
}

XS_HD void dt_set(Tri2D &d, int t, int a, int b, int c, int na, int nb, int nc)
{
    d.tv[3 * t] = (uint16_t)a; d.tv[3 * t + 1] = (uint16_t)b; d.tv[3 * t + 2] = (uint16_t)c;
    d.tn[3 * t] = (uint16_t)na; d.tn[3 * t + 1] = (uint16_t)nb; d.tn[3 * t + 2] = (uint16_t)nc;
}

// in triangle u, the slot whose neighbour is t (DT_NONE-safe: returns 0 if absent, which the callers never hit)
XS_HD int dt_slot_of(const Tri2D &d, int u, int t)
{
    return d.tn[3 * u] == (uint16_t)t ? 0 : (d.tn[3 * u + 1] == (uint16_t)t ? 1 : 2);
}

// Lawson: make the edge opposite slot i of triangle t locally Delaunay, then the edges that a flip exposes
XS_HD void dt_legalize(Tri2D &d, int t0, int i0, int &flip_budget)
{
    int st[64], si[64];
    int sp = 0;
    st[sp] = t0; si[sp] = i0; sp++;
    while (sp > 0 && flip_budget > 0) {
        sp--;
        const int t = st[sp], i = si[sp];
        const int u = d.tn[3 * t + i];
        if (u == DT_NONE) continue;
        const int j = dt_slot_of(d, u, t);
        const int a = d.tv[3 * t + i], b = d.tv[3 * t + (i + 1) % 3], c = d.tv[3 * t + (i + 2) % 3];
        const int q = d.tv[3 * u + j];
        if (!(dt_incircle(d, a, b, c, q) > 0.0)) continue;
        flip_budget--;
        // t = (a, b, c), u = (q, c, b)  ->  t = (a, b, q), u = (a, q, c)
        const int X = d.tn[3 * u + (j + 1) % 3], Y = d.tn[3 * u + (j + 2) % 3];
        const int Z = d.tn[3 * t + (i + 1) % 3], W = d.tn[3 * t + (i + 2) % 3];
        dt_set(d, t, a, b, q, X, u, W);
        dt_set(d, u, a, q, c, Y, Z, t);
        if (X != DT_NONE) d.tn[3 * X + dt_slot_of(d, X, u)] = (uint16_t)t; else d.htri[b] = (uint16_t)t;   // hull edge b -> q
        if (Z != DT_NONE) d.tn[3 * Z + dt_slot_of(d, Z, t)] = (uint16_t)u; else d.htri[c] = (uint16_t)u;   // hull edge c -> a
        if (sp + 2 <= 64) {
            st[sp] = t; si[sp] = 0; sp++;  // edge (b, q), opposite a
            st[sp] = u; si[sp] = 0; sp++;  // edge (q, c), opposite a
        }
    }
}

// insertion order: the nodes sorted by (x, y) — an in-place heapsort of the index array (one thread, no recursion)
XS_HD bool dt_before(const Tri2D &d, int a, int b) { return d.px[a] < d.px[b] || (d.px[a] == d.px[b] && d.py[a] < d.py[b]); }

XS_HD void dt_sort_order(const Tri2D &d, uint16_t *ord)
{
    const int n = d.n;
    for (int i = 0; i < n; i++) ord[i] = (uint16_t)i;
    auto sift = [&](int root, int end) {
        const uint16_t v = ord[root];
        for (;;) {
            int child = 2 * root + 1;
            if (child >= end) break;
            if (child + 1 < end && dt_before(d, ord[child], ord[child + 1])) child++;
            if (!dt_before(d, v, ord[child])) break;
            ord[root] = ord[child];
            root = child;
        }
        ord[root] = v;
    };
    for (int i = n / 2 - 1; i >= 0; i--) sift(i, n);
    for (int end = n - 1; end > 0; end--) {
        const uint16_t t = ord[0]; ord[0] = ord[end]; ord[end] = t;
        sift(0, end);
    }
}

// Build the triangulation, inserting the nodes in the order `ord` (sorted by (x, y): every node then lies outside the hull
// of the earlier ones, and sees only a few edges of its right-hand side — sweeping row by row instead would fan every
// first node of a row to the whole row below, O(Q^2) flips per row).  Returns the number of triangles (0: fewer than 3
// nodes or all collinear).
XS_HD int dt_build(Tri2D &d, const uint16_t *ord)
{
    const int n = d.n;
    d.ntri = 0;
    if (n < 3) return 0;
    // ---- the collinear prefix c_0 .. c_{m-1} = ord[0 .. m)
    int m = 2;
    while (m < n && dt_orient(d, ord[0], ord[1], ord[m]) == 0.0) m++;
    if (m == n) return 0;
    const int p0 = ord[m];  // first node off the line
    const bool left = dt_orient(d, ord[0], ord[m - 1], p0) > 0.0;
    for (int i = 0; i + 1 < m; i++) {
        const int nb_next = i + 2 < m ? i + 1 : DT_NONE, nb_prev = i > 0 ? i - 1 : DT_NONE;
        if (left) dt_set(d, i, ord[i], ord[i + 1], p0, nb_next, nb_prev, DT_NONE);  // (c_i, c_i+1, p): opp c_i = edge (c_i+1, p)
        else dt_set(d, i, ord[i + 1], ord[i], p0, nb_prev, nb_next, DT_NONE);       // (c_i+1, c_i, p): opp c_i+1 = edge (c_i, p)
    }
    d.ntri = m - 1;
    if (left) {  // ring c0 -> c1 -> ... -> c_{m-1} -> p -> c0
        for (int i = 0; i + 1 < m; i++) { d.hnext[ord[i]] = ord[i + 1]; d.hprev[ord[i + 1]] = ord[i]; d.htri[ord[i]] = (uint16_t)i; }
        d.hnext[ord[m - 1]] = (uint16_t)p0; d.hprev[p0] = ord[m - 1]; d.htri[ord[m - 1]] = (uint16_t)(m - 2);
        d.hnext[p0] = ord[0]; d.hprev[ord[0]] = (uint16_t)p0; d.htri[p0] = 0;
    } else {     // ring c_{m-1} -> ... -> c0 -> p -> c_{m-1}
        for (int i = m - 1; i > 0; i--) { d.hnext[ord[i]] = ord[i - 1]; d.hprev[ord[i - 1]] = ord[i]; d.htri[ord[i]] = (uint16_t)(i - 1); }
        d.hnext[ord[0]] = (uint16_t)p0; d.hprev[p0] = ord[0]; d.htri[ord[0]] = 0;
        d.hnext[p0] = ord[m - 1]; d.hprev[ord[m - 1]] = (uint16_t)p0; d.htri[p0] = (uint16_t)(m - 2);
    }
    int flip_budget = 32 * n;
    int last = p0;
    // ---- sweep: every further node is outside the hull of the earlier ones
    for (int k = m + 1; k < n; k++) {
        const int p = ord[k];
        // a visible hull edge (p strictly to its right), searching from the previous node
        int a = last, steps = 0;
        bool found = false;
        // backward first: the edge arriving at `last` is the usual candidate
        {
            int e = d.hprev[last];
            if (dt_orient(d, e, last, p) < 0.0) { a = e; found = true; }
        }
        while (!found && steps <= n) {
            if (dt_orient(d, a, d.hnext[a], p) < 0.0) { found = true; break; }
            a = d.hnext[a];
            steps++;
        }
        if (!found) continue;  // cannot happen for sorted, duplicate-free input; the node is then left out
        // extend the visible chain backwards and forwards
        int first = a;
        for (steps = 0; steps <= n; steps++) {
            const int e = d.hprev[first];
            if (e == a || !(dt_orient(d, e, first, p) < 0.0)) break;
            first = e;
        }
        int end = d.hnext[a];  // chain first -> ... -> end, all edges visible
        for (steps = 0; steps <= n; steps++) {
            const int e = d.hnext[end];
            if (end == first || !(dt_orient(d, end, e, p) < 0.0)) break;
            end = e;
        }
        // fan
        int prev_t = DT_NONE, v = first;
        const int t_first = d.ntri;
        while (v != end) {
            const int w = d.hnext[v];
            const int t = d.ntri++;
            const int old = d.htri[v];
            // (w, v, p): opposite w = edge (v, p) -> previous fan triangle; opposite v = edge (p, w) -> next; opposite p -> old
            dt_set(d, t, w, v, p, prev_t, DT_NONE, old);
            if (prev_t != DT_NONE) d.tn[3 * prev_t + 1] = (uint16_t)t;
            if (old != DT_NONE) {
                // in `old` the edge v -> w is the one without a neighbour among those containing v and w
                for (int kk = 0; kk < 3; kk++) {
                    const int o1 = d.tv[3 * old + (kk + 1) % 3], o2 = d.tv[3 * old + (kk + 2) % 3];
                    if (o1 == v && o2 == w) d.tn[3 * old + kk] = (uint16_t)t;
                }
            }
            prev_t = t;
            v = w;
        }
        const int t_last = d.ntri - 1;
        d.hnext[first] = (uint16_t)p; d.hprev[p] = (uint16_t)first; d.htri[first] = (uint16_t)t_first;
        d.hnext[p] = (uint16_t)end; d.hprev[end] = (uint16_t)p; d.htri[p] = (uint16_t)t_last;
        for (int t = t_first; t <= t_last; t++) dt_legalize(d, t, 2, flip_budget);
        last = p;
    }
    return d.ntri;
}

// ---------------------------------------------------------------------------------------------------------------------
// Row-parallel construction (round 4).  The nodes of a table lie on R horizontal rows (ascending y), ascending x inside a
// row: ANY triangulation of the convex hull can be flipped into the Delaunay one (Lawson), and a table has an obvious one —
// the strips between consecutive non-empty rows, each zipped left to right, plus the pockets between the polylines of the
// rows' first / last nodes and the hull.  The strips are independent (a lane each), the flips of a round are independent
// once every flip owns the triangles it writes (the callers arbitrate with an atomic min per triangle), and no step puts a
// whole cell behind one lane except the pocket scan (2 R steps).  The phases are separate functions so that the device
// kernel (lin2d.hpp: k_lin2d_build_rows) and the host emulation (tests/native/delaunay_host.cpp) share them.

// phase 0 (one lane): the non-empty rows ne[0, m) and the first triangle of every strip sbase[0, m).  Returns m, or -1 when
// the rows do not support the construction (fewer than two non-empty rows, or a non-empty row of a single node: its strip
// edge would bound both pockets) — the sweep of dt_build serves those tables.
XS_HD int dtp_plan(const int *rowb, int R, uint16_t *ne, int *sbase)
{
    int m = 0;
    for (int r = 0; r < R; r++) {
        const int cnt = rowb[r + 1] - rowb[r];
        if (cnt == 1) return -1;
        if (cnt > 0) ne[m++] = (uint16_t)r;
    }
    if (m < 2) return -1;
    int t = 0;
    for (int s2 = 0; s2 + 1 < m; s2++) {
        sbase[s2] = t;
        t += (rowb[ne[s2] + 1] - rowb[ne[s2]]) + (rowb[ne[s2 + 1] + 1] - rowb[ne[s2 + 1]]) - 2;
    }
    sbase[m - 1] = t;
    return m;
}

// phase 1 (a lane per strip s2 in [0, m - 1)): the strip between rows A = ne[s2] (below) and B = ne[s2 + 1] zipped from the
// left; of the two possible next triangles the one whose circumcircle is free of the other candidate.  up[i] / dn[i]: the
// triangle above / below the row edge (i, i + 1) (both preset to DT_NONE by the caller).
XS_HD void dtp_zip(Tri2D &d, const int *rowb, const uint16_t *ne, const int *sbase, int s2, uint16_t *up, uint16_t *dn)
{
    const int A = ne[s2], B = ne[s2 + 1];
    const int a0 = rowb[A], p = rowb[A + 1] - a0, b0 = rowb[B], q = rowb[B + 1] - b0;
    const int t_first = sbase[s2], t_end = sbase[s2 + 1];
    int i = 0, j = 0;
    for (int t = t_first; t < t_end; t++) {
        bool adv_a;
        if (i + 1 >= p) adv_a = false;
        else if (j + 1 >= q) adv_a = true;
        else adv_a = !(dt_incircle(d, a0 + i, a0 + i + 1, b0 + j, b0 + j + 1) > 0.0);
        const int prev = t > t_first ? t - 1 : DT_NONE, next = t + 1 < t_end ? t + 1 : DT_NONE;
        if (adv_a) {  // (a_i, a_i+1, b_j): opposite a_i the right diagonal, opposite a_i+1 the left one, opposite b_j the row edge
            dt_set(d, t, a0 + i, a0 + i + 1, b0 + j, next, prev, DT_NONE);
            up[a0 + i] = (uint16_t)t;
            i++;
        } else {      // (a_i, b_j+1, b_j): opposite a_i the row edge, opposite b_j+1 the left diagonal, opposite b_j the right one
            dt_set(d, t, a0 + i, b0 + j + 1, b0 + j, DT_NONE, prev, next);
            dn[b0 + j] = (uint16_t)t;
            j++;
        }
    }
}

// phase 2 (a lane per strip triangle): the neighbour across its row edge, from the tables of phase 1
XS_HD void dtp_link(Tri2D &d, int t, const uint16_t *up, const uint16_t *dn)
{
    const int v0 = d.tv[3 * t], v1 = d.tv[3 * t + 1], v2 = d.tv[3 * t + 2];
    if (d.py[v1] == d.py[v0]) d.tn[3 * t + 2] = dn[v0];  // row edge (a_i, a_i+1) below the triangle
    else d.tn[3 * t] = up[v2];                            // row edge (b_j, b_j+1) above it
}

// phase 3 (one lane): the pockets left of the rows' first nodes and right of their last ones (a monotone-chain scan per side:
// every pop is a triangle).  stk / own_t / own_s: work arrays of m entries.  Returns the number of triangles.
XS_HD int dtp_pockets(Tri2D &d, const int *rowb, const uint16_t *ne, int m, const int *sbase, uint16_t *stk, uint16_t *own_t,
                      uint16_t *own_s)
{
    int ntri = sbase[m - 1];
    for (int side = 0; side < 2; side++) {
        int sp = 0;
        for (int k = 0; k < m; k++) {
            const int P = side == 0 ? rowb[ne[k]] : rowb[ne[k] + 1] - 1;
            int cur_t = DT_NONE, cur_s = 0;  // the triangle on the edge (stk[sp - 1], P) and its slot there
            if (k > 0) {
                if (side == 0) { cur_t = sbase[k - 1]; cur_s = 1; }
                else {
                    cur_t = sbase[k] - 1;  // the strip's last triangle: its right diagonal
                    cur_s = d.py[d.tv[3 * cur_t + 1]] == d.py[d.tv[3 * cur_t]] ? 0 : 2;
                }
            }
            while (sp >= 2) {
                const int A = stk[sp - 2], B = stk[sp - 1];
                const double o = dt_orient(d, A, B, P);
                if (!(side == 0 ? o > 0.0 : o < 0.0)) break;
                const int T = ntri++;
                const int ab_t = own_t[sp - 2], ab_s = own_s[sp - 2];
                if (side == 0) dt_set(d, T, A, B, P, cur_t, DT_NONE, ab_t);  // new hull edge (P, A): opposite B
                else dt_set(d, T, A, P, B, cur_t, ab_t, DT_NONE);            // new hull edge (A, P): opposite B
                d.tn[3 * cur_t + cur_s] = (uint16_t)T;
                d.tn[3 * ab_t + ab_s] = (uint16_t)T;
                cur_t = T;
                cur_s = side == 0 ? 1 : 2;
                sp--;
            }
            if (sp >= 1) { own_t[sp - 1] = (uint16_t)cur_t; own_s[sp - 1] = (uint16_t)cur_s; }
            stk[sp++] = (uint16_t)P;
        }
    }
    d.ntri = ntri;
    return ntri;
}

// phase 4, per round and per (triangle t, slot i) with a neighbour across the edge opposite slot i: does the edge want a flip,
// and which triangles would the flip write — t, u and the two outer neighbours whose back-pointers move (DT_NONE where there is
// none).  The caller tests an edge once per round and lets a flip happen when it holds all four.
XS_HD bool dtp_flip_sites(const Tri2D &d, int t, int i, int *sites)
{
    const int u = d.tn[3 * t + i];
    if (u == DT_NONE) return false;
    const int j = dt_slot_of(d, u, t);
    const int a = d.tv[3 * t + i], b = d.tv[3 * t + (i + 1) % 3], c = d.tv[3 * t + (i + 2) % 3];
    const int q = d.tv[3 * u + j];
    if (!(dt_incircle(d, a, b, c, q) > 0.0)) return false;
    if (!(dt_orient(d, a, b, q) > 0.0 && dt_orient(d, a, q, c) > 0.0)) return false;  // (rounding: the quadrilateral must stay convex)
    sites[0] = t; sites[1] = u;
    sites[2] = d.tn[3 * u + (j + 1) % 3];
    sites[3] = d.tn[3 * t + (i + 1) % 3];
    return true;
}

// the flip itself (dt_legalize's): t = (a, b, c), u = (q, c, b)  ->  t = (a, b, q), u = (a, q, c)
XS_HD void dtp_flip_apply(Tri2D &d, int t, int i)
{
    const int u = d.tn[3 * t + i];
    const int j = dt_slot_of(d, u, t);
    const int a = d.tv[3 * t + i], b = d.tv[3 * t + (i + 1) % 3], c = d.tv[3 * t + (i + 2) % 3];
    const int q = d.tv[3 * u + j];
    const int X = d.tn[3 * u + (j + 1) % 3], Y = d.tn[3 * u + (j + 2) % 3];
    const int Z = d.tn[3 * t + (i + 1) % 3], W = d.tn[3 * t + (i + 2) % 3];
    dt_set(d, t, a, b, q, X, u, W);
    dt_set(d, u, a, q, c, Y, Z, t);
    if (X != DT_NONE) d.tn[3 * X + dt_slot_of(d, X, u)] = (uint16_t)t;
    if (Z != DT_NONE) d.tn[3 * Z + dt_slot_of(d, Z, t)] = (uint16_t)u;
}

// Triangle containing (qx, qy), walking from `start`; -1 outside the hull.  Points within a relative tolerance of an
// edge count as inside (scipy's _find_simplex works with eps = 100 * DBL_EPSILON on the barycentric coordinates).
XS_HD int dt_locate(const Tri2D &d, double qx, double qy, int start)
{
    if (d.ntri == 0) return -1;
    int t = start >= 0 && start < d.ntri ? start : 0;
    int from = -1;
    for (int steps = 0; steps < 2 * d.ntri + 8; steps++) {
        // the three corners are fetched once per triangle (not once per edge test)
        const int v[3] = {d.tv[3 * t], d.tv[3 * t + 1], d.tv[3 * t + 2]};
        const double vx[3] = {d.px[v[0]], d.px[v[1]], d.px[v[2]]}, vy[3] = {d.py[v[0]], d.py[v[1]], d.py[v[2]]};
        int go = -1;
        bool moved = false;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int ia = (k + 1) % 3, ib = (k + 2) % 3;
            const int nb = d.tn[3 * t + k];
            if (nb == from && from >= 0) continue;
            const double o = (vx[ib] - vx[ia]) * (qy - vy[ia]) - (vy[ib] - vy[ia]) * (qx - vx[ia]);  // dt_orient_q(a, b, q)
            if (o < 0.0) {
                if (nb == DT_NONE) {
                    // a hull edge: only a point clearly beyond it is outside (the tolerance is evaluated here alone)
                    const double scale = (fabs(vx[ib] - vx[ia]) + fabs(vy[ib] - vy[ia])) * (fabs(qx - vx[ia]) + fabs(qy - vy[ia]) + 1.0);
                    if (o < -2.3e-14 * scale) return -1;
                    continue;
                }
                go = nb;
                moved = true;
                break;
            }
        }
        if (!moved) return t;
        from = t;
        t = go;
    }
    return -1;
}

// value of the linear interpolant on triangle t at (qx, qy)
XS_HD double dt_eval(const Tri2D &d, const double *val, int t, double qx, double qy)
{
    const int a = d.tv[3 * t], b = d.tv[3 * t + 1], c = d.tv[3 * t + 2];
    const double m00 = d.px[a] - d.px[c], m01 = d.px[b] - d.px[c], m10 = d.py[a] - d.py[c], m11 = d.py[b] - d.py[c];
    const double det = m00 * m11 - m01 * m10;
    const double rx = qx - d.px[c], ry = qy - d.py[c];
    const double c0 = (m11 * rx - m01 * ry) / det, c1 = (m00 * ry - m10 * rx) / det;
    const double c2 = 1.0 - c0 - c1;
    return c0 * val[a] + c1 * val[b] + c2 * val[c];
}

// ---------------------------------------------------------------------------------------------------------------------
// griddata(method="cubic") = scipy.interpolate.CloughTocher2DInterpolator(points, values, tol=1e-6, maxiter=400) — what the
// reference's grouped interp="cubic" evaluates (utils._interp_on_quantiles_2d, utils.py:391-396).  Its published algorithm
// (scipy/interpolate/interpnd.pyx, restated from the paper trail it cites: Nielson 1983 / Renka & Cline 1984 for the
// gradients, Alfeld 1984 / Farin 1986 for the element): (1) the gradient at every node minimises the curvature functional
// of the cubic Hermite edge curves, solved by Gauss-Seidel sweeps over the nodes in index order until the largest relative
// change of a sweep is below tol; (2) on every triangle the C1 Clough-Tocher element in Bezier form, the cross-boundary
// derivative taken along the direction to the centroid of the neighbouring triangle (affine invariant).

// vertex adjacency in CSR form from the triangle list.  adj_ptr [n + 1], adj_idx [sum of degrees <= 6 n], fill [n] scratch.
XS_HD void dt_adjacency(const Tri2D &d, int *adj_ptr, uint16_t *adj_idx, uint16_t *fill)
{
    const int n = d.n;
    for (int v = 0; v <= n; v++) adj_ptr[v] = 0;
    // degree = incident triangles, + 1 on the hull (every hull vertex starts exactly one counter-clockwise hull edge)
    for (int t = 0; t < d.ntri; t++)
        for (int k = 0; k < 3; k++) {
            adj_ptr[d.tv[3 * t + k] + 1]++;
            if (d.tn[3 * t + k] == DT_NONE) adj_ptr[d.tv[3 * t + (k + 1) % 3] + 1]++;
        }
    for (int v = 0; v < n; v++) { adj_ptr[v + 1] += adj_ptr[v]; fill[v] = 0; }
    for (int t = 0; t < d.ntri; t++)
        for (int k = 0; k < 3; k++) {
            const int a = d.tv[3 * t + k];
            for (int j = 1; j <= 2; j++) {
                const uint16_t b = d.tv[3 * t + (k + j) % 3];
                uint16_t *l = adj_idx + adj_ptr[a];
                const int cap = adj_ptr[a + 1] - adj_ptr[a];
                int f = fill[a], i = 0;
                while (i < f && l[i] != b) i++;
                if (i == f && f < cap) { l[f] = b; fill[a] = (uint16_t)(f + 1); }
            }
        }
}

// Gauss-Seidel estimate of the nodal gradients grad[2 n] (interpnd._estimate_gradients_2d_global)
XS_HD int dt_gradients(const Tri2D &d, const double *val, const int *adj_ptr, const uint16_t *adj_idx, double *grad,
                       int maxiter = 400, double tol = 1e-6)
{
    const int n = d.n;
    for (int i = 0; i < 2 * n; i++) grad[i] = 0.0;
    for (int iter = 0; iter < maxiter; iter++) {
        double err = 0.0;
        for (int ip = 0; ip < n; ip++) {
            if (adj_ptr[ip + 1] == adj_ptr[ip]) continue;  // a node outside the triangulation
            double Q0 = 0.0, Q1 = 0.0, Q3 = 0.0, s0 = 0.0, s1 = 0.0;
            const double f1 = val[ip];
            for (int j = adj_ptr[ip]; j < adj_ptr[ip + 1]; j++) {
                const int ip2 = adj_idx[j];
                const double ex = d.px[ip2] - d.px[ip], ey = d.py[ip2] - d.py[ip];
                const double L = sqrt(ex * ex + ey * ey), L3 = L * L * L;
                const double f2 = val[ip2];
                const double df2 = -ex * grad[2 * ip2] - ey * grad[2 * ip2 + 1];
                Q0 += 4.0 * ex * ex / L3;
                Q1 += 4.0 * ex * ey / L3;
                Q3 += 4.0 * ey * ey / L3;
                s0 += (6.0 * (f1 - f2) - 2.0 * df2) * ex / L3;
                s1 += (6.0 * (f1 - f2) - 2.0 * df2) * ey / L3;
            }
            const double det = Q0 * Q3 - Q1 * Q1;
            const double r0 = (Q3 * s0 - Q1 * s1) / det, r1 = (-Q1 * s0 + Q0 * s1) / det;
            double change = fmax(fabs(grad[2 * ip] + r0), fabs(grad[2 * ip + 1] + r1));
            grad[2 * ip] = -r0;
            grad[2 * ip + 1] = -r1;
            change /= fmax(1.0, fmax(fabs(r0), fabs(r1)));
            err = fmax(err, change);
        }
        if (err < tol) return iter + 1;
    }
    return 0;  // not converged: the last sweep's values are used (scipy warns and does the same)
}

// barycentric coordinates of (qx, qy) in triangle t (c[2] belongs to the third vertex)
XS_HD void dt_bary(const Tri2D &d, int t, double qx, double qy, double *c)
{
    const int a = d.tv[3 * t], b = d.tv[3 * t + 1], cc = d.tv[3 * t + 2];
    const double m00 = d.px[a] - d.px[cc], m01 = d.px[b] - d.px[cc], m10 = d.py[a] - d.py[cc], m11 = d.py[b] - d.py[cc];
    const double det = m00 * m11 - m01 * m10;
    const double rx = qx - d.px[cc], ry = qy - d.py[cc];
    c[0] = (m11 * rx - m01 * ry) / det;
    c[1] = (m00 * ry - m10 * rx) / det;
    c[2] = 1.0 - c[0] - c[1];
}

// value of the Clough-Tocher interpolant on triangle t at (qx, qy) (interpnd._clough_tocher_2d_single)
XS_HD double dt_eval_ct(const Tri2D &d, const double *val, const double *grad, int t, double qx, double qy)
{
    const int v0 = d.tv[3 * t], v1 = d.tv[3 * t + 1], v2 = d.tv[3 * t + 2];
    double b[3];
    dt_bary(d, t, qx, qy, b);
    const double e12x = d.px[v1] - d.px[v0], e12y = d.py[v1] - d.py[v0];
    const double e23x = d.px[v2] - d.px[v1], e23y = d.py[v2] - d.py[v1];
    const double e31x = d.px[v0] - d.px[v2], e31y = d.py[v0] - d.py[v2];
    const double f1 = val[v0], f2 = val[v1], f3 = val[v2];
    const double df12 = +(grad[2 * v0] * e12x + grad[2 * v0 + 1] * e12y);
    const double df21 = -(grad[2 * v1] * e12x + grad[2 * v1 + 1] * e12y);
    const double df23 = +(grad[2 * v1] * e23x + grad[2 * v1 + 1] * e23y);
    const double df32 = -(grad[2 * v2] * e23x + grad[2 * v2 + 1] * e23y);
    const double df31 = +(grad[2 * v2] * e31x + grad[2 * v2 + 1] * e31y);
    const double df13 = -(grad[2 * v0] * e31x + grad[2 * v0 + 1] * e31y);
    const double c3000 = f1, c2100 = (df12 + 3 * c3000) / 3, c2010 = (df13 + 3 * c3000) / 3;
    const double c0300 = f2, c1200 = (df21 + 3 * c0300) / 3, c0210 = (df23 + 3 * c0300) / 3;
    const double c0030 = f3, c1020 = (df31 + 3 * c0030) / 3, c0120 = (df32 + 3 * c0030) / 3;
    const double c2001 = (c2100 + c2010 + c3000) / 3, c0201 = (c1200 + c0300 + c0210) / 3, c0021 = (c1020 + c0120 + c0030) / 3;
    double g[3];
    for (int k = 0; k < 3; k++) {
        const int u = d.tn[3 * t + k];
        if (u == DT_NONE) { g[k] = -0.5; continue; }
        const int u0 = d.tv[3 * u], u1 = d.tv[3 * u + 1], u2 = d.tv[3 * u + 2];
        double c[3];
        dt_bary(d, t, (d.px[u0] + d.px[u1] + d.px[u2]) / 3, (d.py[u0] + d.py[u1] + d.py[u2]) / 3, c);
        if (k == 0) g[k] = (2 * c[2] + c[1] - 1) / (2 - 3 * c[2] - 3 * c[1]);
        else if (k == 1) g[k] = (2 * c[0] + c[2] - 1) / (2 - 3 * c[0] - 3 * c[2]);
        else g[k] = (2 * c[1] + c[0] - 1) / (2 - 3 * c[1] - 3 * c[0]);
    }
    const double c0111 = (g[0] * (-c0300 + 3 * c0210 - 3 * c0120 + c0030) + (-c0300 + 2 * c0210 - c0120 + c0021 + c0201)) / 2;
    const double c1011 = (g[1] * (-c0030 + 3 * c1020 - 3 * c2010 + c3000) + (-c0030 + 2 * c1020 - c2010 + c2001 + c0021)) / 2;
    const double c1101 = (g[2] * (-c3000 + 3 * c2100 - 3 * c1200 + c0300) + (-c3000 + 2 * c2100 - c1200 + c2001 + c0201)) / 2;
    const double c1002 = (c1101 + c1011 + c2001) / 3, c0102 = (c1101 + c0111 + c0201) / 3, c0012 = (c1011 + c0111 + c0021) / 3;
    const double c0003 = (c1002 + c0102 + c0012) / 3;
    double mn = b[0];
    if (b[1] < mn) mn = b[1];
    if (b[2] < mn) mn = b[2];
    const double b1 = b[0] - mn, b2 = b[1] - mn, b3 = b[2] - mn, b4 = 3 * mn;
    return b1 * b1 * b1 * c3000 + 3 * b1 * b1 * b2 * c2100 + 3 * b1 * b1 * b3 * c2010 + 3 * b1 * b1 * b4 * c2001 +
           3 * b1 * b2 * b2 * c1200 + 6 * b1 * b2 * b4 * c1101 + 3 * b1 * b3 * b3 * c1020 + 6 * b1 * b3 * b4 * c1011 +
           3 * b1 * b4 * b4 * c1002 + b2 * b2 * b2 * c0300 + 3 * b2 * b2 * b3 * c0210 + 3 * b2 * b2 * b4 * c0201 +
           3 * b2 * b3 * b3 * c0120 + 6 * b2 * b3 * b4 * c0111 + 3 * b2 * b4 * b4 * c0102 + b3 * b3 * b3 * c0030 +
           3 * b3 * b3 * b4 * c0021 + 3 * b3 * b4 * b4 * c0012 + b4 * b4 * b4 * c0003;
}

}  // namespace xs
