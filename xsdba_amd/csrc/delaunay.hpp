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

// Triangle containing (qx, qy), walking from `start`; -1 outside the hull.  Points within a relative tolerance of an
// edge count as inside (scipy's _find_simplex works with eps = 100 * DBL_EPSILON on the barycentric coordinates).
XS_HD int dt_locate(const Tri2D &d, double qx, double qy, int start)
{
    if (d.ntri == 0) return -1;
    int t = start >= 0 && start < d.ntri ? start : 0;
    int from = -1;
    for (int steps = 0; steps < 2 * d.ntri + 8; steps++) {
        int go = -1;
        bool moved = false;
        for (int k = 0; k < 3; k++) {
            const int a = d.tv[3 * t + (k + 1) % 3], b = d.tv[3 * t + (k + 2) % 3];
            const int nb = d.tn[3 * t + k];
            if (nb == from && from >= 0) continue;
            const double o = dt_orient_q(d, a, b, qx, qy);
            if (o < 0.0) {
                if (nb == DT_NONE) {
                    // a hull edge: only a point clearly beyond it is outside (the tolerance is evaluated here alone)
                    const double scale = (fabs(d.px[b] - d.px[a]) + fabs(d.py[b] - d.py[a])) * (fabs(qx - d.px[a]) + fabs(qy - d.py[a]) + 1.0);
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

}  // namespace xs
