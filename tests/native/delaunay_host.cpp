// Host build of xsdba_amd/csrc/delaunay.hpp for tests/test_delaunay_host.py (checked against scipy there).
// Test infrastructure only: the product runs this header on the GPU (k_lin2d_build / k_lin2d_adjust).
#include <math.h>
#include <stdint.h>
#include <algorithm>
#include <vector>
#define XS_HD inline
#include "../../xsdba_amd/csrc/delaunay.hpp"

extern "C" int dt_host_build(int n, const double *px, const double *py, uint16_t *tv, uint16_t *tn, int row_order)
{
    std::vector<uint16_t> hn(n), hp(n), ht(n), ord(n);
    xs::Tri2D d{px, py, tv, tn, hn.data(), hp.data(), ht.data(), n, 0};
    if (row_order) {  // the storage order (row by row) is a valid sweep too
        for (int i = 0; i < n; i++) ord[i] = (uint16_t)i;
    } else {
        xs::dt_sort_order(d, ord.data());
        for (int i = 0; i + 1 < n; i++)
            if (xs::dt_before(d, ord[i + 1], ord[i])) return -1;  // the order must ascend
    }
    return xs::dt_build(d, ord.data());
}

// the row-parallel construction (dtp_*), its lanes emulated one after the other; rowb[R + 1]: first node of every row.
// Returns the number of triangles (-1: the rows do not support it), *rounds = flip rounds taken.
extern "C" int dt_host_build_rows(int n, const double *px, const double *py, uint16_t *tv, uint16_t *tn, int R, const int *rowb, int *rounds)
{
    std::vector<uint16_t> up(n, xs::DT_NONE), dn(n, xs::DT_NONE), ne(R), stk(R), own_t(R), own_s(R);
    std::vector<int> sbase(R + 1);
    xs::Tri2D d{px, py, tv, tn, nullptr, nullptr, nullptr, n, 0};
    const int m = xs::dtp_plan(rowb, R, ne.data(), sbase.data());
    if (m < 0) return -1;
    for (int s = 0; s + 1 < m; s++) xs::dtp_zip(d, rowb, ne.data(), sbase.data(), s, up.data(), dn.data());
    for (int t = 0; t < sbase[m - 1]; t++) xs::dtp_link(d, t, up.data(), dn.data());
    const int ntri = xs::dtp_pockets(d, rowb, ne.data(), m, sbase.data(), stk.data(), own_t.data(), own_s.data());
    // the rounds of k_lin2d_build_rows: (A) the edges of the triangles that changed in the previous round (all of them in the first)
    // are tested — an edge between two such triangles from the lower-numbered one —, a wanted flip records the triangles it writes
    // and bids for them (atomic min of a round-stamped edge number: later rounds underbid stale bids), (B) a flip that holds all its
    // triangles is applied — nobody else reads or writes those — and its two triangles (a loser: its first) go on the next list
    std::vector<uint32_t> lock(ntri, 0xFFFFFFFFu);
    struct Cand { uint32_t bid; int t, i, sites[4]; };
    std::vector<Cand> cand;
    std::vector<int> cur(ntri), next;
    std::vector<char> in_cur(ntri, 1), in_next(ntri, 0);
    for (int t = 0; t < ntri; t++) cur[t] = t;
    const int MAXR = 8 * n + 64;
    int r = 0;
    for (; r < MAXR; r++) {
        cand.clear();
        for (int t : cur)
            for (int i = 0; i < 3; i++) {
                const int u = tn[3 * t + i];
                if (u == xs::DT_NONE || (in_cur[u] && u < t)) continue;
                Cand c{((uint32_t)(16383 - (r & 16383)) << 18) | (uint32_t)(3 * t + i), t, i, {0, 0, 0, 0}};
                if (!xs::dtp_flip_sites(d, t, i, c.sites)) continue;
                for (int k = 0; k < 4; k++)
                    if (c.sites[k] != xs::DT_NONE) lock[c.sites[k]] = std::min(lock[c.sites[k]], c.bid);
                cand.push_back(c);
            }
        if (cand.empty()) break;
        next.clear();
        auto mark = [&](int t) { if (!in_next[t]) { in_next[t] = 1; next.push_back(t); } };
        for (const Cand &c : cand) {
            bool mine = true;
            for (int k = 0; k < 4; k++) mine = mine && (c.sites[k] == xs::DT_NONE || lock[c.sites[k]] == c.bid);
            if (mine) { xs::dtp_flip_apply(d, c.t, c.i); mark(c.sites[1]); }
            mark(c.t);
        }
        for (int t : cur) in_cur[t] = 0;
        for (int t : next) { in_cur[t] = 1; in_next[t] = 0; }
        cur.swap(next);
    }
    if (rounds) *rounds = r;
    return ntri;
}

extern "C" void dt_host_eval(int n, const double *px, const double *py, const double *val, int ntri, uint16_t *tv,
                             uint16_t *tn, int nq, const double *qx, const double *qy, double *out)
{
    xs::Tri2D d{px, py, tv, tn, nullptr, nullptr, nullptr, n, ntri};
    int t = 0;
    for (int i = 0; i < nq; i++) {
        const int f = xs::dt_locate(d, qx[i], qy[i], t);
        out[i] = f < 0 ? NAN : xs::dt_eval(d, val, f, qx[i], qy[i]);
        if (f >= 0) t = f;
    }
}

// griddata(method="cubic"): nodal gradients + the Clough-Tocher element
extern "C" int dt_host_eval_ct(int n, const double *px, const double *py, const double *val, int ntri, uint16_t *tv,
                               uint16_t *tn, int nq, const double *qx, const double *qy, double *out, double *grad)
{
    xs::Tri2D d{px, py, tv, tn, nullptr, nullptr, nullptr, n, ntri};
    std::vector<int> ap(n + 1);
    std::vector<uint16_t> ai(6 * (size_t)n + 8), fill(n);
    xs::dt_adjacency(d, ap.data(), ai.data(), fill.data());
    const int iters = xs::dt_gradients(d, val, ap.data(), ai.data(), grad);
    int t = 0;
    for (int i = 0; i < nq; i++) {
        const int f = xs::dt_locate(d, qx[i], qy[i], t);
        out[i] = f < 0 ? NAN : xs::dt_eval_ct(d, val, grad, f, qx[i], qy[i]);
        if (f >= 0) t = f;
    }
    return iters;
}
