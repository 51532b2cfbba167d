// Host build of xsdba_amd/csrc/delaunay.hpp for tests/test_delaunay_host.py (checked against scipy there).
// Test infrastructure only: the product runs this header on the GPU (k_lin2d_build / k_lin2d_adjust).
#include <math.h>
#include <stdint.h>
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
