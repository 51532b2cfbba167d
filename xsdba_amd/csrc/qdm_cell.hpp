// qdm_cell.hpp — fast QDM adjust / rank kernel: one 1024-thread workgroup per grid cell, float32.
//
// k_qdm_adjust (kernels.hpp) ranks every element by counting over its ranking row: O(T * row length) per cell,
// fine for exact day-of-year groups (30 members) and hopeless for coarse groups (group="time": T^2) or windowed
// ranking rows (rank_window: 31 x more).  This kernel reuses the k_train_cell machinery instead:
//   A+B. series -> keys in LDS, ONE radix sort of the time indexes, rank[t]                    (cell_load_sort)
//   C.   each wave walks a contiguous chunk of groups with a membership bit-vector of the ranking row IN RANK
//        SPACE (built once per chunk, then XOR toggle lists), plus a per-word prefix popcount table.  For an
//        element with global rank r whose equal-value run is [rs, re):  less = members below rs, leq = members
//        below re (two popcount-prefix lookups), so its average-tie rank is (less + leq + 1) / 2 — exact,
//        utils.rank (utils.py:573-638) with bottleneck/scipy 'average' ties.  Minimum / maximum tie counts of the
//        group (the pmin / pmax rescaling) come from the first / last set bit the same way.
//        2 * rank = less + leq + 1 is parked per time step in the u16 buffer the sort no longer needs.
//   D.   one coalesced pass over time: percent rank sim_q (float64), interpolation of af on the quantile axis
//        (_adjustment.py:783-886, utils.interp_on_quantiles), apply, store scen / sim_q.
// With fewer groups than waves every group is shared by several waves (each keeps its own copy of the vector and
// ranks a slice of the members); a single all-time group ranks straight from the sorted order.
//
// Grouped interpolation: the abscissae are the quantile nodes for every group, so the exact 2-D nearest
// neighbour of (sim_q, group) is the nearest node of the element's OWN row as long as that node's factor is
// not NaN (any other row is >= 1 away, nodes of one row span <= 1).  Cells where that argument fails (a NaN
// factor is met, NaN / too widely spread nodes) are appended to `flag_list` and redone by k_qdm_adjust.
#pragma once
#include "train_cell.hpp"

namespace xs {

template <int EPT> struct QdmCfg {
    using B = CellCfg<EPT>;
    static constexpr int GMAX = 512, QMAX = 64;
    static constexpr size_t off_bits = B::off_rank + (size_t)B::TPAD * 2;          // per wave: BITW dwords
    static constexpr size_t off_pw = off_bits + (size_t)B::NWAVE * B::BITW * 4;     // per wave: BITW u16 prefix counts
    static constexpr size_t sel_end = off_pw + (size_t)B::NWAVE * B::BITW * 2;
    static constexpr size_t off_gstat = ((B::sort_end > sel_end ? B::sort_end : sel_end) + 15) & ~(size_t)15;
    static constexpr size_t off_tab = off_gstat + (size_t)GMAX * 3 * 8;             // N, pmin, pmax per group (float64)
    static constexpr size_t off_red = off_tab + (size_t)2 * QMAX * 4 + 32;          // nodes, 1-D table, its metadata
    static constexpr size_t lds_bytes = off_red + 64 * 8;
};

template <int EPT>
__global__ __launch_bounds__(1024) void k_qdm_cell(
    const float *__restrict__ sim, int64_t T, int64_t C, int64_t sc, const float *__restrict__ af,
    const float *__restrict__ quantiles, int G, int Q, const int *__restrict__ gidx,
    const int *__restrict__ grp_ptr, const int *__restrict__ grp_idx,  // ranking rows (windowed rows or exact groups)
    const int *__restrict__ mem_ptr, const int *__restrict__ mem_idx,  // exact groups: the elements each row ranks
    const int *__restrict__ tog_ptr, const int *__restrict__ tog_idx, const int *__restrict__ tog64,
    int use_tog,  // consecutive ranking rows differ by the toggle lists; 0: rows are the (disjoint) exact groups
    int interp, int extrap, int kind, float *__restrict__ scen, double *__restrict__ sim_q,
    int *__restrict__ flag_count, int *__restrict__ flag_list)
{
    using Cfg = CellCfg<EPT>;
    using QC = QdmCfg<EPT>;
    constexpr int NT = Cfg::NT, TPAD = Cfg::TPAD, WPL = Cfg::WPL, BITW = Cfg::BITW, NWAVE = Cfg::NWAVE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t *keys = reinterpret_cast<const uint32_t *>(smem + Cfg::off_keys);
    uint16_t *idx0 = reinterpret_cast<uint16_t *>(smem + Cfg::off_idx0);
    uint16_t *idx1 = reinterpret_cast<uint16_t *>(smem + Cfg::off_idx1);
    const uint16_t *rank = reinterpret_cast<const uint16_t *>(smem + Cfg::off_rank);
    double *gstat = reinterpret_cast<double *>(smem + QC::off_gstat);  // per group: N, pmin, pmax of the percent-rank rescaling
    float *qn = reinterpret_cast<float *>(smem + QC::off_tab);  // quantile nodes (grouped) / compacted nodes (1-D)
    float *ty = qn + QC::QMAX;                                  // compacted factors of the 1-D table
    int *tmeta = reinterpret_cast<int *>(ty + QC::QMAX);        // [0] pairs, [1] yfirst, [2] ylast (bit patterns), [3] bad nodes
    uint32_t *red = reinterpret_cast<uint32_t *>(smem + QC::off_red);

    const int tid = threadIdx.x, wid = tid >> 6;
    uint32_t *bits = reinterpret_cast<uint32_t *>(smem + QC::off_bits) + wid * BITW;
    uint16_t *pw = reinterpret_cast<uint16_t *>(smem + QC::off_pw) + wid * BITW;

    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        int lane = tid & 63;  // opaque per cell: lane-derived addresses are recomputed, not spilled across the sort
        __asm__ volatile("" : "+v"(lane));
        int nvalid;
        const uint16_t *sorted = cell_load_sort<EPT>(sim + c * sc, T, smem, red, 0, nvalid);
        uint16_t *r2buf = (sorted == idx1) ? idx0 : idx1;  // 2 * average rank per time step, 0 = no rank
        auto val = [&](int r) { return from_key(keys[sorted[r]]); };
        // first rank in [lo, hi] whose value is not below v (v occurs at hi)
        auto run_begin = [&](float v, int lo, int hi) {
            while (lo < hi) { int mid = (lo + hi) >> 1; if (val(mid) < v) lo = mid + 1; else hi = mid; }
            return lo;
        };
        // first rank in [lo, hi) whose value is above v, hi if none
        auto run_end = [&](float v, int lo, int hi) {
            while (lo < hi) { int mid = (lo + hi) >> 1; if (val(mid) > v) hi = mid; else lo = mid + 1; }
            return lo;
        };
        // extreme percent ranks of a group (utils.rank :631-635: mn / mx of rank / N with average ties), once per group
        auto put_stats = [&](int g, int n, int nmin, int nmax) {
            const double N = (double)n;
            gstat[3 * g] = N;
            gstat[3 * g + 1] = (((double)nmin + 1.0) / 2.0) / N;
            gstat[3 * g + 2] = ((N - (double)nmax) + ((double)nmax + 1.0) / 2.0) / N;
        };
        for (int i = tid; i < TPAD; i += NT) r2buf[i] = 0;
        // interpolation table: the nodes (grouped) or the compacted (node, factor) pairs of the only row (1-D)
        if (af && wid == 0) {
            const float xq = lane < Q ? quantiles[lane] : __builtin_nanf("");
            if (G > 1) {
                if (lane < Q) qn[lane] = xq;
                const unsigned long long bad = __ballot(lane < Q && xq != xq);
                const float xlo = __shfl(xq, 0, 64), xhi = __shfl(xq, Q - 1, 64);
                if (lane == 0) tmeta[3] = (bad != 0ull) || !(xhi - xlo <= 1.0f);
            } else {
                const float yq = lane < Q ? af[(size_t)c * Q + lane] : __builtin_nanf("");
                const unsigned long long bx = __ballot(xq == xq), by = __ballot(yq == yq), bxy = bx & by;
                const int pos = __popcll(bxy & ((1ull << lane) - 1ull));
                if ((bxy >> lane) & 1ull) { qn[pos] = xq; ty[pos] = yq; }
                const float yf = __shfl(yq, by ? __ffsll((long long)by) - 1 : 63, 64);
                const float yl = __shfl(yq, by ? 63 - __clzll((long long)by) : 63, 64);
                if (lane == 0) { tmeta[0] = __popcll(bxy); tmeta[1] = __float_as_int(yf); tmeta[2] = __float_as_int(yl); tmeta[3] = 0; }
            }
        }
        __syncthreads();

        const bool all_time = G == 1 && grp_ptr[1] - grp_ptr[0] == (int)T && mem_ptr[1] - mem_ptr[0] == (int)T;
        if (all_time) {
            // ---------------- C0. one group holding every time step: the sorted order is the ranking ----------------
            for (int r = tid; r < nvalid; r += NT) {
                const float v = val(r);
                int rs = r, re = r + 1;
                if (r > 0 && val(r - 1) == v) rs = run_begin(v, 0, r - 1);
                if (re < nvalid && val(re) == v) re = run_end(v, re + 1, nvalid);
                r2buf[sorted[r]] = (uint16_t)(rs + re + 1);
            }
            if (tid == 0 && nvalid) {
                put_stats(0, nvalid, run_end(val(0), 1, nvalid), nvalid - run_begin(val(nvalid - 1), 0, nvalid - 1));
            }
        } else if (nvalid) {
            // ---------------- C. membership bit-vectors of the ranking rows in rank space ----------------
            int g0, g1, slice = 0, nslice = 1;
            if (G >= NWAVE) {
                const int gpc = (G + NWAVE - 1) / NWAVE;
                g0 = min(G, wid * gpc); g1 = min(G, g0 + gpc);
            } else {
                const int wpg = NWAVE / G;  // waves sharing one group
                const int g = wid / wpg;
                g0 = min(G, g); g1 = min(G, g + 1);
                slice = wid % wpg; nslice = wpg;
            }
            for (int i = lane; i < BITW; i += 64) bits[i] = 0u;
            lds_wave_fence();
            auto flip_list = [&](const int *__restrict__ l, int len) {
                for (int i0 = 0; i0 < len; i0 += 256) {
                    int ts[4], rs[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) { int i = i0 + 64 * u + lane; ts[u] = i < len ? l[i] : -1; }
#pragma unroll
                    for (int u = 0; u < 4; u++) rs[u] = ts[u] >= 0 ? (int)rank[ts[u]] : 0x7FFFFFFF;
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        if (rs[u] < nvalid)
                            (void)__hip_atomic_fetch_xor(&bits[rs[u] >> 5], 1u << (rs[u] & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            };
            for (int g = g0; g < g1; g++) {
                if (g == g0) flip_list(grp_idx + grp_ptr[g], grp_ptr[g + 1] - grp_ptr[g]);
                else if (use_tog) {
                    if (tog64) {
                        const int t = tog64[(size_t)g * 64 + lane];
                        const int r = t >= 0 ? (int)rank[t] : 0x7FFFFFFF;
                        if (r < nvalid) (void)__hip_atomic_fetch_xor(&bits[r >> 5], 1u << (r & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    } else flip_list(tog_idx + tog_ptr[g - 1], tog_ptr[g] - tog_ptr[g - 1]);
                } else {
                    flip_list(grp_idx + grp_ptr[g - 1], grp_ptr[g] - grp_ptr[g - 1]);
                    flip_list(grp_idx + grp_ptr[g], grp_ptr[g + 1] - grp_ptr[g]);
                }
                lds_wave_fence();
                // per-lane member counts, wave prefix scan, per-word prefix table
                uint32_t wv[WPL];
                int cl = 0;
#pragma unroll
                for (int j = 0; j < WPL; j++) { wv[j] = bits[lane * WPL + j]; cl += __popc(wv[j]); }
                const int incl = wave_incl_scan(cl);
                const int n = __builtin_amdgcn_readlane(incl, 63);
                if (n == 0) continue;  // no valid value in the ranking row: every member keeps "no rank"
                int run = incl - cl, myfirst = -1, mylast = -1;
#pragma unroll
                for (int j = 0; j < WPL; j++) {
                    pw[lane * WPL + j] = (uint16_t)run;
                    run += __popc(wv[j]);
                    if (wv[j]) {
                        if (myfirst < 0) myfirst = (lane * WPL + j) * 32 + __ffs((int)wv[j]) - 1;
                        mylast = (lane * WPL + j) * 32 + 31 - __clz((int)wv[j]);
                    }
                }
                lds_wave_fence();
                // members of the row with a global rank below r
                auto below = [&](int r) -> int {
                    if (r >= nvalid) return n;
                    const int w = r >> 5;
                    return (int)pw[w] + __popc(bits[w] & ((1u << (r & 31)) - 1u));
                };
                if (slice == 0) {  // extreme tie counts of the row (wave-uniform work)
                    const unsigned long long bm = __ballot(cl > 0);
                    const int p0 = __shfl(myfirst, __ffsll((long long)bm) - 1, 64);
                    const int p1 = __shfl(mylast, 63 - __clzll((long long)bm), 64);
                    const float v0 = val(p0), v1 = val(p1);
                    int e0 = p0 + 1, s1 = p1;
                    if (e0 < nvalid && val(e0) == v0) e0 = run_end(v0, e0 + 1, nvalid);
                    if (s1 > 0 && val(s1 - 1) == v1) s1 = run_begin(v1, 0, s1 - 1);
                    const int nmin = below(e0), nmax = n - below(s1);
                    if (lane == 0) put_stats(g, n, nmin, nmax);
                }
                // average-tie ranks of the elements this row ranks (the exact group)
                const int *ml = mem_idx + mem_ptr[g];
                const int mlen = mem_ptr[g + 1] - mem_ptr[g];
                for (int i0 = slice * 64; i0 < mlen; i0 += nslice * 64) {
                    const int i = i0 + lane;
                    const int t = i < mlen ? ml[i] : -1;
                    const int r = t >= 0 ? (int)rank[t] : 0x7FFFFFFF;
                    if (r < nvalid) {
                        const float v = from_key(keys[t]);
                        int rs = r, re = r + 1;
                        if (r > 0 && val(r - 1) == v) rs = run_begin(v, 0, r - 1);
                        if (re < nvalid && val(re) == v) re = run_end(v, re + 1, nvalid);
                        r2buf[t] = (uint16_t)(below(rs) + below(re) + 1);
                    }
                }
                lds_wave_fence();  // the next toggles rewrite bits / pw
            }
        }
        __syncthreads();

        // ---------------- D. percent ranks, interpolation on the quantile axis, apply (coalesced) ----------------
        bool flag_cell = false;
        const int tab_n = tmeta[0];
        const float tab_yf = __int_as_float(tmeta[1]), tab_yl = __int_as_float(tmeta[2]);
        const bool bad_nodes = af && G > 1 && tmeta[3];
        const float *afc = af ? af + (size_t)c * G * Q : nullptr;
        for (int t = tid; t < (int)T; t += NT) {
            const int g = gidx ? gidx[t] : 0;
            const int r2 = r2buf[t];
            const float v = from_key(keys[t]);
            double sq = __builtin_nan("");
            if (r2 && g >= 0) {
                const double N = gstat[3 * g], pmin = gstat[3 * g + 1], pmax = gstat[3 * g + 2];
                const double r = (double)r2 / 2.0;
                const double pr = r / N;
                sq = pmax * (pr - pmin) / (pmax - pmin);
            }
            if (sim_q) sim_q[(size_t)c * T + t] = sq;
            if (af) {
                float a;
                if (G == 1) {
                    int nd_unused;
                    a = interp1d_eval<float>(sq, true, qn, ty, tab_n, tab_yf, tab_yl, interp, extrap, nd_unused);
                } else if (g < 0 || sq != sq) a = __builtin_nanf("");
                else {
                    const float *row = afc + (size_t)g * Q;
                    if (sq < (double)qn[0]) { a = row[0]; flag_cell |= a != a; if (extrap != 0) a = __builtin_nanf(""); }
                    else if (sq > (double)qn[Q - 1]) { a = row[Q - 1]; flag_cell |= a != a; if (extrap != 0) a = __builtin_nanf(""); }
                    else {
                        int w;
                        (void)row_nearest<float, double>(sq, qn, Q, &w);
                        a = row[w];
                        flag_cell |= a != a;
                    }
                }
                scen[(size_t)c * T + t] = apply_corr<float>(v, a, kind);
            }
        }
        // cells whose own-row shortcut does not hold are queued for the generic kernel
        const int any = __syncthreads_or((flag_cell || bad_nodes) ? 1 : 0);
        if (any && tid == 0) {
            const int slot = atomicAdd(flag_count, 1);
            flag_list[slot] = (int)c;
        }
    }
}

}  // namespace xs
