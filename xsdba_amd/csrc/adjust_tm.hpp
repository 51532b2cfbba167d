// adjust_tm.hpp — qm_adjust (_adjustment.py:594-676, grouped interp="nearest") on TIME-MAJOR series: sim and scen are
// (time, cell) arrays — what a (time, lat, lon) DataArray is (base.py:653-656) — read and written in place.
//
// k_qm_adjust wants a cell's series contiguous; on a time-major cube the library used to re-lay sim with a tiled transpose
// first and hand scen back cell-major.  Here the roles turn by 90 degrees: a LANE owns a cell, a wave a tile of 64 cells and a
// run of groups.  For group g the wave takes the group's member time steps (uniform across lanes: scalar loads of the plan's
// member list): sim[t][c0 .. c0 + 63] is one coalesced 256-byte row, and so is the store of scen.  A lane's table rows
// hist_q / af [c][g][0 .. Q) are 2 x 4 Q contiguous bytes (the rows of consecutive groups follow each other: the wave walks
// its groups in order, so every fetched line is used whole); they are staged into a lane-private strip of LDS — no lane ever
// reads another lane's strip, so there is no barrier anywhere — and the rows of the next group are requested into registers
// while the current one is mapped.  Per element: the fixed-step lower bound of interp2d_nearest_dense in the lane's own row;
// when that row holds a NaN, or its nearest node is further than one row step away (the exact 2-D nearest neighbour may then
// sit in another row), the element goes through interp2d_nearest_global, the same search on the cell's table in global
// memory (rare: the bench data never takes it).  Same values as k_qm_adjust, bit for bit.
#pragma once
#include "kernels.hpp"

namespace xs {

constexpr int TM_WAVES = 2;     // waves per block: 21.5 KB of strips, seven blocks per CU (measured, ms per 64 800 cells x 10 950 days: four waves
                                // per block and three waves per SIMD 3.02, eight 4.44 — one block per CU —, two waves and four per SIMD 2.67)
constexpr int TM_CHUNK = 16;    // member time steps in flight per lane

__host__ __device__ inline int tm_row_stride(int Q) { return Q | 1; }  // odd: the lanes' strips start on different banks
__host__ __device__ inline size_t tm_wave_floats(int Q) { return (size_t)64 * 2 * tm_row_stride(Q) + 64; }  // strips + a flag per row
__host__ __device__ inline size_t tm_lds_bytes(int Q) { return (size_t)TM_WAVES * tm_wave_floats(Q) * sizeof(float); }

// utils._interp_on_quantiles_2d (method="nearest") + nbutils._extrapolate_on_quantiles on a cell's [G][Q] tables in global
// memory, NaN pairs skipped on the fly: the semantics of interp2d_nearest (device_math.hpp) on the compacted rows tab_stage
// builds.  any_valid: -1 = not known yet (found here by one scan of the table, then kept by the caller), else 0 / 1.
__device__ __attribute__((noinline)) float interp2d_nearest_global(float x, int g0, const float *__restrict__ xq, const float *__restrict__ yq, int G, int Q,
                                                int extrap, int &any_valid)
{
    if (x != x) return nan_of<float>();
    if (any_valid < 0) {
        any_valid = 0;
        for (int i = 0; i < G * Q && !any_valid; i++) any_valid = (xq[i] == xq[i] && yq[i] == yq[i]) ? 1 : 0;
    }
    if (!any_valid) return nan_of<float>();
    // nearest valid pair of an ascending row, as row_nearest finds it on the compacted row: the lower bound (first pair whose
    // abscissa is >= x) and the pair in front of it — the LAST pair below x, which matters when abscissae repeat —, the lower
    // one at equal distances
    auto row_scan = [&](int g, double &best, int &w) {
        const float *xr = xq + (size_t)g * Q, *yr = yq + (size_t)g * Q;
        int below = -1, above = -1;
        for (int k = 0; k < Q; k++) {
            const float xv = xr[k], yv = yr[k];
            if (xv == xv && yv == yv) {
                if (xv < x) below = k;
                else if (above < 0) above = k;
            }
        }
        best = __builtin_inf(); w = -1;
        if (below >= 0) { const double d = (double)x - (double)xr[below]; best = d * d; w = below; }
        if (above >= 0) { const double d = (double)x - (double)xr[above]; if (d * d < best) { best = d * d; w = above; } }
    };
    {   // extrapolation bounds of the own row: first / last non-NaN abscissa and factor (nbutils.py:380-389)
        const float *xr = xq + (size_t)g0 * Q, *yr = yq + (size_t)g0 * Q;
        double xf = __builtin_nan(""), xl = __builtin_nan("");
        float yf = nan_of<float>(), yl = nan_of<float>();
        bool fx = false, fy = false;
        for (int k = 0; k < Q; k++) {
            const float xv = xr[k], yv = yr[k];
            if (xv == xv) { if (!fx) { xf = (double)xv; fx = true; } xl = (double)xv; }
            if (yv == yv) { if (!fy) { yf = yv; fy = true; } yl = yv; }
        }
        if ((double)x < xf) return extrap == 0 ? yf : nan_of<float>();
        if ((double)x > xl) return extrap == 0 ? yl : nan_of<float>();
    }
    double best;
    int w;
    row_scan(g0, best, w);
    float val = w >= 0 ? yq[(size_t)g0 * Q + w] : nan_of<float>();
    if (best <= 1.0) return val;
    const int r = g0 + 1;
    for (int dr = 1; dr <= G + 1; dr++) {
        const double d2r = (double)dr * (double)dr;
        if (d2r >= best) break;
        for (int s = 0; s < 2; s++) {
            const int rr = s ? r + dr : r - dr;
            if (rr < 0 || rr > G + 1) continue;
            const int g = ext_row(rr, G);
            double d2;
            int w2;
            row_scan(g, d2, w2);
            if (w2 < 0) continue;
            d2 += d2r;
            if (d2 < best) { best = d2; val = yq[(size_t)g * Q + w2]; }
        }
    }
    return val;
}

// grid: (runs of `gpw` groups) x (quadruples of neighbouring 64-cell tiles); wave w of a block takes tile 4 b + w
// NI: load instructions per table and group.  A lane's row is 4 Q contiguous bytes, the rows of neighbouring lanes 4 G Q bytes
// apart: fetched by its own lane in 16-byte pieces, every piece is a memory request of its own that uses a quarter of what it
// brings in (measured: 3.8 ms per 64 800 cells against 2.6 for the cell-major kernel).  Instead consecutive LANES take the
// consecutive pieces of a row — P = Q / 4 lanes per cell, 64 / P cells per instruction — and drop them into the owner's strip.
// four waves per SIMD (128 VGPRs: the rows requested ahead take 48 of them; a dozen spill around the out-of-line slow path)
#ifndef XS_TM_WPE
#define XS_TM_WPE 4
#endif
template <int NI>
__global__ __launch_bounds__(TM_WAVES * 64) __attribute__((amdgpu_waves_per_eu(XS_TM_WPE, XS_TM_WPE))) void k_qm_adjust_tm(const float *__restrict__ sim, int64_t T, int64_t C, const float *__restrict__ af,
                                                                const float *__restrict__ hist_q, int G, int Q,
                                                                const int *__restrict__ mem_ptr, const int *__restrict__ mem_idx, int gpw,
                                                                int n_runs, int extrap, int kind, float *__restrict__ scen)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int RS = tm_row_stride(Q);
    float *wbase = reinterpret_cast<float *>(smem) + (size_t)wv * tm_wave_floats(Q);  // this wave's strips, then its 64 row flags
    float *xrow = wbase + ((size_t)lane * 2) * RS, *yrow = xrow + RS;                  // this lane's strips
    volatile int *rowflag = reinterpret_cast<volatile int *>(wbase + (size_t)64 * 2 * RS);
    // Lanes hand data to OTHER lanes of their wave through the LDS here (a loader lane writes the strips and the flag of other
    // cells' rows; the lane owning a row reads them).  The hardware serves a wave's LDS accesses in program order, but the
    // compiler knows nothing of the other lanes: the flag is volatile, and a wave-level fence — every access issued, nothing
    // moved across — stands between the writes of one side and the reads of the other.
    auto wave_lds_fence = []() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // the waves of a block take NEIGHBOURING tiles and the same run of groups: they walk the same time steps at about the same
    // time, so the block's loads of a (time, cell) row are TM_WAVES * 256 contiguous bytes, and neighbouring blocks continue the
    // row (a wave's 30 member rows lie 365 rows apart: locality inside DRAM pages only comes from the neighbours)
    const int64_t ntile = (C + 63) / 64, nt4 = (ntile + TM_WAVES - 1) / TM_WAVES;
    const int run = (int)(blockIdx.x / nt4);
    const int64_t tile = (blockIdx.x - (int64_t)run * nt4) * TM_WAVES + wv;
    const int g_lo = run * gpw, g_hi = min(G, g_lo + gpw);
    if (g_lo >= g_hi || tile >= ntile) return;
    const int64_t c = tile * 64 + lane;
    const bool live = c < C;
    const int64_t cc = live ? c : C - 1;  // (lanes past the last cell load the last cell's data and store nothing)
    const float *xq = hist_q + (size_t)cc * G * Q, *yq = af + (size_t)cc * G * Q;
    const int cc32 = (int)(cc - tile * 64);  // the lane's offset in the tile: row addresses are a scalar base + this
    const float *xin = sim + tile * 64;
    float *yout = scen + tile * 64;
    int any_valid = -1;
    // 16-byte pieces when the rows are 16-byte aligned (Q a multiple of 4), else every lane copies its own row
    const bool vec = (Q & 3) == 0 && ((reinterpret_cast<uintptr_t>(hist_q) | reinterpret_cast<uintptr_t>(af)) & 15) == 0;
    const int P = max(Q >> 2, 1), cpi = 64 / P;       // pieces per row, cells per instruction
    const int slot0 = lane / P, piece = lane - slot0 * P;
    const bool loader = slot0 < cpi;
    uint4 nx[NI], ny[NI];
    auto request = [&](int g) {
        if (vec) {
#pragma unroll
            for (int i = 0; i < NI; i++) {
                const int slot = i * cpi + slot0;
                if (loader && slot < 64) {
                    const int64_t cs = min(tile * 64 + slot, C - 1);
                    const size_t off = ((size_t)cs * G + g) * Q + 4 * piece;
                    nx[i] = *reinterpret_cast<const uint4 *>(hist_q + off);
                    ny[i] = *reinterpret_cast<const uint4 *>(af + off);
                }
            }
        }
    };
    auto stage = [&](int g) -> bool {  // registers (or global memory) -> the strips; true when this lane's row holds no NaN
        bool has_nan = false;
        if (vec) {
            wave_lds_fence();  // (the previous group's reads of the strips are done before they are overwritten)
            rowflag[lane] = 0;
            wave_lds_fence();
#pragma unroll
            for (int i = 0; i < NI; i++) {
                const int slot = i * cpi + slot0;
                if (loader && slot < 64) {
                    float *xs_ = wbase + ((size_t)slot * 2) * RS + 4 * piece, *ys_ = xs_ + RS;
                    const float fx[4] = {__uint_as_float(nx[i].x), __uint_as_float(nx[i].y), __uint_as_float(nx[i].z), __uint_as_float(nx[i].w)};
                    const float fy[4] = {__uint_as_float(ny[i].x), __uint_as_float(ny[i].y), __uint_as_float(ny[i].z), __uint_as_float(ny[i].w)};
                    bool bad = false;
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        xs_[u] = fx[u]; ys_[u] = fy[u];
                        bad |= (fx[u] != fx[u]) || (fy[u] != fy[u]);
                    }
                    if (bad) rowflag[slot] = 1;
                }
            }
            wave_lds_fence();  // (strips and flags of every lane are written before any lane reads its row)
            has_nan = rowflag[lane] != 0;
        } else {
            for (int k = 0; k < Q; k++) {
                const float xv = xq[(size_t)g * Q + k], yv = yq[(size_t)g * Q + k];
                xrow[k] = xv; yrow[k] = yv;
                has_nan |= (xv != xv) || (yv != yv);
            }
        }
        return !has_nan;
    };
#ifndef XS_TM_PREFETCH
#define XS_TM_PREFETCH 1
#endif
    if (XS_TM_PREFETCH) request(g_lo);
    for (int g = g_lo; g < g_hi; g++) {
        if (!XS_TM_PREFETCH) request(g);
        const bool dense = stage(g);
        if (XS_TM_PREFETCH && g + 1 < g_hi) request(g + 1);  // under the mapping of this group
        const int p0 = mem_ptr[g], p1 = mem_ptr[g + 1];
        for (int m0 = p0; m0 < p1; m0 += TM_CHUNK) {
            float v[TM_CHUNK];
            int tt[TM_CHUNK];
#pragma unroll
            for (int u = 0; u < TM_CHUNK; u++) {
                tt[u] = mem_idx[min(m0 + u, p1 - 1)];  // (uniform: scalar loads)
                v[u] = (xin + (int64_t)tt[u] * C)[cc32];
            }
            uint32_t slowmask = 0u;  // elements that need the search beyond the lane's own row: done after the others, out of line
#pragma unroll
            for (int u = 0; u < TM_CHUNK; u++) {
                if (m0 + u < p1) {  // (uniform)
                    const float x = v[u];
                    float a = nan_of<float>();
                    bool slow = !dense;
                    if (!slow) {
                        // interp2d_nearest_dense on the lane's own row
                        if (x != x) a = nan_of<float>();
                        else if (x < xrow[0]) a = extrap == 0 ? yrow[0] : nan_of<float>();
                        else if (x > xrow[Q - 1]) a = extrap == 0 ? yrow[Q - 1] : nan_of<float>();
                        else {
                            int lo = 0;
                            for (int n = Q; n > 1;) {
                                const int half = n >> 1;
                                lo = xrow[lo + half - 1] < x ? lo + half : lo;
                                n -= half;
                            }
                            lo += xrow[lo] < x ? 1 : 0;
                            const double d1 = (double)xrow[lo] - (double)x;
                            int w = lo;
                            double d = d1;
                            if (lo > 0) {
                                const double d0 = (double)x - (double)xrow[lo - 1];
                                if (!(d1 < d0)) { w = lo - 1; d = d0; }
                            }
                            a = yrow[w];
                            slow = d * d > 1.0;  // another row may hold the nearest node
                        }
                    }
                    if (slow) slowmask |= 1u << u;
                    else if (live) (yout + (int64_t)tt[u] * C)[cc32] = kind ? apply_corr<float>(x, a, kind) : a;
                }
            }
            if (__any(slowmask != 0u)) {
                // (rare; the values are read again rather than kept alive across the call)
                for (int u = 0; u < TM_CHUNK && m0 + u < p1; u++) {
                    if ((slowmask >> u) & 1u) {
                        const int64_t t = mem_idx[m0 + u];
                        const float x = (xin + t * C)[cc32];
                        const float a = interp2d_nearest_global(x, g, xq, yq, G, Q, extrap, any_valid);
                        if (live) (yout + t * C)[cc32] = kind ? apply_corr<float>(x, a, kind) : a;
                    }
                }
            }
        }
    }
}

}  // namespace xs
