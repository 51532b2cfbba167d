// preproc.hpp — the pre-processing steps that sit right before the quantile kernels for precipitation-like variables
// (SURVEY §8 row f1): jitter (processing.py:178-257) and frequency adaptation (_processing.py:20-142).
//
// Both draw random numbers.  The reference uses numpy's unseeded global generator (processing.py:234,
// _processing.py:122), so its output is not reproducible; here every draw is a pure function of
// (seed, cell, time step, stream) — the same function in the CPU checker under oracle/ — so device results are reproducible,
// independent of the launch geometry and bit-identical to the oracle:
//     h(cell, t, stream) = mix64( mix64(seed ^ (cell + 1) * 0x9E3779B97F4A7C15) ^ (stream << 40) ^ t )
//     u53(h) = (h >> 11) * 2^-53  in [0, 1)
// streams: 1 jitter lower, 2 jitter upper, 3 adapt_freq tie-break, 4 adapt_freq fill.
#pragma once
#include "kernels.hpp"

namespace xs {

__device__ __forceinline__ uint64_t rng_hash(uint64_t seed, int64_t cell, int64_t t, int stream)
{
    return mix64(mix64(seed ^ ((uint64_t)(cell + 1) * 0x9E3779B97F4A7C15ull)) ^ ((uint64_t)stream << 40) ^ (uint64_t)t);
}
__device__ __forceinline__ double u53(uint64_t h) { return (double)(h >> 11) * (1.0 / 9007199254740992.0); }

// ---------------------------------------------------------------------------------------------
// jitter (processing.py:178-257): values < lower (and not NaN) -> U(nextafter(minimum), lower);
// values >= upper -> U(upper, maximum) (maximum stepped down once for dtypes narrower than float64).
// np.random.uniform(low, high) = low + (high - low) * random_sample() in float64, cast to the data dtype.
// lower / upper NaN = that end is off.  Cell-major rows.
// ---------------------------------------------------------------------------------------------
template <typename R>
__global__ void k_jitter(const R *__restrict__ x, int64_t T, int64_t sc, double lower, double jmin, double upper, double jmax,
                         uint64_t seed, int64_t cell0, R *__restrict__ out, int64_t osc)
{
    const int64_t c = blockIdx.y;
    const R *xr = x + c * sc;
    R *o = out + c * osc;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < (int)T; t += gridDim.x * blockDim.x) {
        const R v = xr[t];
        R r = v;
        if (v == v) {
            if (lower == lower && (double)v < lower) r = (R)(jmin + (lower - jmin) * u53(rng_hash(seed, cell0 + c, t, 1)));
            if (upper == upper && (double)v >= upper) r = (R)(upper + (jmax - upper) * u53(rng_hash(seed, cell0 + c, t, 2)));
        }
        o[t] = r;
    }
}

// In-LDS bitonic sort of P (power of two) (key, tag) pairs, lexicographic order, by the whole block.
template <typename U> __device__ __forceinline__ void block_bitonic_sort_pairs(U *keys, uint32_t *tags, int P)
{
    const int half = P >> 1;
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < half; i += blockDim.x) {
                int l = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                int r = l | j;
                U a = keys[l], b = keys[r];
                uint32_t ta = tags[l], tb = tags[r];
                bool gt = a > b || (a == b && ta > tb);
                bool up = (l & k) == 0;
                if (gt == up) { keys[l] = b; keys[r] = a; tags[l] = tb; tags[r] = ta; }
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------
// adapt_freq (_processing.py:75-142), one workgroup per (cell, group) row (the group's windowed block, dims
// [time, window] pooled; base.py:261-272):
//   P0_sim = #(sim <= thresh) / #valid(sim)  (utils.ecdf :87-105), P0_ref likewise (training) or given (adjusting),
//   dP0 = (P0_hist - P0_ref) / P0_hist (NaN when P0_hist == 0); nothing changes unless dP0 > 0;
//   pth = np.nanquantile(ref row in float64, P0_hist) cast to the data dtype (nbutils.vecquantiles :164-195);
//   rnk = percent rank of every value of the row with ties broken at random (utils.rank use_random_tiebreak :622-629:
//         a random permutation inside each tie group), (k-1)/(N-1);
//   values with  rnk in [P0_ref / P0_hist * P0_sim, P0_sim]  become  (pth - thresh) * U[0,1) + thresh  (in the data dtype);
//   the time series keeps the window-centre element of each block (base.py:438-439), i.e. time step t is corrected by the
//   block of ITS group.
// Ties are broken by the order of (key, tag) pairs, tag = 16 random bits (stream 3, mixed with the group) | slot in the row.
// sim_ad must hold a copy of sim on entry (time steps in no group and untouched values keep it).
//
// adapt_freq_block_sel (the default): by SELECTION, no row limit.  The replaced values are those whose rank k lies in a band
// [k_lo, k_hi] (the percent rank is monotone in k: the band's ends by bisection on the very expression), i.e. whose (key, tag)
// lies between the k_lo-th and the k_hi-th smallest of the sub-threshold values; pth interpolates two order statistics of the
// ref row.  The four order statistics are found TOGETHER by one radix select, a byte per pass, most significant first
// (sizeof(key) passes over the keys of both rows, then 4 — rows beyond 65 536 values: 6 — over the tags of the sim elements on
// a selected key; four 256-bin histograms in LDS, a wave per histogram finds the bin), then one pass compares.  The rows are
// read from global memory once and staged in LDS (values, time steps) when they fit (Lcap > 0), else gathered in every pass.
// Two barriers per pass: histograms and results are double-buffered (a pass clears the next pass's bins while it counts).
// dynamic LDS: [4 counters, 2 x 8 results: 32 ints][2 x 4 x 256 bins][Lcap sim values][Lcap ref values (training)][Lcap time steps]
//
// adapt_freq_block (XSDBA_HIP_AF_SELECT=0: A/B runs, tests; rows within the LDS, at most 65 535 values): bitonic sorts in LDS of
// the ref row and of the sub-threshold (key, tag) pairs; a block with more pairs than the LDS takes goes through
// adapt_freq_block_sel.
// dynamic LDS: [red: 2*(NT/64)+2 doubles][keys: max(P_max, Ps) * sizeof(U)] with [tags: Ps * 4] behind the first Ps keys,
// Ps = min(P_max, P_sub_cap) pairs for the sub-threshold values
// ---------------------------------------------------------------------------------------------
#ifndef XS_AF_CLK
#define XS_AF_CLK 0
#endif
#ifndef XS_AF_BATCH  // elements of a thread whose index loads, then value loads, are issued together when the rows are read (0: one by one)
#define XS_AF_BATCH 2
#endif
#if XS_AF_CLK  // shader-clock shares of adapt_freq_block_sel's phases, printed by the first work item of block 0 (tuning builds)
#define AF_CLK(i) do { const long long now_ = clock64(); clk_[i] += now_ - last_; last_ = now_; } while (0)
#else
#define AF_CLK(i) do { } while (0)
#endif
// inclusive wave64 prefix sum on the DPP path (train_cell.hpp's wave_incl_scan; this header does not see it)
__device__ __forceinline__ int af_wave_incl_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
}

template <typename R>
__device__ __forceinline__ void adapt_freq_block_sel(const R *__restrict__ sim, const R *__restrict__ ref, int64_t sc, int64_t rsc,
                             const int *__restrict__ row_ptr, const int *__restrict__ row_idx, const int *__restrict__ gidx, int G,
                             double thresh, uint64_t seed, int64_t cell0, int train, double *__restrict__ P0_ref,
                             double *__restrict__ P0_hist, R *__restrict__ pth, double *__restrict__ dP0_out,
                             R *__restrict__ sim_ad, int64_t osc, int Lcap, int64_t block_id)
{
    using U = typename KeyOf<R>::U;
    constexpr int KBY = (int)sizeof(U);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *cnt = reinterpret_cast<int *>(smem);  // [0..3] #valid / #(<= thresh) of sim, of ref
    int *res2 = cnt + 8;                       // [pass & 1][12]: [2 s], [2 s + 1]: the digit of target s, its rank among that bin's; [8 + s]: the bin's count
    int *hist2 = cnt + 32;                     // [pass & 1][4][256]
    R *sv = reinterpret_cast<R *>(hist2 + 2048);
    R *srv = sv + Lcap;
    int *stt = reinterpret_cast<int *>(sv + (train ? 2 : 1) * (size_t)Lcap);
    const bool stage = Lcap > 0;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    const int64_t c = block_id / G;
    const int g = (int)(block_id % G);
    const int p0 = row_ptr[g], L = row_ptr[g + 1] - p0;
    const R *xs = sim + c * sc;
    const R *xr = train ? ref + c * rsc : nullptr;
    const size_t og = (size_t)c * G + g;

#if XS_AF_CLK
    long long clk_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = clock64();
#endif
    // ---- the rows, once: counts (utils.ecdf :87-105), staged copies
    if (threadIdx.x < 4) cnt[threadIdx.x] = 0;
    int c_n = 0, c_0 = 0, r_n = 0, r_0 = 0;
#if XS_AF_BATCH
    for (int i0 = threadIdx.x; i0 < L; i0 += XS_AF_BATCH * blockDim.x) {
        int tt[XS_AF_BATCH];
        R v[XS_AF_BATCH], rv[XS_AF_BATCH];
#pragma unroll
        for (int u = 0; u < XS_AF_BATCH; u++) tt[u] = row_idx[p0 + min(i0 + u * (int)blockDim.x, L - 1)];
#pragma unroll
        for (int u = 0; u < XS_AF_BATCH; u++) { v[u] = xs[tt[u]]; rv[u] = train ? xr[tt[u]] : (R)0; }
#pragma unroll
        for (int u = 0; u < XS_AF_BATCH; u++) {
            const int i = i0 + u * (int)blockDim.x;
            if (i < L) {
                if (stage) { sv[i] = v[u]; stt[i] = tt[u]; }
                if (v[u] == v[u]) { c_n++; if ((double)v[u] <= thresh) c_0++; }
                if (train) {
                    if (stage) srv[i] = rv[u];
                    if (rv[u] == rv[u]) { r_n++; if ((double)rv[u] <= thresh) r_0++; }
                }
            }
        }
    }
#else
    for (int i = threadIdx.x; i < L; i += blockDim.x) {
        const int tt = row_idx[p0 + i];
        const R v = xs[tt];
        if (stage) { sv[i] = v; stt[i] = tt; }
        if (v == v) { c_n++; if ((double)v <= thresh) c_0++; }
        if (train) {
            const R rv = xr[tt];
            if (stage) srv[i] = rv;
            if (rv == rv) { r_n++; if ((double)rv <= thresh) r_0++; }
        }
    }
#endif
    // (two elements of a thread at a time: -DXS_AF_BATCH=0 / 2 / 4: 1.94 / 1.76 / 1.68 ms per 8 192 one-group cells, 46.7 / 46.6 /
    //  46.3 per 8 192 day-of-year cells; four spill under the 64-VGPR cap)
#if defined(XS_AF_STOP) && XS_AF_STOP == 1
    if (c_n == -12345) sim_ad[0] = (R)0;
    return;
#endif
    c_n = wave_sum(c_n); c_0 = wave_sum(c_0); r_n = wave_sum(r_n); r_0 = wave_sum(r_0);
    __syncthreads();
    if (lane == 0) { atomicAdd(&cnt[0], c_n); atomicAdd(&cnt[1], c_0); atomicAdd(&cnt[2], r_n); atomicAdd(&cnt[3], r_0); }
    __syncthreads();
    AF_CLK(0);
    const int N = cnt[0], n0 = cnt[1], nr = cnt[2];
    const double P0_sim = N ? (double)n0 / (double)N : __builtin_nan("");
    double p0r, p0h;
    if (train) { p0r = nr ? (double)cnt[3] / (double)nr : __builtin_nan(""); p0h = P0_sim; }
    else { p0r = P0_ref[og]; p0h = P0_hist[og]; }
    const double dP0 = p0h == 0.0 ? __builtin_nan("") : (p0h - p0r) / p0h;
    if (threadIdx.x == 0) {
        if (train) { P0_ref[og] = p0r; P0_hist[og] = p0h; }
        if (dP0_out) dP0_out[og] = dP0;
    }
    if (!(dP0 > 0.0)) {  // no adaptation needed (or undefined): sim_ad keeps sim
        if (train && threadIdx.x == 0) pth[og] = nan_of<R>();
        return;
    }

    // ---- the band of ranks that are replaced.  With n0 > 0 values <= thresh only those can be replaced: a value above it has
    // rnk >= n0 / (N - 1) > n0 / N = P0_sim ("too big"); they come first in the full order, hence their ranks are k = 1 .. n0.
    // rnk(k) — xarray's k/N rescaled as utils.rank does, mx * (rnk - mn) / (mx - mn) with mx = N/N, mn = 1/N — is
    // non-decreasing in k (every operation is monotone under rounding; 0 / 0 = NaN for N = 1 fails both tests: replaced), so
    // "rnk < lowb" holds on a prefix of k = 1 .. n0 and "rnk > P0_sim" on a suffix.
    const double lowb = (p0r / p0h) * P0_sim;
    const R th = (R)thresh;
    auto rnk_of = [&](int k) {
        const double rk = (double)k / (double)N, mn = 1.0 / (double)N, mx = (double)N / (double)N;
        return mx * (rk - mn) / (mx - mn);
    };
    // (Found by wave 0, thirty-two probes of each search per round: the expression holds four float64 divisions, a bisection by
    //  every thread of the block was ~20 evaluations per thread — a fifth of the kernel on day-of-year rows.)
    if (wid == 0 && n0 > 0) {
        const int half = lane >> 5, l5 = lane & 31;
        int lo = 1, hi = n0 + 1;  // the first k with "not (rnk < lowb)" (lanes 0-31) / with "rnk > P0_sim" (lanes 32-63); n0 + 1: none
        while (__any(hi > lo)) {
            const int step = (hi - lo + 31) >> 5;
            const int kq = lo + l5 * step;  // ascending probes; at or beyond hi: true
            const bool pr = kq >= hi ? true : (half == 0 ? !(rnk_of(kq) < lowb) : (rnk_of(kq) > P0_sim));
            const unsigned long long bal = __ballot(pr);
            const uint32_t m = half == 0 ? (uint32_t)bal : (uint32_t)(bal >> 32);
            const int ft = m ? __ffs((int)m) - 1 : 32;  // the first true probe: the answer lies behind the probe before it, up to it
            const int nhi = ft < 32 ? min(lo + ft * step, hi) : hi;
            const int nlo = ft > 0 ? lo + (ft - 1) * step + 1 : lo;
            lo = nlo; hi = nhi;
        }
        if (l5 == 0) cnt[4 + half] = lo;
    }
    __syncthreads();
    const int k_lo = n0 > 0 ? cnt[4] : 1, k_hi = n0 > 0 ? cnt[5] - 1 : 0;
    AF_CLK(1);
    const bool do_s = k_lo <= k_hi, do_r = train != 0;
    // the (at most) two order statistics of the ref row that vecquantile_sorted reads (dP0 > 0: nr >= 1, p0h is a number)
    int ia = 0, ib = 0;
    if (do_r && nr > 1) {
        const double pp = p0h * 100.0;
        if (pp == 100.0) ia = ib = nr - 1;
        else if (pp != 0.0) {
            const double rank = 1.0 + (double)(nr - 1) * (pp / 100.0);
            int lo = (int)floor(rank) - 1;
            lo = lo < 0 ? 0 : (lo > nr - 2 ? nr - 2 : lo);
            ia = lo; ib = lo + 1;
        }
    }
    const uint64_t gm = mix64((uint64_t)(g + 1) * 0xD1B54A32D192ED03ull);
    const int TB = L > 65536 ? 48 : 32;  // bits of a tag: 16 random ones above the slot (the same order either way)
    auto tag_of = [&](int tt, int slot) { return ((uint64_t)(mix64(rng_hash(seed, cell0 + c, tt, 3) ^ gm) >> 48) << (TB - 16)) | (uint64_t)(uint32_t)slot; };
    auto sim_at = [&](int i) { return stage ? sv[i] : xs[row_idx[p0 + i]]; };
    auto ref_at = [&](int i) { return stage ? srv[i] : xr[row_idx[p0 + i]]; };
    auto tt_at = [&](int i) { return stage ? stt[i] : row_idx[p0 + i]; };

    // ---- the radix select.  Targets 0, 1: the (k_lo - 1)-th and (k_hi - 1)-th smallest (key, tag) of the sub-threshold sim
    // values (-0.0 ranks as +0.0, a tie, like the float comparison of the reference); 2, 3: the ia-th and ib-th smallest valid
    // ref key.
    U keyA = 0, keyB = 0, keyC = 0, keyD = 0;
    uint64_t tagA = 0, tagB = 0;
    int kA = k_lo - 1, kB = k_hi - 1, kC = ia, kD = ib;
#if defined(XS_AF_STOP) && XS_AF_STOP == 2
    const int npass = KBY;
#else
    const int npass = do_s ? KBY + TB / 8 : (do_r ? KBY : 0);
#endif
    // (two targets of a row share a histogram as long as the digits chosen for them are the same: the exponent bytes of a row's
    //  keys nearly always, every byte when both statistics fall on one value — half the adds of those passes)
    // An end of the band whose bin holds ONE candidate is known without its remaining digits (nothing else shares its prefix:
    // every other element compares by the prefix alone) — they are filled with zeros (the lower end) / ones (the upper end) and
    // the target leaves the passes.  Distinct values settle after two or three key passes; a tie group (the zeros of a dry
    // season) after two tag passes as a rule (16 random bits among a few hundred candidates).
    bool sameAB = true, sameCD = true, doneA = false, doneB = false, act_s = do_s;
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) hist2[i] = 0;
    __syncthreads();
    for (int p = 0; p < npass; p++) {
        if (!act_s && !(do_r && p < KBY)) break;  // (both ends of the band known, the ref row's statistics complete)
        int *hist = hist2 + 1024 * (p & 1), *res = res2 + 12 * (p & 1);
        {   // the bins of the next pass (last read before the second barrier of the previous pass)
            int *hn = hist2 + 1024 * ((p + 1) & 1);
            for (int i = threadIdx.x; i < 1024; i += blockDim.x) hn[i] = 0;
        }
        const bool kp = p < KBY;
        const int sh = kp ? 8 * (KBY - 1 - p) : TB - 8 - 8 * (p - KBY);
        const bool first = p == 0 || p == KBY;  // (no digits of the key / of the tag chosen yet)
        for (int i0 = 0; i0 < L; i0 += blockDim.x) {
            const int i = i0 + (int)threadIdx.x;
            if (act_s) {
                bool onA = false, onB = false;
                int d = 0;
                if (i < L) {
                    const R v = sim_at(i);
                    if (v == v && (double)v <= thresh) {
                        const U k = to_key(v + (R)0);
                        if (kp) {
                            d = (int)((k >> sh) & 0xFF);
                            onA = !doneA && (first || ((k ^ keyA) >> (sh + 8)) == 0);
                            onB = !doneB && (first || ((k ^ keyB) >> (sh + 8)) == 0);
                        } else {
                            const bool mA = !doneA && k == keyA, mB = !doneB && k == keyB;
                            if (mA || mB) {
                                const uint64_t tg = tag_of(tt_at(i), i);
                                d = (int)((tg >> sh) & 0xFF);
                                onA = mA && (first || ((tg ^ tagA) >> (sh + 8)) == 0);
                                onB = mB && (first || ((tg ^ tagB) >> (sh + 8)) == 0);
                            }
                        }
                    }
                }
                if (onA) atomicAdd(&hist[d], 1);
                if (onB && !sameAB) atomicAdd(&hist[256 + d], 1);
            }
            if (do_r && kp) {
                bool onC = false, onD = false;
                int d = 0;
                if (i < L) {
                    const R rv = ref_at(i);
                    if (rv == rv) {
                        const U k = to_key(rv);
                        d = (int)((k >> sh) & 0xFF);
                        onC = first || ((k ^ keyC) >> (sh + 8)) == 0;
                        onD = first || ((k ^ keyD) >> (sh + 8)) == 0;
                    }
                }
                if (onC) atomicAdd(&hist[512 + d], 1);
                if (onD && !sameCD) atomicAdd(&hist[768 + d], 1);
            }
        }
        AF_CLK(2);
        __syncthreads();
        AF_CLK(3);
        // a wave per target: the bin that holds its k-th candidate, and k among that bin's
        for (int s_ = wid; s_ < 4; s_ += nw) {
            if (s_ < 2 ? (!act_s || (s_ == 0 ? doneA : doneB)) : !(do_r && kp)) continue;
            const int k = s_ == 0 ? kA : (s_ == 1 ? kB : (s_ == 2 ? kC : kD));
            const int *h = hist + 256 * ((s_ == 1 && sameAB) ? 0 : ((s_ == 3 && sameCD) ? 2 : s_)) + 4 * lane;
            const int c0 = h[0], c1 = h[1], c2 = h[2], c3 = h[3];
            const int tot = c0 + c1 + c2 + c3;
            const int incl = af_wave_incl_scan(tot);
            const int excl = incl - tot;
            if (k >= excl && k < incl) {
                int r = k - excl, j = 0;
                if (r >= c0) { r -= c0; j = 1; if (r >= c1) { r -= c1; j = 2; if (r >= c2) { r -= c2; j = 3; } } }
                res[2 * s_] = 4 * lane + j;
                res[2 * s_ + 1] = r;
                res[8 + s_] = j == 0 ? c0 : (j == 1 ? c1 : (j == 2 ? c2 : c3));
            }
        }
        AF_CLK(4);
        __syncthreads();
        AF_CLK(5);
        if (act_s) {
            const bool sab = sameAB && !doneA && !doneB && res[0] == res[2];
            if (!doneA) {
                if (kp) keyA |= (U)res[0] << sh; else tagA |= (uint64_t)res[0] << sh;
                kA = res[1];
                doneA = res[8] == 1;  // (the digits below stay zero)
            }
            if (!doneB) {
                if (kp) keyB |= (U)res[2] << sh; else tagB |= (uint64_t)res[2] << sh;
                kB = res[3];
                if (res[9] == 1) {
                    doneB = true;  // the digits below: ones
                    if (kp) { keyB |= ((U)1 << sh) - (U)1; tagB = ~0ull; }
                    else tagB |= (1ull << sh) - 1ull;
                }
            }
            sameAB = sab;
            act_s = !(doneA && doneB);
        }
        if (do_r && kp) { keyC |= (U)res[4] << sh; keyD |= (U)res[6] << sh; kC = res[5]; kD = res[7]; sameCD = sameCD && res[4] == res[6]; }
    }

    // ---- pth = np.nanquantile on the float64 copy of the ref row (the gufunc's float64 loop), rounded once
    R pthv;
    if (train) {
        auto load = [&](int i) { return (double)from_key(i == ia ? keyC : keyD); };
        pthv = (R)vecquantile_sorted<double>(load, nr, p0h);
        if (threadIdx.x == 0) pth[og] = pthv;
    } else pthv = pth[og];

    if (n0 == 0) {
        // nothing under the threshold although the training asks for adaptation: P0_sim = 0 and the smallest value has
        // rnk = 0 = P0_sim, inside the band, so the reference replaces it.  Lexicographic (key, tag) minimum of the row.
        U bk = KeyOf<R>::MAXK;
        uint64_t bt = ~0ull;
        for (int i = threadIdx.x; i < L; i += blockDim.x) {
            const R v = sim_at(i);
            if (v == v) {
                const U k = to_key(v + (R)0);
                if (k <= bk) {
                    const uint64_t tg = tag_of(tt_at(i), i);
                    if (k < bk || tg < bt) { bk = k; bt = tg; }
                }
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const U ok = __shfl_down(bk, o, 64);
            const uint64_t ot = __shfl_down(bt, o, 64);
            if (ok < bk || (ok == bk && ot < bt)) { bk = ok; bt = ot; }
        }
        __syncthreads();
        U *wk = reinterpret_cast<U *>(hist2);                // per wave: key, tag
        uint64_t *wt = reinterpret_cast<uint64_t *>(hist2 + 256);
        if (lane == 0) { wk[wid] = bk; wt[wid] = bt; }
        __syncthreads();
        if (threadIdx.x == 0 && N > 0) {
            for (int w = 1; w < nw; w++) if (wk[w] < bk || (wk[w] == bk && wt[w] < bt)) { bk = wk[w]; bt = wt[w]; }
            const int tt = tt_at((int)(bt & (TB == 48 ? 0xFFFFFFFFull : 0xFFFFull)));
            const double rnk = rnk_of(1);
            if (gidx[tt] == g && !((rnk < lowb) || (rnk > P0_sim))) {
                const R u = (R)u53(rng_hash(seed, cell0 + c, tt, 4));
                sim_ad[c * osc + tt] = (pthv - th) * u + th;
            }
        }
        return;
    }
    if (!do_s) return;
    AF_CLK(6);
    for (int i = threadIdx.x; i < L; i += blockDim.x) {
        const R v = sim_at(i);
        if (!(v == v && (double)v <= thresh)) continue;
        const U k = to_key(v + (R)0);
        if (k < keyA || k > keyB) continue;
        const int tt = tt_at(i);
        if (gidx[tt] != g) continue;  // (only the block's centre elements reach the time series)
        if (k == keyA || k == keyB) {
            const uint64_t tg = tag_of(tt, i);
            if ((k == keyA && tg < tagA) || (k == keyB && tg > tagB)) continue;
        }
        const R u = (R)u53(rng_hash(seed, cell0 + c, tt, 4));
        sim_ad[c * osc + tt] = (pthv - th) * u + th;
    }
    AF_CLK(7);
#if XS_AF_CLK
    if (blockIdx.x == 0 && threadIdx.x == 0)
        printf("adapt_freq_block_sel clocks: stage+count %lld  band %lld  count %lld  bar1 %lld  scan %lld  bar2 %lld  update+pth %lld  final %lld  (L %d n0 %d passes %d)\n",
               clk_[0], clk_[1], clk_[2], clk_[3], clk_[4], clk_[5], clk_[6], clk_[7], L, n0, npass);
#endif
}

template <typename R>
__device__ __forceinline__ void adapt_freq_block(const R *__restrict__ sim, const R *__restrict__ ref, int64_t sc, int64_t rsc,
                             const int *__restrict__ row_ptr, const int *__restrict__ row_idx, const int *__restrict__ gidx, int G,
                             double thresh, uint64_t seed, int64_t cell0, int train, double *__restrict__ P0_ref,
                             double *__restrict__ P0_hist, R *__restrict__ pth, double *__restrict__ dP0_out,
                             R *__restrict__ sim_ad, int64_t osc, int P_max, int P_sub_cap, int64_t block_id)
{
    using U = typename KeyOf<R>::U;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *red = reinterpret_cast<double *>(smem);
    const int nw = (blockDim.x + 63) >> 6;
    U *keys = reinterpret_cast<U *>(red + 2 * nw + 2);
    uint32_t *tags = reinterpret_cast<uint32_t *>(keys + (P_max < P_sub_cap ? P_max : P_sub_cap));
    const int64_t c = block_id / G;
    const int g = (int)(block_id % G);
    const int p0 = row_ptr[g], L = row_ptr[g + 1] - p0;
    int P = 2;
    while (P < L) P <<= 1;
    const R *xs = sim + c * sc;
    const size_t og = (size_t)c * G + g;

    // ---- counts of the sim row
    double cnt0 = 0;  // values <= thresh
    int nval = 0;
    for (int i = threadIdx.x; i < L; i += blockDim.x) {
        const R v = xs[row_idx[p0 + i]];
        if (v == v) { nval++; if ((double)v <= thresh) cnt0 += 1.0; }
    }
    block_sum_count(cnt0, nval, red);
    const int N = nval;
    const double P0_sim = N ? cnt0 / (double)N : __builtin_nan("");
    double p0r, p0h;
    R pthv;
    if (train) {
        const R *xr = ref + c * rsc;
        double c0r = 0;
        int nr = 0;
        for (int i = threadIdx.x; i < P; i += blockDim.x) {
            U k = KeyOf<R>::MAXK;
            if (i < L) {
                const R v = xr[row_idx[p0 + i]];
                k = to_key(v);
                if (v == v) { nr++; if ((double)v <= thresh) c0r += 1.0; }
            }
            keys[i] = k;
        }
        block_sum_count(c0r, nr, red);
        p0r = nr ? c0r / (double)nr : __builtin_nan("");
        p0h = P0_sim;
        const double d = p0h == 0.0 ? __builtin_nan("") : (p0h - p0r) / p0h;
        pthv = nan_of<R>();
        if (d > 0.0) {  // block-uniform
            block_bitonic_sort<U>(keys, P);
            auto load = [&](int i) { return (double)from_key(keys[i]); };
            // np.nanquantile on the float64 copy of the row (the gufunc's float64 loop), rounded once
            pthv = (R)vecquantile_sorted<double>(load, nr, p0h);
        }
        __syncthreads();
        if (threadIdx.x == 0) { P0_ref[og] = p0r; P0_hist[og] = p0h; pth[og] = pthv; if (dP0_out) dP0_out[og] = d; }
    } else {
        p0r = P0_ref[og]; p0h = P0_hist[og]; pthv = pth[og];
        if (threadIdx.x == 0 && dP0_out) dP0_out[og] = p0h == 0.0 ? __builtin_nan("") : (p0h - p0r) / p0h;
    }
    const double dP0 = p0h == 0.0 ? __builtin_nan("") : (p0h - p0r) / p0h;
    if (!(dP0 > 0.0)) return;  // no adaptation needed (or undefined): sim_ad keeps sim

    // ---- ranks with random tie-break.  With n0 > 0 values <= thresh only those can be replaced: a value above it has
    // rnk >= n0 / (N - 1) > n0 / N = P0_sim ("too big").  So only the n0 sub-threshold values are sorted, as (key, tag)
    // pairs; they come first in the full order, hence the same ranks k = 1 .. n0.
    const int n0 = (int)cnt0;
    const double lowb = (p0r / p0h) * P0_sim;
    const R th = (R)thresh;
    // percent rank of the k-th value (k = 1 .. N): xarray's k/N rescaled as utils.rank does, mx * (rnk - mn) / (mx - mn)
    // with mx = N/N, mn = 1/N; 0/0 = NaN for a single value, for which neither test holds -> replaced
    auto replace_if_in_band = [&](int k, int tt) {
        const double rk = (double)k / (double)N, mn = 1.0 / (double)N, mx = (double)N / (double)N;
        const double rnk = mx * (rk - mn) / (mx - mn);
        const bool keep = (rnk < lowb) || (rnk > P0_sim);
        if (!keep) {
            const R u = (R)u53(rng_hash(seed, cell0 + c, tt, 4));
            sim_ad[c * osc + tt] = (pthv - th) * u + th;
        }
    };
    const uint64_t gm = mix64((uint64_t)(g + 1) * 0xD1B54A32D192ED03ull);
    auto tag_of = [&](int tt, int slot) { return ((uint32_t)(mix64(rng_hash(seed, cell0 + c, tt, 3) ^ gm) >> 48) << 16) | (uint32_t)slot; };
    if (n0 == 0) {
        // nothing under the threshold although the training asks for adaptation: P0_sim = 0 and the smallest value has
        // rnk = 0 = P0_sim, inside the band, so the reference replaces it.  Lexicographic (key, tag) minimum of the row.
        U bk = KeyOf<R>::MAXK;
        uint32_t bt = 0xFFFFFFFFu;
        for (int i = threadIdx.x; i < L; i += blockDim.x) {
            const int tt = row_idx[p0 + i];
            const R v = xs[tt];
            if (v == v) {
                const U k = to_key(v + (R)0);
                const uint32_t tg = tag_of(tt, i);
                if (k < bk || (k == bk && tg < bt)) { bk = k; bt = tg; }
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const U ok = __shfl_down(bk, o, 64);
            const uint32_t ot = __shfl_down(bt, o, 64);
            if (ok < bk || (ok == bk && ot < bt)) { bk = ok; bt = ot; }
        }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) { keys[threadIdx.x >> 6] = bk; tags[threadIdx.x >> 6] = bt; }
        __syncthreads();
        if (threadIdx.x == 0 && N > 0) {
            for (int w = 1; w < nw; w++) if (keys[w] < bk || (keys[w] == bk && tags[w] < bt)) { bk = keys[w]; bt = tags[w]; }
            const int tt = row_idx[p0 + (int)(bt & 0xFFFFu)];
            if (gidx[tt] == g) replace_if_in_band(1, tt);
        }
        return;
    }
    int Ps = 2;
    while (Ps < n0) Ps <<= 1;
    if (Ps > P_sub_cap) {  // block-uniform: more pairs than the LDS sorts — the whole block again, by selection (rows in global memory)
        __syncthreads();
        adapt_freq_block_sel<R>(sim, ref, sc, rsc, row_ptr, row_idx, gidx, G, thresh, seed, cell0, train, P0_ref, P0_hist, pth, dP0_out,
                                sim_ad, osc, 0, block_id);
        return;
    }
    int *fill = reinterpret_cast<int *>(red);
    if (threadIdx.x == 0) *fill = 0;
    for (int i = threadIdx.x; i < Ps; i += blockDim.x) { keys[i] = KeyOf<R>::MAXK; tags[i] = 0xFFFFFFFFu; }
    __syncthreads();
    for (int i = threadIdx.x; i < L; i += blockDim.x) {
        const int tt = row_idx[p0 + i];
        const R v = xs[tt];
        if (v == v && (double)v <= thresh) {
            const int pos = atomicAdd(fill, 1);
            keys[pos] = to_key(v + (R)0);  // -0.0 ranks as +0.0 (a tie), like the float comparison of the reference
            tags[pos] = tag_of(tt, i);
        }
    }
    __syncthreads();
    block_bitonic_sort_pairs<U>(keys, tags, Ps);
    for (int p = threadIdx.x; p < n0; p += blockDim.x) {
        const int tt = row_idx[p0 + (int)(tags[p] & 0xFFFFu)];
        if (gidx[tt] == g) replace_if_in_band(p + 1, tt);  // only the block's centre elements reach the time series
    }
}


// k_af_screen: the cheap first half of adapt_freq for a whole cell at once.  #valid and #(<= thresh) of every windowed
// group block come from prefix sums over time (a block of group g is the union, with multiplicity, of the windows of g's
// time steps: two prefix lookups per time step), i.e. O(T) per series instead of O(T * window).  Writes P0_sim -> P0_hist,
// P0_ref, dP0 and pth = NaN (training), and queues the blocks with dP0 > 0 — the only ones k_adapt_freq has to sort.
// Typical model precipitation has too FEW dry days (the drizzle effect): dP0 <= 0 almost everywhere and the queue is short.
// dynamic LDS: [4 prefix arrays of (T + 1) u16: valid / below-threshold counts of sim and ref][NT + 1 ints]
template <typename R>
__global__ void k_af_screen(const R *__restrict__ sim, const R *__restrict__ ref, int64_t T, int64_t C, int64_t sc, int64_t rsc,
                            const int *__restrict__ mem_ptr, const int *__restrict__ mem_idx, int G, int W, double thresh, int train,
                            double *__restrict__ P0_ref, double *__restrict__ P0_hist, R *__restrict__ pth,
                            double *__restrict__ dP0_out, int *__restrict__ work_list, int *__restrict__ n_work)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int Ti = (int)T, NT = blockDim.x;
    uint16_t *pv[2], *pl[2];
    for (int a = 0; a < 2; a++) { pv[a] = reinterpret_cast<uint16_t *>(smem) + (size_t)(2 * a) * (Ti + 1); pl[a] = pv[a] + (Ti + 1); }
    int *tot = reinterpret_cast<int *>(smem + (((size_t)8 * (Ti + 1) + 15) & ~(size_t)15));  // [2][NT] chunk totals
    const int chunk = (Ti + NT - 1) / NT, t0 = min(Ti, (int)threadIdx.x * chunk), t1 = min(Ti, t0 + chunk);
    const int h = W / 2;
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        for (int a = 0; a < (train ? 2 : 1); a++) {
            const R *x = (a ? ref + c * rsc : sim + c * sc);
            int nv = 0, nl = 0;
            for (int t = t0; t < t1; t++) { const R v = x[t]; if (v == v) { nv++; if ((double)v <= thresh) nl++; } }
            tot[threadIdx.x] = nv; tot[NT + threadIdx.x] = nl;
            __syncthreads();
            if (threadIdx.x < 128) {  // exclusive scan of the chunk totals: wave 0 scans the valid counts, wave 1 the others
                int *tt = tot + (threadIdx.x >> 6) * NT;
                const int ln = threadIdx.x & 63, per = (NT + 63) / 64, i0 = ln * per, i1 = min(NT, i0 + per);
                int run = 0;
                for (int i = i0; i < i1; i++) run += tt[i];
                int incl = run;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(incl, o, 64); if (ln >= o) incl += up; }
                int pre = incl - run;
                for (int i = i0; i < i1; i++) { const int v = tt[i]; tt[i] = pre; pre += v; }
            }
            __syncthreads();
            nv = tot[threadIdx.x]; nl = tot[NT + threadIdx.x];
            for (int t = t0; t < t1; t++) {
                pv[a][t] = (uint16_t)nv; pl[a][t] = (uint16_t)nl;
                const R v = x[t];
                if (v == v) { nv++; if ((double)v <= thresh) nl++; }
            }
            if (t1 == Ti && t0 < Ti) { pv[a][Ti] = (uint16_t)nv; pl[a][Ti] = (uint16_t)nl; }
            if (Ti == 0 && threadIdx.x == 0) { pv[a][0] = 0; pl[a][0] = 0; }
            __syncthreads();
        }
        for (int g = threadIdx.x; g < G; g += NT) {
            int64_t cnt[4] = {0, 0, 0, 0};  // sim valid, sim below, ref valid, ref below
            for (int p = mem_ptr[g]; p < mem_ptr[g + 1]; p++) {
                const int t = mem_idx[p];
                const int lo = max(0, t - h), hi = min(Ti - 1, t - h + W - 1) + 1;
                cnt[0] += pv[0][hi] - pv[0][lo]; cnt[1] += pl[0][hi] - pl[0][lo];
                if (train) { cnt[2] += pv[1][hi] - pv[1][lo]; cnt[3] += pl[1][hi] - pl[1][lo]; }
            }
            const size_t og = (size_t)c * G + g;
            const double P0_sim = cnt[0] ? (double)cnt[1] / (double)cnt[0] : __builtin_nan("");
            double p0r, p0h;
            if (train) {
                p0r = cnt[2] ? (double)cnt[3] / (double)cnt[2] : __builtin_nan("");
                p0h = P0_sim;
                P0_ref[og] = p0r; P0_hist[og] = p0h; pth[og] = nan_of<R>();
            } else { p0r = P0_ref[og]; p0h = P0_hist[og]; }
            const double d = p0h == 0.0 ? __builtin_nan("") : (p0h - p0r) / p0h;
            if (dP0_out) dP0_out[og] = d;
            if (d > 0.0) work_list[atomicAdd(n_work, 1)] = (int)og;
        }
        __syncthreads();  // the prefix arrays are rebuilt for the next cell
    }
}

// SORTS: the LDS-sort block function (XSDBA_HIP_AF_SELECT=0) — its own instantiation, so that the selection keeps its registers
// (the two in one kernel: 88 VGPRs, five waves per SIMD)
template <typename R, bool SORTS = false>
__global__ __attribute__((amdgpu_waves_per_eu(SORTS ? 4 : 8))) void k_adapt_freq(const R *__restrict__ sim, const R *__restrict__ ref, int64_t sc, int64_t rsc,
                             const int *__restrict__ row_ptr, const int *__restrict__ row_idx, const int *__restrict__ gidx, int G,
                             double thresh, uint64_t seed, int64_t cell0, int train, double *__restrict__ P0_ref,
                             double *__restrict__ P0_hist, R *__restrict__ pth, double *__restrict__ dP0_out,
                             R *__restrict__ sim_ad, int64_t osc, int P_max, int P_sub_cap,
                             int Lcap /* >= 0: by selection, rows of up to Lcap values staged in LDS (0: none); -1: the LDS sorts */,
                             const int *__restrict__ work_list, const int *__restrict__ n_work, int64_t n_all)
{
    // work_list (from k_af_screen: the (cell, group) blocks with dP0 > 0, count read on the device) or every block
    const int64_t n_items = work_list ? (int64_t)*n_work : n_all;
    for (int64_t item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int64_t id = work_list ? (int64_t)work_list[item] : item;
        if constexpr (!SORTS)
            adapt_freq_block_sel<R>(sim, ref, sc, rsc, row_ptr, row_idx, gidx, G, thresh, seed, cell0, train, P0_ref, P0_hist, pth, dP0_out,
                                    sim_ad, osc, Lcap, id);
        else
            adapt_freq_block<R>(sim, ref, sc, rsc, row_ptr, row_idx, gidx, G, thresh, seed, cell0, train, P0_ref, P0_hist, pth, dP0_out,
                                sim_ad, osc, P_max, P_sub_cap, id);
        __syncthreads();  // the LDS is reused by the next block of work
    }
}

// ---------------------------------------------------------------------------------------------
// Scaling.  k_group_mean: NaN-skipping mean of one (cell, group) block (dims [time, window] pooled,
// _adjustment.py:953-956), float64 accumulation, one workgroup per block.  k_scale_groups: scen = sim (+|*) af[group of t].
// ---------------------------------------------------------------------------------------------
template <typename R>
__global__ void k_group_mean(const R *__restrict__ x, int64_t sc, const int *__restrict__ row_ptr, const int *__restrict__ row_idx,
                             int G, R *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *red = reinterpret_cast<double *>(smem);
    const int64_t c = blockIdx.x / G;
    const int g = blockIdx.x % G;
    const int p0 = row_ptr[g], L = row_ptr[g + 1] - p0;
    const R *xs = x + c * sc;
    double s = 0;
    int n = 0;
    for (int i = threadIdx.x; i < L; i += blockDim.x) {
        const R v = xs[row_idx[p0 + i]];
        if (v == v) { s += (double)v; n++; }
    }
    block_sum_count(s, n, red);
    if (threadIdx.x == 0) out[(size_t)c * G + g] = n ? (R)(s / (double)n) : nan_of<R>();
}

// Factor of time step t: the value of its group (nearest), or — g_hi / frac given — the linear interpolation between two
// neighbouring groups as xarray's DataArray.interp / scipy interp1d compute it on the cyclically extended table
// (utils.broadcast :222-246, add_cyclic_bounds :284-314): slope = (y_hi - y_lo) in the table dtype, then
// slope * frac + y_lo in float64 (the node spacing is 1).
template <typename R>
__device__ __forceinline__ double group_factor(const R *f, int g_lo, const int *g_hi, const double *frac, int t)
{
    if (!frac) return (double)f[g_lo];
    const R lo = f[g_lo], hi = f[g_hi[t]];
    return (double)(R)(hi - lo) * frac[t] + (double)lo;
}

template <typename R>
__global__ void k_scale_groups(const R *__restrict__ sim, int64_t T, int64_t sc, const R *__restrict__ af, int G,
                               const int *__restrict__ gidx, const int *__restrict__ g_hi, const double *__restrict__ frac,
                               int kind, R *__restrict__ scen, int64_t osc)
{
    const int64_t c = blockIdx.y;
    const R *x = sim + c * sc, *f = af + (size_t)c * G;
    R *o = scen + c * osc;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < (int)T; t += gridDim.x * blockDim.x) {
        const int g = G > 1 ? gidx[t] : 0;
        R y = nan_of<R>();
        if (g >= 0) {
            if (!frac) y = apply_corr<R>(x[t], f[g], kind);
            else { const double a = group_factor<R>(f, g, g_hi, frac, t); y = (R)(kind == '+' ? (double)x[t] + a : (double)x[t] * a); }
        }
        o[t] = y;
    }
}

// utils.apply_correction of one series by another (detrending.py:102-120: detrend = x (+|*) invert(trend), retrend =
// x (+|*) trend; mult_skip_zeros keeps x where the trend is 0).  Both cell-major [cell][time], f broadcast over the cells
// when f_stride_c == 0.
template <typename R>
__global__ void k_series_correction(const R *__restrict__ x, const R *__restrict__ f, int64_t T, int64_t sc, int64_t fsc, int kind,
                                    int inv, int skip0, R *__restrict__ out, int64_t osc)
{
    const int64_t c = blockIdx.y;
    const R *xr = x + c * sc, *fr = f + c * fsc;
    R *o = out + c * osc;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += (int64_t)gridDim.x * blockDim.x) {
        const R v = xr[t], w = fr[t];
        R y = apply_corr<R>(v, inv ? invert<R>(w, kind) : w, kind);
        if (skip0 && w == (R)0) y = v;
        o[t] = y;
    }
}

// ---------------------------------------------------------------------------------------------
// LOCI (_adjustment.py:889-939).  Two kernels per training, one workgroup per (cell, group) block:
//   k_loci_thresh: q = (#(ref <= thresh) + 1) / (#valid ref + 1)  (utils._ecdf_1d :35-37: the sorted sample is prefixed
//                  with -inf), hist_thresh = np.nanquantile(hist block, q) (linear method, float64, cast to the dtype);
//   k_loci_factor: ms = mean of the block's hist values >= hist_thresh OF THEIR OWN GROUP (the mask is built on the
//                  time series, :918-919, before the blocks are formed), mo = mean of ref values >= thresh,
//                  af = (mo - thresh) / (ms - hist_thresh)  (get_correction, multiplicative).
//   k_loci_adjust: scen = max(af * (sim - hist_thresh) + thresh, 0) with the group of each time step (:934-937).
// dynamic LDS of k_loci_thresh: [red][keys: P * sizeof(U)]
// ---------------------------------------------------------------------------------------------
template <typename R>
__global__ void k_loci_thresh(const R *__restrict__ ref, const R *__restrict__ hist, int64_t sc, const int *__restrict__ row_ptr,
                              const int *__restrict__ row_idx, int G, double thresh, R *__restrict__ hist_thresh)
{
    using U = typename KeyOf<R>::U;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *red = reinterpret_cast<double *>(smem);
    const int nw = (blockDim.x + 63) >> 6;
    U *keys = reinterpret_cast<U *>(red + 2 * nw + 2);
    const int64_t c = blockIdx.x / G;
    const int g = blockIdx.x % G;
    const int p0 = row_ptr[g], L = row_ptr[g + 1] - p0;
    int P = 2;
    while (P < L) P <<= 1;
    const R *xr = ref + c * sc, *xh = hist + c * sc;
    double nle = 0;
    int nr = 0;
    for (int i = threadIdx.x; i < L; i += blockDim.x) {
        const R v = xr[row_idx[p0 + i]];
        if (v == v) { nr++; if ((double)v <= thresh) nle += 1.0; }
    }
    block_sum_count(nle, nr, red);
    const double q = (nle + 1.0) / ((double)nr + 1.0);
    double dummy = 0;
    int nh = 0;
    for (int i = threadIdx.x; i < P; i += blockDim.x) {
        U k = KeyOf<R>::MAXK;
        if (i < L) { const R v = xh[row_idx[p0 + i]]; k = to_key(v); if (v == v) nh++; }
        keys[i] = k;
    }
    block_sum_count(dummy, nh, red);
    block_bitonic_sort<U>(keys, P);
    if (threadIdx.x == 0) {
        // np.nanquantile(x, q), method "linear": virtual index (n - 1) q, lerp between the neighbours
        R res = nan_of<R>();
        if (nh > 0) {
            const double vi = (double)(nh - 1) * q;
            int lo = (int)floor(vi);
            lo = lo < 0 ? 0 : (lo > nh - 1 ? nh - 1 : lo);
            const int hi = lo + 1 > nh - 1 ? nh - 1 : lo + 1;
            const double a = (double)from_key(keys[lo]), b = (double)from_key(keys[hi]), gam = vi - (double)lo;
            res = (R)(a + (b - a) * gam);
        }
        hist_thresh[(size_t)c * G + g] = res;
    }
}

// k_loci_thresh for a row that does not fit the LDS (round 5): the hist row has been sorted in global memory by the pipeline of
// train_bigrows.hpp (k_big_fill / k_big_local / k_big_step: `keys` [rows][P], `nvalid`); this kernel counts ref and reads the two
// order statistics.  One block per (cell, group) row of the batch that starts at row0.
template <typename R>
__global__ __launch_bounds__(256) void k_loci_thresh_big(const R *__restrict__ ref, int64_t sc, const int *__restrict__ row_ptr,
                                                         const int *__restrict__ row_idx, int G, int64_t row0, int P, double thresh,
                                                         const typename KeyOf<R>::U *__restrict__ keys, const int *__restrict__ nvalid,
                                                         R *__restrict__ hist_thresh)
{
    __shared__ double red[2 * 4 + 2];
    const int64_t b = row0 + blockIdx.x;
    const int64_t c = b / G;
    const int g = (int)(b - c * G);
    const int p0 = row_ptr[g], L = row_ptr[g + 1] - p0;
    const R *xr = ref + c * sc;
    double nle = 0;
    int nr = 0;
    for (int i = threadIdx.x; i < L; i += blockDim.x) {
        const R v = xr[row_idx[p0 + i]];
        if (v == v) { nr++; if ((double)v <= thresh) nle += 1.0; }
    }
    block_sum_count(nle, nr, red);
    if (threadIdx.x == 0) {
        const double q = (nle + 1.0) / ((double)nr + 1.0);
        const typename KeyOf<R>::U *kr = keys + (size_t)blockIdx.x * (size_t)P;
        const int nh = nvalid[blockIdx.x];
        R res = nan_of<R>();
        if (nh > 0) {  // (the expressions of k_loci_thresh)
            const double vi = (double)(nh - 1) * q;
            int lo = (int)floor(vi);
            lo = lo < 0 ? 0 : (lo > nh - 1 ? nh - 1 : lo);
            const int hi = lo + 1 > nh - 1 ? nh - 1 : lo + 1;
            const double a = (double)from_key(kr[lo]), bb = (double)from_key(kr[hi]), gam = vi - (double)lo;
            res = (R)(a + (bb - a) * gam);
        }
        hist_thresh[(size_t)c * G + g] = res;
    }
}

template <typename R>
__global__ void k_loci_factor(const R *__restrict__ ref, const R *__restrict__ hist, int64_t sc, const int *__restrict__ row_ptr,
                              const int *__restrict__ row_idx, const int *__restrict__ gidx, int G, double thresh,
                              const R *__restrict__ hist_thresh, R *__restrict__ af)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *red = reinterpret_cast<double *>(smem);
    const int64_t c = blockIdx.x / G;
    const int g = blockIdx.x % G;
    const int p0 = row_ptr[g], L = row_ptr[g + 1] - p0;
    const R *xr = ref + c * sc, *xh = hist + c * sc, *sth = hist_thresh + (size_t)c * G;
    double so = 0, ss = 0;
    int no = 0, ns = 0;
    for (int i = threadIdx.x; i < L; i += blockDim.x) {
        const int tt = row_idx[p0 + i];
        const R vo = xr[tt], vs = xh[tt];
        if ((double)vo >= thresh) { so += (double)vo; no++; }
        const int gt = gidx[tt];
        if (gt >= 0 && vs >= sth[gt]) { ss += (double)vs; ns++; }
    }
    block_sum_count(so, no, red);
    block_sum_count(ss, ns, red);
    if (threadIdx.x == 0) {
        const R mo = no ? (R)(so / (double)no) : nan_of<R>(), ms = ns ? (R)(ss / (double)ns) : nan_of<R>();
        af[(size_t)c * G + g] = (mo - (R)thresh) / (ms - sth[g]);
    }
}

template <typename R>
__global__ void k_loci_adjust(const R *__restrict__ sim, int64_t T, int64_t sc, const R *__restrict__ af,
                              const R *__restrict__ hist_thresh, int G, const int *__restrict__ gidx,
                              const int *__restrict__ g_hi, const double *__restrict__ frac, double thresh,
                              R *__restrict__ scen, int64_t osc)
{
    const int64_t c = blockIdx.y;
    const R *x = sim + c * sc, *f = af + (size_t)c * G, *h = hist_thresh + (size_t)c * G;
    R *o = scen + c * osc;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < (int)T; t += gridDim.x * blockDim.x) {
        const int g = G > 1 ? gidx[t] : 0;
        R y = nan_of<R>();
        if (g >= 0) {
            if (!frac) { y = f[g] * (x[t] - h[g]) + (R)thresh; if (y < (R)0) y = (R)0; }  // clip(min=0) keeps NaN
            else {
                const double fa = group_factor<R>(f, g, g_hi, frac, t), th = group_factor<R>(h, g, g_hi, frac, t);
                double yd = fa * ((double)x[t] - th) + thresh;
                if (yd < 0.0) yd = 0.0;
                y = (R)yd;
            }
        }
        o[t] = y;
    }
}

}  // namespace xs
