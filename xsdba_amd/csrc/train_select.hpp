// train_select.hpp — training kernel for UN-WINDOWED groups (group="time", "time.month", "time.season", any grouping with
// window == 1 and a few groups): histogram multi-select instead of a sort.
//
// The reference sorts every group row to read 2*Q order statistics off it (nbutils.py:113-148: arr.sort(), then the
// neighbours of Q virtual indexes).  Only those 2*Q (+1: a[n-1] for the NaN rule) order statistics per group are needed,
// so the kernel never sorts:
//   A. the cell's series goes from HBM into REGISTERS (order-preserving keys, 2*EPT2 per thread), min / max / count;
//   B. one counting pass: hist[group][bin] (packed u16 LDS atomics), bin = (key - kmin) >> shift over the occupied key
//      range (equal-width bins in key space); the values equal to kmin or kmax are counted apart per group (zero-inflated
//      precipitation, capped humidity: their bins would be long runs of equal keys);
//   C. exclusive scan of every group's histogram (each thread 32 bins, block scan of the thread totals);
//   D. every target ordinal (group, j) finds its bin by bisection on the scanned histogram; shift == 0 bins are exact
//      values; otherwise the bin is marked and given a slice of a small candidate buffer;
//   E. second pass over the registers: the elements of marked bins are appended to their bin's slice;
//   F. every candidate ranks itself inside its slice (key, then slot: equal keys own distinct ranks), every target reads
//      the candidate holding its residual rank;
//   G. type-7 interpolation exactly as device_math.hpp::quantile_sorted, hist_q kept in LDS to form af in the ref pass
//      (every output byte is written once, nothing is read back).
// Exact and bit-identical to the sort-based kernels.  Bins that hold more than CAPBIN values are resolved only when all
// their keys are equal (min/max atomics); cells where that fails or whose candidates overflow the buffer are appended to
// flag_list and redone by the generic kernel.
// HBM traffic per cell = the series once + the tables once = the algorithmic bytes.
#pragma once
#include "train_cell.hpp"

namespace xs {

struct SelLayout {
    int NT, TP, G, Q, NB, log2nb, NTG, NTGT, NTGTP, cap;
    size_t off_hist, off_gc, off_pre, off_tkey, off_tbin, off_tr, off_tlead, off_tcnt, off_ltbase, off_ltcur, off_ltmin,
        off_ltmax, off_ckey, off_cinfo, off_hq, off_gmass, off_red, fixed_bytes, bytes;
};

constexpr int SEL_NT = 512, SEL_LOG2TOT = 14, SEL_CAPBIN = 64, SEL_MAXG = 32, SEL_MAXTGT = 2048;

// LDS layout, shared by the host (sizing) and the kernel.  cap = candidate slots (6 B each).
__host__ __device__ inline SelLayout sel_layout(int TP, int G, int Q, int cap)
{
    SelLayout L;
    L.NT = SEL_NT; L.TP = TP; L.G = G; L.Q = Q; L.cap = cap;
    int lg = 0;
    while ((1 << lg) < G) lg++;
    L.log2nb = SEL_LOG2TOT - lg;
    L.NB = 1 << L.log2nb;
    L.NTG = 2 * Q + 1;
    L.NTGT = G * L.NTG;
    L.NTGTP = ((L.NTGT + SEL_NT - 1) / SEL_NT) * SEL_NT;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = (o + bytes + 15) & ~(size_t)15; return r; };
    L.off_hist = take((size_t)G * L.NB * 2);
    L.off_gc = take((size_t)TP);
    L.off_pre = take((size_t)(SEL_NT + 1) * 4);
    L.off_tkey = take((size_t)L.NTGTP * 4);
    L.off_tbin = take((size_t)L.NTGTP * 2);
    L.off_tr = take((size_t)L.NTGTP * 2);
    L.off_tlead = take((size_t)L.NTGTP * 2);
    L.off_tcnt = take((size_t)L.NTGTP * 2);
    L.off_ltbase = take((size_t)L.NTGTP * 2);
    L.off_ltcur = take((size_t)L.NTGTP * 4);
    L.off_ltmin = take((size_t)L.NTGTP * 4);
    L.off_ltmax = take((size_t)L.NTGTP * 4);
    L.off_hq = take((size_t)G * Q * 4);
    L.off_gmass = take((size_t)2 * SEL_MAXG * 4);
    L.off_red = take(64 * 4);
    L.fixed_bytes = o;
    L.off_ckey = take((size_t)cap * 4);
    L.off_cinfo = take((size_t)cap * 2);
    L.bytes = o;
    return L;
}

// ordinal(s) a quantile needs from a row of n valid values (nbutils.py:29-68, 113-148): o0 (and o0 + 1 when `two`,
// with the interpolation weight gamma of nbutils.py:133-141)
__device__ __forceinline__ void sel_ordinals(int n, float qk_f, int &o0, bool &two, float &gamma)
{
    const double qk = (double)qk_f;
    const double vi = (double)n * qk + (1.0 + qk * (1.0 - 1.0 - 1.0)) - 1.0;
    two = false;
    gamma = 0.f;
    if (vi != vi || vi >= (double)(n - 1)) o0 = n - 1;
    else if (vi < 0.0) o0 = 0;
    else { o0 = (int)floor(vi); two = true; gamma = (float)(vi - (double)o0); }
}

// exclusive block scan of one int per thread (SEL_NT threads); two barriers; `red` = 8 words of LDS scratch
__device__ __forceinline__ int sel_block_excl_scan(int v, uint32_t *red, int lane, int wid, int &total)
{
    const int incl = wave_incl_scan(v);
    if (lane == 63) red[wid] = (uint32_t)incl;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SEL_NT / 64; w++) { const int r = (int)red[w]; woff += w < wid ? r : 0; tot += r; }
    __syncthreads();
    total = tot;
    return woff + incl - v;
}

// mode 0: quantiles of x0 -> out_hq.  mode 1 (eqm_train): x1 = hist -> out_hq, x0 = ref -> af = corr(hist_q, ref_q).
template <int EPT2>
__global__ __launch_bounds__(SEL_NT, (EPT2 <= 11 ? 4 : 2)) void k_train_sel(const float *__restrict__ x0, const float *__restrict__ x1, int64_t T, int64_t sc,
                                                       const int *__restrict__ gidx, int G, const float *__restrict__ q, int Q, int kind,
                                                       int mode, float *__restrict__ af, float *__restrict__ out_hq,
                                                       int *__restrict__ flag_count, int *__restrict__ flag_list, int64_t C, int cap)
{
    constexpr int NT = SEL_NT, NE = 2 * EPT2, TP = NE * NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const SelLayout L = sel_layout(TP, G, Q, cap);
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem + L.off_hist);
    uint16_t *hist16 = reinterpret_cast<uint16_t *>(smem + L.off_hist);
    uint8_t *gc = smem + L.off_gc;
    uint32_t *pre = reinterpret_cast<uint32_t *>(smem + L.off_pre);
    uint32_t *tkey = reinterpret_cast<uint32_t *>(smem + L.off_tkey);
    uint16_t *tbin = reinterpret_cast<uint16_t *>(smem + L.off_tbin);
    uint16_t *tr = reinterpret_cast<uint16_t *>(smem + L.off_tr);
    uint16_t *tlead = reinterpret_cast<uint16_t *>(smem + L.off_tlead);
    uint16_t *tcnt = reinterpret_cast<uint16_t *>(smem + L.off_tcnt);
    uint16_t *ltbase = reinterpret_cast<uint16_t *>(smem + L.off_ltbase);
    uint32_t *ltcur = reinterpret_cast<uint32_t *>(smem + L.off_ltcur);
    uint32_t *ltmin = reinterpret_cast<uint32_t *>(smem + L.off_ltmin);
    uint32_t *ltmax = reinterpret_cast<uint32_t *>(smem + L.off_ltmax);
    uint32_t *ckey = reinterpret_cast<uint32_t *>(smem + L.off_ckey);
    uint16_t *cinfo = reinterpret_cast<uint16_t *>(smem + L.off_cinfo);
    float *hqk = reinterpret_cast<float *>(smem + L.off_hq);
    uint32_t *gmass = reinterpret_cast<uint32_t *>(smem + L.off_gmass);  // [g][0: == kmin, 1: == kmax]
    uint32_t *red = reinterpret_cast<uint32_t *>(smem + L.off_red);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int NB = L.NB, log2nb = L.log2nb, NTG = L.NTG, NTGT = L.NTGT, NTGTP = L.NTGTP;
    const int ND = G * NB / 2;      // histogram dwords in use
    const int tpg = NB >> 5;        // scan threads per group (32 bins each)
    const int Ti = (int)T;
    constexpr uint32_t MAXK = 0xFFFFFFFFu;

    // the group of every time step, once per block (255: none)
    for (int i = tid; i < TP; i += NT) {
        const int g = i < Ti ? gidx[i] : -1;
        gc[i] = (uint8_t)((g >= 0 && g < G) ? g : 255);
    }
    __syncthreads();

    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        bool flag_cell = false;
        const int npass_arr = mode == 0 ? 1 : 2;
        for (int arr = 0; arr < npass_arr; arr++) {
            const float *x = ((mode == 0 || arr == 1) ? x0 : x1) + c * sc;  // hist first, then ref
            // ---------------- A. series -> keys in registers; range ----------------
            // (the thread's first element index is made opaque per pass: addresses and bounds derived from it are recomputed
            //  here instead of being hoisted out of the cell loop and spilled)
            int t2 = 2 * tid;
            __asm__ volatile("" : "+v"(t2));
            uint32_t key[NE];
            if ((reinterpret_cast<uintptr_t>(x) & 7) == 0) {
                const int Te = Ti & ~1;
                const float xlast = x[Ti - 1];
                float2 v2[EPT2];
#pragma unroll
                for (int j = 0; j < EPT2; j++) {
                    const int i = t2 + j * 2 * NT;
                    v2[j] = *reinterpret_cast<const float2 *>(x + min(i, Te - 2));
                }
                // the LDS tables are cleared while the loads are in flight
                for (int i = tid; i < ND / 4; i += NT) reinterpret_cast<uint4 *>(hist)[i] = make_uint4(0u, 0u, 0u, 0u);
                if (tid < 2 * G) gmass[tid] = 0u;
#pragma unroll
                for (int j = 0; j < EPT2; j++) {
                    const int i = t2 + j * 2 * NT;
                    float a = v2[j].x, b = v2[j].y;
                    if (i >= Te) { a = (i == Ti - 1) ? xlast : __builtin_nanf(""); b = __builtin_nanf(""); }
                    key[2 * j] = to_key(a);
                    key[2 * j + 1] = to_key(b);
                }
            } else {
                float v[NE];
#pragma unroll
                for (int e = 0; e < NE; e++) {
                    const int i = t2 + (e >> 1) * 2 * NT + (e & 1);
                    v[e] = x[min(i, Ti - 1)];
                }
                for (int i = tid; i < ND / 4; i += NT) reinterpret_cast<uint4 *>(hist)[i] = make_uint4(0u, 0u, 0u, 0u);
                if (tid < 2 * G) gmass[tid] = 0u;
#pragma unroll
                for (int e = 0; e < NE; e++) {
                    const int i = t2 + (e >> 1) * 2 * NT + (e & 1);
                    key[e] = i < Ti ? to_key(v[e]) : MAXK;
                }
            }
            // the group of each element, 8 bits per element (time steps without a group count as missing)
            uint32_t gpk[(NE + 3) / 4];
#pragma unroll
            for (int j = 0; j < EPT2; j++) {
                const int i = t2 + j * 2 * NT;
                const uint32_t g2 = *reinterpret_cast<const uint16_t *>(gc + i);
                if (j & 1) gpk[j >> 1] |= g2 << 16; else gpk[j >> 1] = g2;
            }
            uint32_t kmin = MAXK, kmax = 0u;
#pragma unroll
            for (int e = 0; e < NE; e++) {
                const uint32_t g = (gpk[e >> 2] >> (8 * (e & 3))) & 255u;
                if (g == 255u) key[e] = MAXK;
                if (key[e] != MAXK) { kmin = min(kmin, key[e]); kmax = max(kmax, key[e]); }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, o, 64));
                kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, o, 64));
            }
            if (lane == 0) { red[16 + 2 * wid] = kmin; red[16 + 2 * wid + 1] = kmax; }
            __syncthreads();  // (also: tables cleared)
#pragma unroll
            for (int w = 0; w < NT / 64; w++) { kmin = min(kmin, red[16 + 2 * w]); kmax = max(kmax, red[16 + 2 * w + 1]); }
            const bool any_valid = kmin != MAXK;
            const uint32_t range = any_valid ? kmax - kmin : 0u;
            const int nbits = range ? 32 - __clz((int)range) : 0;
            const int shift = max(0, nbits - log2nb);

            // ---------------- B. counting pass ----------------
#pragma unroll
            for (int i = 0; i < (NE + 3) / 4; i++) __asm__ volatile("" : "+v"(gpk[i]));
#pragma unroll
            for (int e = 0; e < NE; e++) {
                uint32_t k = key[e];
                __asm__ volatile("" : "+v"(k));
                const int g = (int)((gpk[e >> 2] >> (8 * (e & 3))) & 255u);
                const uint32_t rel = k - kmin;
                const bool valid = k != MAXK;
                const bool mass = valid && (rel == 0u || rel == range);
                if (valid && !mass) {
                    const uint32_t b = (uint32_t)(g << log2nb) + (rel >> shift);
                    (void)__hip_atomic_fetch_add(&hist[b >> 1], 1u << (16u * (b & 1u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                // (plain atomics: long runs of the extreme value serialise on one LDS word — twice the time of a
                //  run-free cell on zero-inflated data, still several times faster than sorting)
                if (mass) (void)__hip_atomic_fetch_add(&gmass[2 * g + (rel != 0u ? 1 : 0)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            __syncthreads();

            // ---------------- C. exclusive scan of every group's histogram ----------------
            {
                uint32_t w[16];
                const bool own = 16 * tid < ND;
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const uint4 v = own ? reinterpret_cast<const uint4 *>(hist)[4 * tid + d] : make_uint4(0u, 0u, 0u, 0u);
                    w[4 * d] = v.x; w[4 * d + 1] = v.y; w[4 * d + 2] = v.z; w[4 * d + 3] = v.w;
                }
                uint32_t s2 = 0u;  // both halves summed at once (each half-sum <= T < 65536)
#pragma unroll
                for (int d = 0; d < 16; d++) s2 += w[d];
                const int total = (int)(s2 & 0xFFFFu) + (int)(s2 >> 16);
                int btot;
                const int excl = sel_block_excl_scan(total, red, lane, wid, btot);
                pre[tid] = (uint32_t)excl;
                if (tid == NT - 1) pre[NT] = (uint32_t)btot;
                __syncthreads();
                if (own) {
                    const int gth = tid / tpg;
                    int run = excl - (int)pre[gth * tpg];
#pragma unroll
                    for (int d = 0; d < 16; d++) {
                        const int c0 = (int)(w[d] & 0xFFFFu), c1 = (int)(w[d] >> 16);
                        w[d] = (uint32_t)run | ((uint32_t)(run + c0) << 16);
                        run += c0 + c1;
                    }
#pragma unroll
                    for (int d = 0; d < 4; d++)
                        reinterpret_cast<uint4 *>(hist)[4 * tid + d] = make_uint4(w[4 * d], w[4 * d + 1], w[4 * d + 2], w[4 * d + 3]);
                }
            }
            __syncthreads();

            // ---------------- D. targets: ordinal -> bin (bisection on the scanned histogram) ----------------
            for (int tt = tid; tt < NTGTP; tt += NT) {
                uint32_t kres = MAXK;   // resolved key (MAXK: none / pending)
                uint32_t bsel = 0xFFFFu;  // pending: the bin
                int rres = 0, cres = 0;
                if (tt < NTGT) {
                    const int g = tt / NTG, j = tt - g * NTG;
                    const int nh = (int)pre[(g + 1) * tpg] - (int)pre[g * tpg];
                    const int nlo = (int)gmass[2 * g], nhi = (int)gmass[2 * g + 1];
                    const int n = nlo + nh + nhi;
                    if (n > 0) {
                        int o;
                        if (j == 2 * Q) o = n - 1;
                        else {
                            bool two;
                            float gm;
                            sel_ordinals(n, q[j >> 1], o, two, gm);
                            if (two) o += j & 1;
                        }
                        if (o < nlo) kres = kmin;
                        else if (o >= n - nhi) kres = kmax;
                        else {
                            const int oh = o - nlo;
                            const uint16_t *ex = hist16 + ((size_t)g << log2nb);
                            int b = 0;
                            for (int s = NB >> 1; s > 0; s >>= 1) {
                                const int cand = b + s;
                                b = (int)ex[cand] <= oh ? cand : b;
                            }
                            const int eb = (int)ex[b];
                            const int en = b + 1 < NB ? (int)ex[b + 1] : nh;
                            if (shift == 0) kres = kmin + (uint32_t)b;  // one key per bin
                            else { bsel = (uint32_t)b; rres = oh - eb; cres = en - eb; }
                        }
                    }
                }
                tkey[tt] = kres;
                tbin[tt] = (uint16_t)bsel;
                tr[tt] = (uint16_t)rres;
                tcnt[tt] = (uint16_t)cres;
            }
            __syncthreads();
            // every pending target claims its bin; the claim that survives names the bin's leader
            for (int tt = tid; tt < NTGT; tt += NT) {
                const uint32_t b = tbin[tt];
                if (b != 0xFFFFu) hist16[((size_t)(tt / NTG) << log2nb) + b] = (uint16_t)(0x8000u | (uint32_t)tt);
            }
            __syncthreads();
            int ctot = 0;  // candidate slots handed out so far
            for (int t0 = 0; t0 < NTGTP; t0 += NT) {
                const int tt = t0 + tid;
                int need = 0;
                bool leader = false, fat = false;
                if (tt < NTGT) {
                    const uint32_t b = tbin[tt];
                    if (b != 0xFFFFu) {
                        const int ld = (int)(hist16[((size_t)(tt / NTG) << log2nb) + b] & 0x7FFFu);
                        tlead[tt] = (uint16_t)ld;
                        leader = ld == tt;
                        fat = (int)tcnt[tt] > SEL_CAPBIN;
                        need = (leader && !fat) ? (int)tcnt[tt] : 0;
                    }
                }
                int rtot;
                const int base = ctot + sel_block_excl_scan(need, red, lane, wid, rtot);
                ctot += rtot;
                if (leader) {
                    ltbase[tt] = fat ? (uint16_t)0xFFFFu : (uint16_t)min(base, 0xFFFE);
                    ltcur[tt] = 0u;
                    ltmin[tt] = MAXK;
                    ltmax[tt] = 0u;
                }
            }
            const bool overflow = ctot > cap || ctot > 0xFFF0;  // block-uniform
            if (overflow) flag_cell = true;
            __syncthreads();

            // ---------------- E. second pass: the elements of marked bins -> candidate slices ----------------
            if (shift != 0 && !overflow) {
                // (the registers are made opaque: everything derived from them is recomputed here instead of being kept live —
                //  spilled — from the counting pass on)
#pragma unroll
                for (int i = 0; i < (NE + 3) / 4; i++) __asm__ volatile("" : "+v"(gpk[i]));
#pragma unroll
                for (int e = 0; e < NE; e++) {
                    uint32_t k = key[e];
                    __asm__ volatile("" : "+v"(k));
                    const int g = (int)((gpk[e >> 2] >> (8 * (e & 3))) & 255u);
                    const uint32_t rel = k - kmin;
                    if (k != MAXK && rel != 0u && rel != range) {
                        const uint32_t e16 = hist16[((size_t)g << log2nb) + (rel >> shift)];
                        if (e16 & 0x8000u) {
                            const int ld = (int)(e16 & 0x7FFFu);
                            const uint32_t b0 = ltbase[ld];
                            if (b0 == 0xFFFFu) {
                                (void)__hip_atomic_fetch_min(&ltmin[ld], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                (void)__hip_atomic_fetch_max(&ltmax[ld], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            } else {
                                const uint32_t s = b0 + __hip_atomic_fetch_add(&ltcur[ld], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                ckey[s] = k;
                                cinfo[s] = (uint16_t)ld;
                            }
                        }
                    }
                }
            }
            __syncthreads();
            // ---------------- F. candidates rank themselves inside their slice; targets pick theirs ----------------
            if (shift != 0 && !overflow) {
                for (int s = tid; s < ctot; s += NT) {
                    const int ld = (int)cinfo[s];
                    const int b0 = (int)ltbase[ld], m = (int)ltcur[ld];
                    const uint32_t k = ckey[s];
                    int rk = 0;
                    for (int i = 0; i < m; i++) {
                        const uint32_t ki = ckey[b0 + i];
                        rk += (ki < k || (ki == k && b0 + i < s)) ? 1 : 0;
                    }
                    cinfo[s] = (uint16_t)rk;
                }
            }
            __syncthreads();
            if (shift != 0 && !overflow) {
                for (int tt = tid; tt < NTGT; tt += NT) {
                    if (tbin[tt] == 0xFFFFu) continue;
                    const int ld = (int)tlead[tt];
                    const int b0 = (int)ltbase[ld];
                    uint32_t k = MAXK;
                    if (b0 == 0xFFFF) {
                        if (ltmin[ld] == ltmax[ld]) k = ltmin[ld];  // a long run of one value
                        else flag_cell = true;
                    } else {
                        const int m = (int)ltcur[ld], r = (int)tr[tt];
                        for (int i = 0; i < m; i++) k = (int)cinfo[b0 + i] == r ? ckey[b0 + i] : k;
                        if (k == MAXK) flag_cell = true;  // (cannot happen: the slice holds the whole bin)
                    }
                    tkey[tt] = k;
                }
            }
            __syncthreads();

            // ---------------- G. interpolation, factors, stores ----------------
            for (int o = tid; o < G * Q; o += NT) {
                const int g = o / Q, k = o - g * Q;
                const int nh = (int)pre[(g + 1) * tpg] - (int)pre[g * tpg];
                const int n = nh + (int)gmass[2 * g] + (int)gmass[2 * g + 1];
                float res = __builtin_nanf("");
                if (n > 0) {
                    int o0;
                    bool two;
                    float gamma;
                    sel_ordinals(n, q[k], o0, two, gamma);
                    const float v0 = from_key(tkey[g * NTG + 2 * k]), v1 = from_key(tkey[g * NTG + 2 * k + 1]);
                    // device_math.hpp::quantile_sorted with a(o0) = v0, a(o0 + 1) = v1, a(n - 1) = the group's maximum
                    if (!two) res = v0;
                    else {
                        const float diff = v1 - v0;
                        if (gamma >= 0.5f) res = (float)((double)v1 - (double)diff * (1.0 - (double)gamma));
                        else res = v0 + diff * gamma;
                        if (res != res) res = from_key(tkey[g * NTG + 2 * Q]);
                    }
                }
                const size_t oo = (size_t)c * G * Q + o;
                if (mode == 0) out_hq[oo] = res;
                else if (arr == 0) { out_hq[oo] = res; hqk[o] = res; }
                else af[oo] = get_corr<float>(hqk[o], res, kind);
            }
            __syncthreads();
        }
        const int any = __syncthreads_or(flag_cell ? 1 : 0);
        if (any && tid == 0 && flag_list) {
            const int slot = atomicAdd(flag_count, 1);
            flag_list[slot] = (int)c;
        }
    }
}

}  // namespace xs
