// train_cell.hpp — fast training kernel: one 1024-thread workgroup per grid cell, float32.
//
// Windowed groups overlap (doy/31: 30 of 31 columns shared by neighbouring groups), so instead of sorting
// every (cell, group) row the kernel
//   A. loads the cell's whole series into LDS once (order-preserving uint keys, NaN -> MAXK),
//   B. sorts the T time indexes by key ONCE: LSD radix sort, 4-bit digits, keys reduced to their occupied
//      range (fewer passes), every thread owns EPT contiguous positions and counts its digits in a packed
//      nibble register (no atomics; stable by construction), a transposed scan gives each (digit, thread)
//      its base slot,
//   C. gives every wave a contiguous chunk of groups and a membership BIT-VECTOR IN RANK SPACE (T bits):
//      the first group of the chunk is built from its row list, each following group by XOR-ing the
//      host-built toggle list (symmetric difference of consecutive rows: 2 x n_years entries for doy windows);
//      the k-th order statistic of a group is select(k) on that bit-vector: per-lane popcounts, wave
//      prefix scan, owner-lane search, in-word select.  Exact, distribution-independent, ties included
//      (equal values have distinct ranks).  32 quantiles per round: lanes 0-31 select the lower neighbour,
//      lanes 32-63 the upper one; type-7 interpolation and the correction factor follow as in k_train_rows.
//
// Requirements (host checks, else k_train_rows): float32, time-contiguous series, T <= EPT*1024 with
// EPT <= 11, no (time, group) multiplicity > 1.  DQM: order statistics commute with the monotone group-mean
// normalisation, so they are selected on the raw values and normalised afterwards; cells where that
// argument fails (non-finite mean, non-positive multiplicative mean, NaN produced by the interpolation) are
// appended to `flag_list` and recomputed by k_train_rows.
#pragma once
#include "kernels.hpp"

namespace xs {

// Inclusive wave64 prefix sum on the DPP cross-lane path (no LDS traffic): Hillis-Steele inside each row of 16
// lanes (row_shr 1,2,4,8; lanes without a source read 0), then row_bcast:15 / row_bcast:31 carry the row totals.
__device__ __forceinline__ int wave_incl_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
}

__device__ __forceinline__ void lds_wave_fence() { __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <int EPT> struct CellCfg {
    static constexpr int NT = 1024;
    static constexpr int TPAD = EPT * NT;
    static constexpr int NWORDS = TPAD / 32;
    static constexpr int WPL = (NWORDS + 63) / 64;  // bit-vector words per lane
    static constexpr int BITW = WPL * 64;
    static constexpr int NWAVE = NT / 64;
    static constexpr int CNTN = NT + NT / 16;        // padded per-thread counter arrays (stride 17 dwords per 16)
    static constexpr int BS = NT + 2 * (NT / 16);    // padded base row (u16 units)
    static constexpr size_t off_keys = 0;
    static constexpr size_t off_idx0 = off_keys + (size_t)TPAD * 4;
    static constexpr size_t off_idx1 = off_idx0 + (size_t)TPAD * 2;
    static constexpr size_t off_union = off_idx1 + (size_t)TPAD * 2;
    // sort view of the union
    static constexpr size_t off_cntlo = off_union;
    static constexpr size_t off_cnthi = off_cntlo + (size_t)CNTN * 4;
    static constexpr size_t off_base = off_cnthi + (size_t)CNTN * 4;
    static constexpr size_t sort_end = off_base + (size_t)16 * BS * 2;
    // select view of the union
    static constexpr size_t off_rank = off_union;
    static constexpr size_t off_bits = off_rank + (size_t)TPAD * 2;
    static constexpr size_t sel_end = off_bits + (size_t)NWAVE * BITW * 4;
    static constexpr size_t off_misc = ((sort_end > sel_end ? sort_end : sel_end) + 15) & ~(size_t)15;
    static constexpr size_t lds_bytes = off_misc + 64 * 8;  // reduction scratch
};

template <int EPT>
__global__ __launch_bounds__(1024) void k_train_cell(
    const float *__restrict__ x0, const float *__restrict__ x1, int64_t T, int64_t sc, const int *__restrict__ row_ptr,
    const int *__restrict__ row_sorted, const int *__restrict__ tog_ptr, const int *__restrict__ tog_idx, int G,
    const float *__restrict__ q, int Q, int kind, int mode, float *af, float *__restrict__ out_hq, float *scaling,
    int *__restrict__ flag_count, int *__restrict__ flag_list, int debug_phase)
{
    using Cfg = CellCfg<EPT>;
    constexpr int NT = Cfg::NT, TPAD = Cfg::TPAD, WPL = Cfg::WPL, BITW = Cfg::BITW, BS = Cfg::BS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *keys = reinterpret_cast<uint32_t *>(smem + Cfg::off_keys);
    uint16_t *idx0 = reinterpret_cast<uint16_t *>(smem + Cfg::off_idx0);
    uint16_t *idx1 = reinterpret_cast<uint16_t *>(smem + Cfg::off_idx1);
    uint32_t *cntlo = reinterpret_cast<uint32_t *>(smem + Cfg::off_cntlo);
    uint32_t *cnthi = reinterpret_cast<uint32_t *>(smem + Cfg::off_cnthi);
    uint16_t *base = reinterpret_cast<uint16_t *>(smem + Cfg::off_base);
    uint16_t *rank = reinterpret_cast<uint16_t *>(smem + Cfg::off_rank);
    uint32_t *bits_all = reinterpret_cast<uint32_t *>(smem + Cfg::off_bits);
    uint32_t *red = reinterpret_cast<uint32_t *>(smem + Cfg::off_misc);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int64_t c = blockIdx.x;
    const int npass_arr = mode == 0 ? 1 : 2;
    bool flag_cell = false;

    for (int arr = 0; arr < npass_arr; arr++) {
        const float *x = (arr == 0 ? x0 : x1) + c * sc;
        // ---------------- A. load, keys, range ----------------
        int nv = 0;
        uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
        if ((reinterpret_cast<uintptr_t>(x) & 7) == 0) {  // 8-byte aligned rows: two elements per load
            for (int i = 2 * tid; i < TPAD; i += 2 * NT) {
                float2 v2 = make_float2(__builtin_nanf(""), __builtin_nanf(""));
                if (i + 1 < T) v2 = *reinterpret_cast<const float2 *>(x + i);
                else if (i < T) v2.x = x[i];
                uint32_t k0 = to_key(v2.x), k1 = to_key(v2.y);
                *reinterpret_cast<uint2 *>(keys + i) = make_uint2(k0, k1);
                if (k0 != 0xFFFFFFFFu) { nv++; kmin = min(kmin, k0); kmax = max(kmax, k0); }
                if (k1 != 0xFFFFFFFFu) { nv++; kmin = min(kmin, k1); kmax = max(kmax, k1); }
            }
        } else {
            for (int i = tid; i < TPAD; i += NT) {
                uint32_t k = i < T ? to_key(x[i]) : 0xFFFFFFFFu;
                keys[i] = k;
                if (k != 0xFFFFFFFFu) { nv++; kmin = min(kmin, k); kmax = max(kmax, k); }
            }
        }
#pragma unroll
        for (int e = 0; e < EPT; e++) idx0[tid * EPT + e] = (uint16_t)(tid * EPT + e);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            nv += __shfl_down(nv, o, 64);
            kmin = min(kmin, (uint32_t)__shfl_down((int)kmin, o, 64));
            kmax = max(kmax, (uint32_t)__shfl_down((int)kmax, o, 64));
        }
        if (lane == 0) { red[3 * wid] = (uint32_t)nv; red[3 * wid + 1] = kmin; red[3 * wid + 2] = kmax; }
        __syncthreads();
        nv = 0; kmin = 0xFFFFFFFFu; kmax = 0u;
        for (int w = 0; w < Cfg::NWAVE; w++) { nv += (int)red[3 * w]; kmin = min(kmin, red[3 * w + 1]); kmax = max(kmax, red[3 * w + 2]); }
        const int nvalid = nv;
        const uint32_t range = nvalid ? kmax - kmin : 0u;  // NaN / padding are remapped to range + 1
        const int nbits = 32 - __clz((int)(range + 1u) | 0) ;
        const int npass = max(1, ((range + 1u == 0u ? 32 : nbits) + 3) / 4);
        __syncthreads();

        if (debug_phase == 1) continue;  // timing aid (XSDBA_HIP_DEBUG_PHASE): stop after load/keys
        // ---------------- B. LSD radix sort of the time indexes ----------------
        int cur = 0;
        for (int p = 0; p < npass; p++) {
            const uint16_t *src = cur ? idx1 : idx0;
            uint16_t *dst = cur ? idx0 : idx1;
            const int shift = 4 * p;
            uint64_t cnt = 0;
            uint16_t ids[EPT];
            uint64_t dgs = 0;
#pragma unroll
            for (int e = 0; e < EPT; e++) {
                uint16_t id = src[tid * EPT + e];
                uint32_t k = keys[id];
                uint32_t r = (k == 0xFFFFFFFFu) ? range + 1u : k - kmin;
                uint32_t d = (r >> shift) & 15u;
                ids[e] = id;
                dgs |= (uint64_t)d << (4 * e);
                cnt += 1ull << (4 * d);
            }
            const int pi = tid + (tid >> 4);
            cntlo[pi] = (uint32_t)cnt;
            cnthi[pi] = (uint32_t)(cnt >> 32);
            __syncthreads();
            {   // transposed scan: thread j owns digit d = j>>6 of threads 16*chunk .. 16*chunk+15
                const int d = tid >> 6, chunk = tid & 63;
                const uint32_t *arrp = d < 8 ? cntlo : cnthi;
                const int sh = 4 * (d & 7);
                int run = 0;
                uint16_t pre[16];
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    int th = 16 * chunk + k;
                    int cdig = (int)((arrp[th + (th >> 4)] >> sh) & 15u);
                    pre[k] = (uint16_t)run;
                    run += cdig;
                }
                int incl = wave_incl_scan(run);
                if (lane == 63) red[wid] = (uint32_t)incl;
                __syncthreads();
                int woff = 0;
                for (int w = 0; w < wid; w++) woff += (int)red[w];
                const int off = woff + incl - run;
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    int th = 16 * chunk + k;
                    base[d * BS + th + 2 * (th >> 4)] = (uint16_t)(off + pre[k]);
                }
            }
            __syncthreads();
            {
                const int pb = tid + 2 * (tid >> 4);
#pragma unroll
                for (int e = 0; e < EPT; e++) {
                    int d = (int)((dgs >> (4 * e)) & 15u);
                    uint16_t pos = base[d * BS + pb];
                    base[d * BS + pb] = (uint16_t)(pos + 1);
                    dst[pos] = ids[e];
                }
            }
            __syncthreads();
            cur ^= 1;
        }
        const uint16_t *sorted = cur ? idx1 : idx0;
        if (debug_phase == 2) { if (tid == 0) out_hq[(size_t)c * G * Q] = (float)sorted[nvalid / 2]; __syncthreads(); continue; }
        for (int r = tid; r < TPAD; r += NT) rank[sorted[r]] = (uint16_t)r;
        __syncthreads();

        // ---------------- C. per-wave group chunks: membership bit-vector + select ----------------
        const int gpw = (G + Cfg::NWAVE - 1) / Cfg::NWAVE;
        const int g_begin = wid * gpw, g_end = min(G, g_begin + gpw);
        uint32_t *bits = bits_all + wid * BITW;
        for (int i = lane; i < BITW; i += 64) bits[i] = 0u;
        lds_wave_fence();
        double gsum = 0.0;
        // software prefetch of the next group's toggle list (first 64 entries): its L2 latency hides under the
        // current group's select
        int pre_t = -1, pre_len = 0, pre_base = 0;
        if (g_begin + 1 < g_end) {
            pre_base = tog_ptr[g_begin];
            pre_len = tog_ptr[g_begin + 1] - pre_base;
            pre_t = lane < pre_len ? tog_idx[pre_base + lane] : -1;
        }
        // quantile-index math is a function of (n, k) only: cached while n does not change between groups
        int n_prev = -1, c_case = 3, c_lo = 0;
        float c_gamma = 0.f;
        for (int g = g_begin; g < g_end; g++) {
            double dloc = 0.0;
            auto toggle = [&](int t) {
                int r = rank[t];
                if (r < nvalid) {
                    uint32_t m = 1u << (r & 31);
                    uint32_t old = atomicXor(&bits[r >> 5], m);
                    if (mode == 2) { float v = from_key(keys[t]); dloc += (old & m) ? -(double)v : (double)v; }
                }
            };
            if (g == g_begin) {
                const int *list = row_sorted + row_ptr[g];
                const int len = row_ptr[g + 1] - row_ptr[g];
                for (int i = lane; i < len; i += 64) toggle(list[i]);
            } else {
                const int t0 = pre_t, len = pre_len, lbase = pre_base;
                if (g + 1 < g_end) {
                    pre_base = tog_ptr[g];
                    pre_len = tog_ptr[g + 1] - pre_base;
                    pre_t = lane < pre_len ? tog_idx[pre_base + lane] : -1;
                }
                if (t0 >= 0) toggle(t0);
                for (int i = lane + 64; i < len; i += 64) toggle(tog_idx[lbase + i]);
            }
            lds_wave_fence();
            int cl = 0;
#pragma unroll
            for (int j = 0; j < WPL; j++) cl += __popc(bits[lane * WPL + j]);
            const int incl = wave_incl_scan(cl);
            const int excl = incl - cl;
            const int n = __builtin_amdgcn_readlane(incl, 63);
            float mu = 0.f, inv = 0.f;
            if (mode == 2) {
                gsum += wave_sum(dloc);
                gsum = __shfl(gsum, 0, 64);
                mu = n ? (float)(gsum / (double)n) : __builtin_nanf("");
                inv = invert<float>(mu, kind);
                if (n && (!isfinite(mu) || (kind == '*' && !(inv > 0.f && isfinite(inv))))) flag_cell = true;
            }
            const size_t obase = ((size_t)c * G + g) * (size_t)Q;
            for (int kb = 0; kb < Q; kb += 32) {
                const int k = kb + (lane & 31);
                const bool act = k < Q;
                const bool hi_lane = lane >= 32;
                int casek, lo;  // casek 0: general lerp, 1: clamp to a[n-1], 2: clamp to a[0], 3: empty
                float gamma;
                if (kb == 0 && n == n_prev) { casek = c_case; lo = c_lo; gamma = c_gamma; }
                else {
                    const double qk = (double)q[act ? k : 0];
                    const double vi = (double)n * qk + (1.0 + qk * (1.0 - 1.0 - 1.0)) - 1.0;
                    lo = 0; gamma = 0.f;
                    if (n == 0) casek = 3;
                    else if (vi != vi || vi >= (double)(n - 1)) { casek = 1; lo = n - 1; }
                    else if (vi < 0.0) { casek = 2; lo = 0; }
                    else { casek = 0; lo = (int)floor(vi); gamma = (float)(vi - (double)lo); }
                    if (kb == 0) { c_case = casek; c_lo = lo; c_gamma = gamma; }
                }
                const int j = (casek == 0 && hi_lane) ? lo + 1 : lo;
                // ---- select(j): owner lane (binary search over the lane prefix counts) ----
                int L = 0, eL = 0;
#pragma unroll
                for (int s = 32; s > 0; s >>= 1) {
                    int cand = L + s;
                    int e = __shfl(excl, cand & 63, 64);
                    bool ok = cand < 64 && e <= j;
                    L = ok ? cand : L;
                    eL = ok ? e : eL;
                }
                int rem = j - eL;
                // ---- word inside the owner's WPL words: all loads first, then a branch-free prefix compare ----
                const uint32_t *wp = bits + L * WPL;
                uint32_t wv[WPL];
#pragma unroll
                for (int i = 0; i < WPL; i++) wv[i] = wp[i];
                int wi = 0, sub = 0, run = 0;
#pragma unroll
                for (int i = 0; i < WPL - 1; i++) {
                    int ci = __popc(wv[i]);
                    run += ci;
                    bool ge = rem >= run;
                    wi += ge ? 1 : 0;
                    sub += ge ? ci : 0;
                }
                rem -= sub;
                uint32_t w = wv[0];
#pragma unroll
                for (int i = 1; i < WPL; i++) w = (wi == i) ? wv[i] : w;
                // ---- bit inside the word ----
                int pos = 0;
#pragma unroll
                for (int s = 16; s > 0; s >>= 1) {
                    int cw = __popc(w & ((1u << s) - 1u));
                    bool ge = rem >= cw;
                    rem -= ge ? cw : 0;
                    w = ge ? (w >> s) : w;
                    pos += ge ? s : 0;
                }
                int rsel = min((L * WPL + wi) * 32 + pos, TPAD - 1);
                int id = min((int)sorted[rsel], TPAD - 1);
                float v = from_key(keys[id]);
                if (mode == 2) v = apply_corr<float>(v, inv, kind);
                const float vhi = __shfl(v, (lane & 31) + 32, 64);
                float res;
                if (casek == 3) res = __builtin_nanf("");
                else if (casek != 0) res = v;
                else {
                    float left = v, right = vhi;
                    float diff = right - left;
                    if (gamma >= 0.5f) res = (float)((double)right - (double)diff * (1.0 - (double)gamma));
                    else res = left + diff * gamma;
                    if (res != res) flag_cell = true;  // inf - inf: needs a[n-1]; the generic kernel redoes the cell
                }
                if (act && !hi_lane) {
                    if (mode == 0) out_hq[obase + k] = res;
                    else if (arr == 0) af[obase + k] = res;  // ref_q parked in the af buffer until the hist pass
                    else {
                        out_hq[obase + k] = res;
                        af[obase + k] = get_corr<float>(res, af[obase + k], kind);
                    }
                }
            }
            n_prev = n;
            if (mode == 2 && lane == 0) {
                if (arr == 0) scaling[(size_t)c * G + g] = mu;
                else scaling[(size_t)c * G + g] = get_corr<float>(mu, scaling[(size_t)c * G + g], kind);
            }
        }
        __syncthreads();  // all waves are done with keys / sorted / rank before the next array reuses them
    }
    // cells whose shortcut assumptions failed are queued for the generic kernel
    int any = __syncthreads_or(flag_cell ? 1 : 0);
    if (any && tid == 0 && flag_list) {
        int slot = atomicAdd(flag_count, 1);
        flag_list[slot] = (int)c;
    }
}

}  // namespace xs
