// train_cell3.hpp — k_train_cell with THREE group streams per wave (float32, EQM / plain quantiles, Q <= 21, the padded
// [G][64] toggle table).
//
// Phase C of k_train_cell serves one quantile per lane and two streams per wave: with the usual 20 quantiles, 40 of 64
// lanes work in the select — the most expensive part of a group step — and its instruction count does not depend on how
// many lanes take part.  Here a wave walks three streams (21 lanes each, lane 63 idle): the wave's groups [r0, r1) are cut
// in three runs, B walks down from gb - 1, A up from gb, C up from gb + nA.  ONE vector is built (group gb) and copied
// twice; C first fast-forwards nA toggle lists without selecting.  A third fewer group steps per wave for the same select
// cost; the popcount scan runs twice (A|B packed, C alone) and the owner-lane bisection reads two registers per step.
// LDS: three vectors per wave only fit because rank[] moves into the u16 index buffer the sort leaves free and the
// vectors are packed at 354 words (the last word any lane of a T <= 11264 series owns) instead of 384.
#pragma once
#include "train_cell.hpp"

namespace xs {

template <int EPT> struct CellCfg3 {
    using Base = CellCfg<EPT>;
    static constexpr int NT = 1024, TPAD = Base::TPAD, WPL = Base::WPL, NWAVE = Base::NWAVE;
    static constexpr int NLANE = (TPAD / 32 + WPL - 1) / WPL;            // lanes that own words
    static constexpr int VSTRIDE = NLANE * WPL + 4;                      // words per vector: whole lanes + two zero words (the
                                                                         // select looks two words past the owner lane's own) +
                                                                         // a scratch word (+ one for an even count)
    static constexpr int NAN_RANK = (NLANE * WPL + 2) * 32;              // "rank" of NaN / padding steps: a bit of the scratch word
    static_assert(NAN_RANK + 31 < 65536, "ranks are 16-bit");
    static_assert(WPL % 2 == 0 && VSTRIDE % 2 == 0, "the vectors are read in 8-byte pieces");
    static constexpr size_t off_bits = Base::off_union;
    static constexpr size_t sel_end = off_bits + (size_t)3 * NWAVE * VSTRIDE * 4;
    static constexpr size_t off_misc = ((Base::sort_end > sel_end ? Base::sort_end : sel_end) + 15) & ~(size_t)15;
    static constexpr size_t off_lut = off_misc + 64 * 8;   // [256] select-in-a-byte table (byte_select_lut)
    static constexpr size_t lds_bytes = off_lut + 256 * 4;
};

// the same words of three vectors `stride` bytes apart (the recount of the three streams): 3 * NW / 2 reads, one wait
template <int NW, int STRIDE> __device__ __forceinline__ void lds_read_words64x3(const uint32_t *p, uint32_t (&wa)[NW], uint32_t (&wb)[NW], uint32_t (&wc)[NW])
{
    static_assert(NW == 2 || NW == 4 || NW == 6, "1 to 3 ds_read_b64 per vector");
    const uint32_t a = (uint32_t)(uintptr_t)p;
    uint64_t v[9];
    if constexpr (NW == 2)
        __asm__ volatile("ds_read_b64 %0, %3\n\tds_read_b64 %1, %3 offset:%c4\n\tds_read_b64 %2, %3 offset:%c5\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(v[0]), "=&v"(v[3]), "=&v"(v[6]) : "v"(a), "i"(STRIDE), "i"(2 * STRIDE) : "memory");
    if constexpr (NW == 4)
        __asm__ volatile("ds_read_b64 %0, %6\n\tds_read_b64 %1, %6 offset:8\n\tds_read_b64 %2, %6 offset:%c7\n\tds_read_b64 %3, %6 offset:%c8\n\t"
                         "ds_read_b64 %4, %6 offset:%c9\n\tds_read_b64 %5, %6 offset:%c10\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[6]), "=&v"(v[7])
                         : "v"(a), "i"(STRIDE), "i"(STRIDE + 8), "i"(2 * STRIDE), "i"(2 * STRIDE + 8) : "memory");
    if constexpr (NW == 6)
        __asm__ volatile("ds_read_b64 %0, %9\n\tds_read_b64 %1, %9 offset:8\n\tds_read_b64 %2, %9 offset:16\n\t"
                         "ds_read_b64 %3, %9 offset:%c10\n\tds_read_b64 %4, %9 offset:%c11\n\tds_read_b64 %5, %9 offset:%c12\n\t"
                         "ds_read_b64 %6, %9 offset:%c13\n\tds_read_b64 %7, %9 offset:%c14\n\tds_read_b64 %8, %9 offset:%c15\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8])
                         : "v"(a), "i"(STRIDE), "i"(STRIDE + 8), "i"(STRIDE + 16), "i"(2 * STRIDE), "i"(2 * STRIDE + 8), "i"(2 * STRIDE + 16) : "memory");
#pragma unroll
    for (int i = 0; i < NW / 2; i++) {
        wa[2 * i] = (uint32_t)v[i]; wa[2 * i + 1] = (uint32_t)(v[i] >> 32);
        wb[2 * i] = (uint32_t)v[3 + i]; wb[2 * i + 1] = (uint32_t)(v[3 + i] >> 32);
        wc[2 * i] = (uint32_t)v[6 + i]; wc[2 * i + 1] = (uint32_t)(v[6 + i] >> 32);
    }
}

// Select inside a byte: entry b holds, three bits each, the places of the set bits of b in ascending order (the place of its
// j-th set bit is (lut[b] >> 3 j) & 7).  Built once per block; replaces the last three levels of the in-word bisection (21
// dependent vector instructions) by one LDS read and four instructions.
__device__ __forceinline__ void byte_select_lut(uint32_t *lut, int tid)
{
    if (tid < 256) {
        uint32_t t = 0u;
        int j = 0;
#pragma unroll
        for (int b = 0; b < 8; b++)
            if ((tid >> b) & 1) { t |= (uint32_t)b << (3 * j); j++; }
        lut[tid] = t;
    }
}

// inclusive wave64 prefix sum of a double on the DPP cross-lane path (the steps of wave_incl_scan on both halves of the value)
__device__ __forceinline__ double wave_incl_scan_f64(double v)
{
    auto step = [&](auto ctrl, auto rmask) {
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), decltype(ctrl)::value, decltype(rmask)::value, 0xf, false);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), decltype(ctrl)::value, decltype(rmask)::value, 0xf, false);
        v += __hiloint2double(hi, lo);
    };
    using std::integral_constant;
    step(integral_constant<int, 0x111>{}, integral_constant<int, 0xf>{});
    step(integral_constant<int, 0x112>{}, integral_constant<int, 0xf>{});
    step(integral_constant<int, 0x114>{}, integral_constant<int, 0xf>{});
    step(integral_constant<int, 0x118>{}, integral_constant<int, 0xf>{});
    step(integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{});
    step(integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{});
    return v;
}

// dqm_train's group means (_adjustment.py:150-163: the mean over the windowed block of every group) without walking the rows:
// the row of group g is the union of the windows of its members (base.py:261-272), so
//     sum(row g) = sum over the members t of g of  S(t),   S(t) = sum of x over the window of t = P[hi(t)] - P[lo(t) - 1]
// with P the float64 running sum of the series over time (NaN counts as 0; window clipped at both ends exactly like the
// plan's rows, multiplicities included).  Runs as the load hook of cell_load_sort: P lives in the LDS the sort has not
// claimed yet (10 bytes per time step would not fit any later), the G sums go to `gsum` behind the kernel's other LDS.
// A lane per group adds its members in member order: a fixed summation order, unlike running sums updated by toggles.
// Also leaves max |x| over the finite values in guard[0] (the guard of the multiplicative kind, see cell_group_quantiles3).
template <int EPT> struct GroupSumHook {
    static constexpr bool uses_sort_scratch = true;  // P[] reaches into it: the sort clears its histogram itself
    unsigned char *smem;
    double *gsum;          // [G] (LDS)
    double *guard;         // [1] (LDS): max |x|
    const int *mem_ptr, *mem_idx, *mem_t16;  // mem_t16: the plan's packed member table (groups of <= 32 members) or null
    int G, W;
    __device__ __forceinline__ void operator()(const uint32_t *keys, int Ti, int tid) const
    {
        using Cfg = CellCfg<EPT>;
        constexpr int NT = Cfg::NT, NW = Cfg::NWAVE;
        double *P = reinterpret_cast<double *>(smem + Cfg::off_idx0);  // [T] inclusive running sums (8 T bytes of the 10 T free ones)
        double *part = reinterpret_cast<double *>(smem + CellCfg3<EPT>::off_misc);  // [NW] wave totals, [NW] wave maxima
        const int lane = tid & 63, wid = tid >> 6;
        // every thread owns EPT consecutive time steps: their sum first (nothing else is kept: the kernel has no registers to
        // spare here, and a spilled partial costs a memory round trip in a phase every wave waits on), ...
        auto value = [&](int j, bool &ok) {
            const int i = tid * EPT + j;
            const float v = from_key(keys[min(i, Cfg::TPAD - 1)]);
            ok = i < Ti && v == v;
            return v;
        };
        double run = 0.0;
        float amaxf = 0.0f;
#pragma unroll
        for (int j = 0; j < EPT; j++) {
            bool ok;
            const float v = value(j, ok);
            run += ok ? (double)v : 0.0;
            amaxf = fmaxf(amaxf, (ok && fabsf(v) <= 3.4028234663852886e38f) ? fabsf(v) : 0.0f);
        }
        const double incl = wave_incl_scan_f64(run);
        // (maxima on the DPP path like the sums: the steps of wave_incl_scan, lanes without a source see 0; lane 63 ends up with
        //  the wave's — no LDS crossbar shuffles in front of a barrier every wave waits at)
        auto dpp_max = [&](auto ctrl, auto rmask) {
            amaxf = fmaxf(amaxf, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(amaxf), decltype(ctrl)::value, decltype(rmask)::value, 0xf, false)));
        };
        using std::integral_constant;
        dpp_max(integral_constant<int, 0x111>{}, integral_constant<int, 0xf>{});
        dpp_max(integral_constant<int, 0x112>{}, integral_constant<int, 0xf>{});
        dpp_max(integral_constant<int, 0x114>{}, integral_constant<int, 0xf>{});
        dpp_max(integral_constant<int, 0x118>{}, integral_constant<int, 0xf>{});
        dpp_max(integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{});
        dpp_max(integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{});
        float *pmax = reinterpret_cast<float *>(part + NW);  // [NW] wave maxima
        if (lane == 63) { part[wid] = incl; pmax[wid] = amaxf; }
        __syncthreads();
        // ... the NW wave totals scanned by the first lanes of wave 0, ...
        if (wid == 0) {
            static_assert(NW == 16, "the wave maxima fill one DPP row");
            const double tot = lane < NW ? part[lane] : 0.0;
            amaxf = pmax[lane & (NW - 1)];
            const double ex = wave_incl_scan_f64(tot) - tot;
            dpp_max(integral_constant<int, 0x111>{}, integral_constant<int, 0xf>{});
            dpp_max(integral_constant<int, 0x112>{}, integral_constant<int, 0xf>{});
            dpp_max(integral_constant<int, 0x114>{}, integral_constant<int, 0xf>{});
            dpp_max(integral_constant<int, 0x118>{}, integral_constant<int, 0xf>{});
            if (lane < NW) part[lane] = ex;
            if (lane == NW - 1) guard[0] = (double)amaxf;  // (the row's last lane holds the maximum of all sixteen)
        }
        __syncthreads();
        // ... then the running sums themselves, from the thread's offset (the values are read again)
        {
            double acc = (incl - run) + part[wid];
#pragma unroll
            for (int j = 0; j < EPT; j++) {
                bool ok;
                const float v = value(j, ok);
                acc += ok ? (double)v : 0.0;
                if (tid * EPT + j < Ti) P[tid * EPT + j] = acc;
            }
        }
        __syncthreads();
        const int hw = W / 2;
        auto window_sum = [&](int t) {
            const int lo = max(0, t - hw), hi = min(Ti - 1, t - hw + W - 1);
            return P[hi] - (lo > 0 ? P[lo - 1] : 0.0);
        };
        if (mem_t16) {
            // groups of <= 32 members (one per year): the packed member table of k_qdm_lanes, [16][G] words of two 16-bit time
            // indexes — coalesced, all sixteen words of a lane in flight.  Waves 2j and 2j + 1 share the groups [64j, 64j + 64):
            // the even wave adds members 0..15, the odd one members 16..31, in member order; the halves meet in `half`.
            double *half = reinterpret_cast<double *>(smem + Cfg::off_idx0) + Cfg::TPAD;  // [2][G] behind P (2 T bytes are left: G <= T / 8)
            for (int gb = 0; gb < G; gb += (NW / 2) * 64) {
                const int g = gb + (wid >> 1) * 64 + lane, h = wid & 1;
                uint32_t w[8];
#pragma unroll
                for (int u = 0; u < 8; u++) w[u] = g < G ? (uint32_t)mem_t16[(size_t)(8 * h + u) * G + g] : 0xFFFFFFFFu;
                double acc = 0.0;
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const uint32_t t0 = w[u] & 0xFFFFu, t1 = w[u] >> 16;
                    if (t0 != 0xFFFFu) acc += window_sum((int)t0);
                    if (t1 != 0xFFFFu) acc += window_sum((int)t1);
                }
                if (g < G) half[(size_t)h * G + g] = acc;
            }
            __syncthreads();
            for (int g = tid; g < G; g += NT) gsum[g] = half[g] + half[(size_t)G + g];
        } else {
            for (int g = tid; g < G; g += NT) {
                const int p0 = mem_ptr[g], p1 = mem_ptr[g + 1];
                double acc = 0.0;
                for (int pb = p0; pb < p1; pb += 8) {  // eight member indexes in flight
                    int tt[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) tt[u] = mem_idx[min(pb + u, p1 - 1)];
#pragma unroll
                    for (int u = 0; u < 8; u++)
                        if (pb + u < p1) acc += window_sum(tt[u]);
                }
                gsum[g] = acc;
            }
        }
        __syncthreads();
    }
};

// P10: every group row holds < 1024 values (day-of-year windows over up to ~33 years): the three per-lane counts share ONE
// register (10-bit fields), one prefix scan and one bpermute per bisection step serve all three streams.
// DQM: group means from running float64 sums updated by the same toggles (one per stream), normalisation of the selected
// order statistics, scaling — with the guards of cell_group_quantiles (train_cell.hpp).
// Vals: where the values come from (LdsVals: float32 keys in LDS; GlobalF64Vals: the float64 series in global memory).
// PRESUM (DQM): the group sums were formed by GroupSumHook (`gsum` [G], `guard` in LDS) instead of running sums kept by the toggles.
// STOP (diagnostic instantiations of scripts/ubench/train3_bench.hip only; the library builds STOP = 0): leave after
// 3 = the build of the middle group + copies + fast-forward, 4 = run the loop with toggles / recount / scan but no select.
// PADT: `tog64` is the table padded with time step n_time instead of -1 (float32 kernel, n_time < TPAD) and rank[] holds NAN_RANK
// for NaN / padding steps.
// NR: rounds of the select per group step — a stream's 21 lanes serve quantiles k, k + 21, ... (Q <= 21 NR nodes); everything up
// to the packed counts is done once per step whatever NR.
// Cfg: the geometry of the vectors (TPAD, WPL, VSTRIDE, NLANE, NWAVE: CellCfg3<EPT>, or train_cell4.hpp's 8-wave blocks);
// RankT: `const uint16_t *` into the LDS (lists in time steps), or a view of a rank table in global memory indexed by POSITION
// (train_xlong.hpp: RankGlobal; the row and toggle lists are then in positions too).
template <typename Cfg, typename RankT, bool P10, bool DQM, typename Vals, bool PRESUM = false, int STOP = 0, bool PADT = false, int NR = 1>
__device__ __forceinline__ void cell_group_quantiles3g(const Vals &vals, const RankT rank, uint32_t *bits_all, int nvalid,
                                                      int lane, int wid, const int *__restrict__ row_ptr,
                                                      const int *__restrict__ row_sorted, const int *__restrict__ tog64, int G,
                                                      const typename Vals::value_type *__restrict__ q, int Q, int kind, int mode,
                                                      int arr, int64_t c, typename Vals::value_type *__restrict__ af,
                                                      typename Vals::value_type *out_hq, typename Vals::value_type *scaling,
                                                      bool &flag_cell, const double *gsum = nullptr, const double *guard = nullptr,
                                                      int n_time = 0, const uint32_t *lut = nullptr,
                                                      typename Vals::value_type *hqk = nullptr)
{
    using R = typename Vals::value_type;
    constexpr int TPAD = Cfg::TPAD, WPL = Cfg::WPL, VS = Cfg::VSTRIDE, NL = Cfg::NLANE, QL = 21;
    wid = __builtin_amdgcn_readfirstlane(wid);  // (wave-uniform, and known to be: the group ranges below live in scalar registers)
    const int gpw = (G + Cfg::NWAVE - 1) / Cfg::NWAVE;
    const int r0 = min(G, wid * gpw), r1 = min(G, r0 + gpw), n = r1 - r0;
    if (n <= 0) return;
    const int nA = (n + 2) / 3, nB = (n - nA + 1) / 2, nC = n - nA - nB;  // nA >= nB >= nC; nA iterations
    const int gb = r0 + nB, gc3 = gb + nA;
    uint32_t *bA = bits_all + (size_t)(3 * wid) * VS, *bB = bA + VS, *bC = bB + VS;
    for (int i = lane; i < VS; i += 64) bA[i] = 0u;
    lds_wave_fence();
    const int sid = lane / QL, k = lane - QL * sid;  // lane 63: sid 3, takes no part
    const uint32_t *bits_s = sid == 0 ? bA : (sid == 1 ? bB : bC);
    double amax = 0.0;  // DQM: largest magnitude this lane pushed through the running sums
    auto flip = [&](uint32_t *bits, int t, double &dloc) {
        if (t >= 0) {
            const int r = (int)rank[t];
            if (r < nvalid) {
                const uint32_t m = 1u << (r & 31);
                if (DQM && !PRESUM) {
                    const uint32_t old = atomicXor(&bits[r >> 5], m);
                    const R v = vals.at_time(t);
                    dloc += (old & m) ? -(double)v : (double)v;
                    amax = fmax(amax, fabs((double)v));
                } else {
                    (void)__hip_atomic_fetch_xor(&bits[r >> 5], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    };
    double dlocA = 0.0, dlocB = 0.0, dlocC = 0.0, gsumA = 0.0, gsumB = 0.0, gsumC = 0.0;
    if (DQM && kind == '+' && wid == 0 && lane == 0 && nvalid > 1) {  // additive kind: extremes against the IQR, once per array
        R lo, hi, q1, q3;
        vals.at_ranks(0, nvalid - 1, lo, hi);
        vals.at_ranks(nvalid / 4, (3 * nvalid) / 4, q1, q3);
        constexpr double LIM_ADD = sizeof(R) == 8 ? 1.0e4 : 1.0e8;
        if (!(fmax(fabs((double)lo), fabs((double)hi)) <= LIM_ADD * ((double)q3 - (double)q1))) flag_cell = true;
    }
    // ---- the vector of group gb from its row list, copied to B and C
    bool early = false;
    int tf[8] = {0, 0, 0, 0, 0, 0, 0, 0}, e0B = 0, e1A = 0, e1B = 0, e1C = 0;
    {
        const int *l = row_sorted + row_ptr[gb];
        const int len = row_ptr[gb + 1] - row_ptr[gb];
        // (wave-uniform; len > 0: the early lists are requested inside the row loop, which an EMPTY middle group — a series that
        //  covers part of the year only, gaps in the group index — never enters: the lists are then fetched where they are used)
        early = PADT && len > 0 && len <= 1024 && nA <= 8;
        // 1024 entries per round (all of a day-of-year window's row): their reads are in flight together — one L2 round trip
        // where every wave of the block waits at the same time
        for (int i0 = 0; i0 < len; i0 += 1024) {
            int ts[16];
            if constexpr (PADT) {
                // all the row entries, then all their ranks, then all the flips in flight; the lanes past the end of the row sit
                // out (masked off: as flips of the scratch word they would all hit one address and be serialised)
#pragma unroll
                for (int u = 0; u < 16; u++) { const int i = i0 + 64 * u + lane; ts[u] = l[min(i, len - 1)]; }
                if (early) {
                    // the toggle lists of the fast-forward and of the first two steps are requested behind the row: the three L2
                    // round trips of the start-up — row, fast-forward lists, first lists — become one (sixteen waves wait here at once)
#pragma unroll
                    for (int u = 0; u < 8; u++) tf[u] = tog64[(size_t)(1 + u <= nA ? gb + 1 + u : 0) * 64 + lane];
                    e0B = tog64[(size_t)(nB > 0 ? gb : 0) * 64 + lane];
                    e1A = tog64[(size_t)(1 < nA ? gb + 1 : 0) * 64 + lane];
                    e1B = tog64[(size_t)(1 < nB ? gb - 1 : 0) * 64 + lane];
                    e1C = tog64[(size_t)(1 < nC ? gc3 + 1 : 0) * 64 + lane];
                }
                int rs[16];
#pragma unroll
                for (int u = 0; u < 16; u++) rs[u] = (int)rank[ts[u]];
#pragma unroll
                for (int u = 0; u < 16; u++)
                    if (i0 + 64 * u + lane < len)
                        (void)__hip_atomic_fetch_xor(&bA[rs[u] >> 5], 1u << (rs[u] & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
#pragma unroll
                for (int u = 0; u < 16; u++) { const int i = i0 + 64 * u + lane; ts[u] = i < len ? l[i] : -1; }
#pragma unroll
                for (int u = 0; u < 16; u++) flip(bA, ts[u], dlocA);
            }
        }
        lds_wave_fence();
        for (int i = lane; i < VS; i += 64) { const uint32_t w = bA[i]; bB[i] = w; bC[i] = w; }
        lds_wave_fence();
    }
    // ---- C fast-forwards to group gc3 (lists gb+1 .. gc3)
    if (nC > 0) {
        if constexpr (PADT) {
            // eight lists in flight (XOR commutes); rows past the last list read row 0, the table's all-padding row
            for (int s0 = 1; s0 <= nA; s0 += 8) {
                int ts[8], rs[8];
                if (early) {
#pragma unroll
                    for (int u = 0; u < 8; u++) ts[u] = tf[u];
                } else {
#pragma unroll
                    for (int u = 0; u < 8; u++) ts[u] = tog64[(size_t)(s0 + u <= nA ? gb + s0 + u : 0) * 64 + lane];
                }
#pragma unroll
                for (int u = 0; u < 8; u++) rs[u] = (int)rank[ts[u]];
#pragma unroll
                for (int u = 0; u < 8; u++) (void)__hip_atomic_fetch_xor(&bC[rs[u] >> 5], 1u << (rs[u] & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            lds_wave_fence();
        } else {
            for (int s0 = 1; s0 <= nA; s0 += 4) {
                int ts[4];
#pragma unroll
                for (int u = 0; u < 4; u++) ts[u] = s0 + u <= nA ? tog64[(size_t)(gb + s0 + u) * 64 + lane] : -1;
#pragma unroll
                for (int u = 0; u < 4; u++) { flip(bC, ts[u], dlocC); if (DQM && !PRESUM) lds_wave_fence(); }  // (running sums read the old bit: one list at a time)
                lds_wave_fence();
            }
        }
    }
    if constexpr (STOP == 3) { if (bC[lane] == 0x12345u) out_hq[0] = (R)1; return; }
    // ---- lock-step walk; the next lists are prefetched under the select.  Row 0 of the table is all -1 (list r sits in row r,
    // there is no list 0): a stream with nothing to toggle reads that row, so that the loads need no branches
    const int nolist = PADT ? n_time : -1;  // an entry that is no toggle
    int preA = nolist, preB = nolist, preC = nolist;  // A and C stand on their first group: only B starts with a step (gb -> gb - 1)
    if (nB > 0) preB = early ? e0B : tog64[(size_t)gb * 64 + lane];
    if constexpr (PADT) {
        // step 0's flips (B only) now, then the lists of step 1
        if (nB > 0) { const int r = (int)rank[preB]; (void)__hip_atomic_fetch_xor(&bB[r >> 5], 1u << (r & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        if (early) { preA = e1A; preB = e1B; preC = e1C; }
        else {
            preA = tog64[(size_t)(1 < nA ? gb + 1 : 0) * 64 + lane];
            preB = tog64[(size_t)(1 < nB ? gb - 1 : 0) * 64 + lane];
            preC = tog64[(size_t)(1 < nC ? gc3 + 1 : 0) * 64 + lane];
        }
    }
    int nA_prev = -1, nB_prev = -1, nC_prev = -1, c_case[NR], c_lo[NR];
    R c_gamma[NR];
#pragma unroll
    for (int rd = 0; rd < NR; rd++) { c_case[rd] = 3; c_lo[rd] = 0; c_gamma[rd] = (R)0; }
    // this cell's tables (the offsets inside them fit 32 bits: G * Q entries)
    R *const hq_c = out_hq + (size_t)c * G * (size_t)Q, *const af_c = af + (mode == 0 ? (size_t)0 : (size_t)c * G * (size_t)Q);
    const bool ref_pass = mode != 0 && arr != 0;
    // hqk (eight registers of the caller, alive across both passes of a cell): the hist quantiles this lane forms in the first pass,
    // for the second pass to form af against — a lane meets the same (group, quantile) at the same step of both passes.  Without
    // them the second pass reads the whole hist_q table back (20 % on top of the kernel's algorithmic bytes).  A queue rather than
    // an indexed array (that would live in scratch): pushed at [0] in the first pass, moved up so that step 0 stands at [7],
    // popped from [7] in the second.  More than eight steps per wave (G > 384): the table is read back.
#ifndef XS_HQK_KEEP
#define XS_HQK_KEEP 1
#endif
    const bool keep = XS_HQK_KEEP && NR == 1 && hqk != nullptr && mode != 0 && nA <= 8;  // (wave-uniform)
    auto hqk_up = [&]() {
#pragma unroll
        for (int j = 7; j > 0; j--) hqk[j] = hqk[j - 1];
    };
    // per-lane walk state: the group of this lane's stream moves by gdir per step, its output slot by gdir * Q
    const int gdir = sid == 1 ? -1 : 1, odir = gdir * Q;
    const int nlim = sid == 0 ? nA : (sid == 1 ? nB : (sid == 2 ? nC : 0));  // steps in which this lane's stream has a group
    int g = sid == 0 ? gb : (sid == 1 ? gb - 1 : gc3);
    int o = g * Q + k;
    const int fsh = (P10 ? 10 : 16) * (sid < 3 ? sid : 0);  // where this stream's field sits in the packed counts
    int Lprev4[NR];  // (owner lane of the previous step) * 4, the bpermute address, per round
#pragma unroll
    for (int rd = 0; rd < NR; rd++) Lprev4[rd] = -1;
    for (int it = 0; it < nA; it++, g += gdir, o += odir) {
        const bool vB = it < nB, vC = it < nC;
        const int tA = preA, tB = preB, tC = preC;
        if constexpr (PADT) {
            // (the flips of this step were issued in the middle of the previous one — before the loop for step 0 — see below)
        } else if constexpr (!DQM || PRESUM) {
            // branch-free: the three rank reads are in flight together, an entry that is no toggle (padding of the table, a NaN
            // step) XORs a zero mask into a word of its vector
            const int rA = (int)rank[max(tA, 0)], rB = (int)rank[max(tB, 0)], rC = (int)rank[max(tC, 0)];
            const uint32_t mA = (tA >= 0 && rA < nvalid) ? 1u << (rA & 31) : 0u, mB = (tB >= 0 && rB < nvalid) ? 1u << (rB & 31) : 0u,
                           mC = (tC >= 0 && rC < nvalid) ? 1u << (rC & 31) : 0u;
            (void)__hip_atomic_fetch_xor(&bA[rA >> 5], mA, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            (void)__hip_atomic_fetch_xor(&bB[rB >> 5], mB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            (void)__hip_atomic_fetch_xor(&bC[rC >> 5], mC, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            flip(bA, tA, dlocA);
            flip(bB, tB, dlocB);
            flip(bC, tC, dlocC);
        }
        int rnA = 0, rnB = 0, rnC = 0;  // PADT: ranks of the next step's flips
        if constexpr (PADT) {
            // the lists of step it + 1 (requested a step ago) -> the ranks of its flips, which are issued further down, as soon
            // as this step has read the vectors for the last time: their LDS latency is then covered by the rest of the step
            // instead of standing in front of the recount.  The lists of step it + 2 are requested in their place.
            rnA = (int)rank[preA]; rnB = (int)rank[preB]; rnC = (int)rank[preC];
            const int rowA = it + 2 < nA ? gb + it + 2 : 0, rowB = it + 2 < nB ? gb - it - 2 : 0, rowC = it + 2 < nC ? gc3 + it + 2 : 0;
            preA = tog64[(size_t)rowA * 64 + lane];
            preB = tog64[(size_t)rowB * 64 + lane];
            preC = tog64[(size_t)rowC * 64 + lane];
        } else {   // the lists of the next step (wave-uniform rows: scalar address arithmetic, three plain loads)
            const int rowA = it + 1 < nA ? gb + it + 1 : 0, rowB = it + 1 < nB ? gb - it - 1 : 0, rowC = it + 1 < nC ? gc3 + it + 1 : 0;
            preA = tog64[(size_t)rowA * 64 + lane];
            preB = tog64[(size_t)rowB * 64 + lane];
            preC = tog64[(size_t)rowC * 64 + lane];
        }
        // PADT: the flips of the next step (a stream without a next group would flip the padding step's scratch word 64 times
        // over: skipped, wave-uniformly)
        auto next_flips = [&]() {
            if constexpr (PADT) {
                if (it + 1 < nA) (void)__hip_atomic_fetch_xor(&bA[rnA >> 5], 1u << (rnA & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (it + 1 < nB) (void)__hip_atomic_fetch_xor(&bB[rnB >> 5], 1u << (rnB & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (it + 1 < nC) (void)__hip_atomic_fetch_xor(&bC[rnC >> 5], 1u << (rnC & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        };
        lds_wave_fence();
        // per-lane member counts: A | B packed in one register, C in another; two prefix scans
        constexpr int FB = P10 ? 10 : 16;            // field width of A and B
        constexpr int FM = (1 << FB) - 1;
        int clAB = 0, clC = 0;
        if (lane < NL) {
            // the lane's own words of the three vectors: WPL / 2 8-byte reads each, free of bank conflicts at a lane stride of
            // WPL * 4 bytes (ds_read_b64 spreads over 64 banks)
            uint32_t wa[WPL], wb[WPL], wc[WPL];
            lds_read_words64x3<WPL, VS * 4>(bA + lane * WPL, wa, wb, wc);
            int ca = 0, cb = 0, cc = 0;
#pragma unroll
            for (int j = 0; j < WPL; j++) { ca += __popc(wa[j]); cb += __popc(wb[j]); cc += __popc(wc[j]); }
            clAB = ca | (cb << FB);
            if (P10) clAB |= cc << 20;
            else clC = cc;
        }
        const int inAB = wave_incl_scan(clAB), inC = P10 ? 0 : wave_incl_scan(clC);
        const int exAB = inAB - clAB, exC = inC - clC;
        const int totAB = __builtin_amdgcn_readlane(inAB, 63);
        const int cntA = totAB & FM, cntB = (totAB >> FB) & FM;
        const int cntC = P10 ? ((totAB >> 20) & FM) : __builtin_amdgcn_readlane(inC, 63);
        if constexpr (STOP == 4) { if (cntA + exAB + cntB + cntC == 0x7FFFFFF1) out_hq[0] = (R)1; next_flips(); continue; }
        const int ns = P10 ? (int)__builtin_amdgcn_ubfe((unsigned)totAB, (unsigned)fsh, 10u) : (sid == 0 ? cntA : (sid == 1 ? cntB : cntC));
        const bool vs = it < nlim;
        // the ref pass forms af against the hist quantile written by the first pass: requested here, consumed after the select
        R hq_prev[NR];
#pragma unroll
        for (int rd = 0; rd < NR; rd++) {
            hq_prev[rd] = (R)0;
            if (ref_pass) {
                if (keep) { hq_prev[rd] = hqk[7]; hqk_up(); }
                else if (vs && k + QL * rd < Q) hq_prev[rd] = hq_c[o + QL * rd];
            }
        }
        R mu = (R)0, inv = (R)0;
        if (DQM && PRESUM) {
            const double gs = gsum[vs && sid < 3 ? g : 0];
            // sum / n through a refined reciprocal — the IEEE float64 division is ~35 instructions in every lane of every group
            // step.  v_rcp_f64 is good to ~2^-23 only: two Newton steps (2^-46, then below the float64 ulp) bring the product
            // within an ulp of the quotient, which the cast to R absorbs (one step left 1 float32 ulp of difference to the
            // oracle's mean in ~2 of 10^7 groups, seen on the 259 200-cell check)
            const double dn = (double)ns, r0 = __builtin_amdgcn_rcp(dn);
            const double r1 = __builtin_fma(__builtin_fma(-dn, r0, 1.0), r0, r0);
            mu = ns ? (R)(gs * __builtin_fma(__builtin_fma(-dn, r1, 1.0), r1, r1)) : nan_of<R>();
            inv = invert<R>(mu, kind);
            if (vs && sid < 3 && ns && (!isfinite(mu) || (kind == '*' && !(inv > (R)0 && isfinite(inv))))) flag_cell = true;
            // The sums are differences of float64 running sums over time: absolute error ~ eps * n_time * max |x|.  The
            // multiplicative kind divides by the mean: keep it below ~3e-9 of the group sum, else the generic kernel redoes the
            // cell with fresh sums (a group mean next to zero beside large values).  The additive kind is guarded above (IQR).
            constexpr double LIM = 1.0e3;
            if (kind == '*' && vs && sid < 3 && ns && !(guard[0] <= LIM * fabs(gs))) flag_cell = true;
        } else if (DQM) {
            // it == 0: dlocA holds the sum of the built vector (B and C start from it), dlocC the fast-forward of C on top
            // (the totals sit in lane 0; read into scalar registers: the running sums are wave-uniform and stay out of the VGPRs)
            auto uni = [](double v) {
                const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
                return __hiloint2double(hi, lo);
            };
            const double sA = wave_total_dpp(dlocA), sB = wave_total_dpp(dlocB), sC = wave_total_dpp(dlocC);
            dlocA = dlocB = dlocC = 0.0;
            if (it == 0) { gsumB = sA; gsumC = sA; }
            gsumA = uni(gsumA + sA);
            gsumB = uni(gsumB + sB);
            gsumC = uni(gsumC + sC);
            const double gs = sid == 0 ? gsumA : (sid == 1 ? gsumB : gsumC);
            mu = ns ? (R)(gs / (double)ns) : nan_of<R>();
            inv = invert<R>(mu, kind);
            if (vs && sid < 3 && ns && (!isfinite(mu) || (kind == '*' && !(inv > (R)0 && isfinite(inv))))) flag_cell = true;
            const double inf = __builtin_inf();
            const double gmin = fmin(fmin(cntA ? fabs(gsumA) : inf, (vB && cntB) ? fabs(gsumB) : inf), (vC && cntC) ? fabs(gsumC) : inf);
            constexpr double LIM = sizeof(R) == 8 ? 200.0 : 1.0e6;
            if (kind == '*' && !(amax <= LIM * gmin)) flag_cell = true;  // see cell_group_quantiles: rounding of the running sums
        }
        const bool same_counts = cntA == nA_prev && cntB == nB_prev && cntC == nC_prev;
        nA_prev = cntA; nB_prev = cntB; nC_prev = cntC;
#pragma unroll
        for (int rd = 0; rd < NR; rd++) {
        const int kr = k + QL * rd;  // this round's quantile of the lane
        const bool act = vs && kr < Q;
        // quantile-index math depends on (n, k) only: cached while the three streams keep their counts
        int casek, lo;  // 0: general lerp, 1: clamp to a[n-1], 2: clamp to a[0], 3: empty
        R gamma;
        if (same_counts) { casek = c_case[rd]; lo = c_lo[rd]; gamma = c_gamma[rd]; }
        else {
            const double qk = (double)q[kr < Q ? kr : 0];
            const double vi = (double)ns * qk + (1.0 + qk * (1.0 - 1.0 - 1.0)) - 1.0;
            lo = 0; gamma = (R)0;
            if (ns == 0) casek = 3;
            else if (vi != vi || vi >= (double)(ns - 1)) { casek = 1; lo = ns - 1; }
            else if (vi < 0.0) { casek = 2; lo = 0; }
            else { casek = 0; lo = (int)floor(vi); gamma = (R)(vi - (double)lo); }
            c_case[rd] = casek; c_lo[rd] = lo; c_gamma[rd] = gamma;
        }
        // owner lane of member `lo`: the last lane whose exclusive count is <= lo.  Consecutive groups share 900 of 930 members, so
        // the owner is nearly always the previous step's lane or a neighbour: four exclusive counts around it are fetched
        // together (ONE cross-lane latency instead of six dependent ones); a wave in which some lane's owner moved further
        // (or the first step) runs the bisection.  Lanes NL .. 63 own no words: their exclusive count is the stream's total,
        // above every `lo`, which ends both searches inside the vector.
        auto field = [&](int eab, int ec) {
            if (P10) return (int)__builtin_amdgcn_ubfe((unsigned)eab, (unsigned)fsh, 10u);
            return sid == 0 ? (eab & 0xFFFF) : (sid == 1 ? (int)((unsigned)eab >> 16) : ec);
        };
        int L4 = 0, eL = 0;
        bool miss = true;
        if (Lprev4[rd] >= 0) {  // (wave-uniform: -1 only in the first step)
            const int a0 = Lprev4[rd] - 4;  // lanes Lprev - 1 .. Lprev + 2 (the address wraps for Lprev == 0: that count is never used)
            const int e_m = field(__builtin_amdgcn_ds_bpermute(a0, exAB), P10 ? 0 : __builtin_amdgcn_ds_bpermute(a0, exC));
            const int e_0 = field(__builtin_amdgcn_ds_bpermute(a0 + 4, exAB), P10 ? 0 : __builtin_amdgcn_ds_bpermute(a0 + 4, exC));
            const int e_1 = field(__builtin_amdgcn_ds_bpermute(a0 + 8, exAB), P10 ? 0 : __builtin_amdgcn_ds_bpermute(a0 + 8, exC));
            const int e_2 = field(__builtin_amdgcn_ds_bpermute(a0 + 12, exAB), P10 ? 0 : __builtin_amdgcn_ds_bpermute(a0 + 12, exC));
            const bool dn = e_0 > lo, st = e_1 > lo;  // the owner is below Lprev / not above it
            L4 = dn ? a0 : (st ? Lprev4[rd] : Lprev4[rd] + 4);
            eL = dn ? e_m : (st ? e_0 : e_1);
            miss = dn ? (e_m > lo || Lprev4[rd] == 0) : (!st && e_2 <= lo);
        }
        if (__any(miss)) {
            L4 = 0; eL = 0;
#pragma unroll
            for (int s = 32; s > 0; s >>= 1) {
                const int cand4 = L4 + 4 * s;
                const int e = field(__builtin_amdgcn_ds_bpermute(cand4, exAB), P10 ? 0 : __builtin_amdgcn_ds_bpermute(cand4, exC));
                const bool ok = e <= lo;
                L4 = ok ? cand4 : L4;
                eL = ok ? e : eL;
            }
        }
        L4 = min(L4, 4 * (NL - 1));  // (empty streams / idle lanes: stay inside the vector)
        Lprev4[rd] = L4;
        const int L = L4 >> 2;
        int rem = lo - eL;
        // the owner lane's words + the two behind them (the next member is nearly always there: one set bit per ~12 places)
        uint32_t wv[WPL + 2];
        lds_read_words64<WPL + 2>(bits_s + L * WPL, wv);
        // the word holding member `rem` of these words, and the two words behind it
        int run = 0, sub = 0, wi = 0;
        uint32_t w = wv[0], n1 = wv[1], n2 = wv[2];
#pragma unroll
        for (int i = 0; i < WPL - 1; i++) {
            run += __popc(wv[i]);
            const bool ge = rem >= run;
            w = ge ? wv[i + 1] : w;
            n1 = ge ? wv[i + 2] : n1;
            n2 = ge ? wv[i + 3] : n2;
            sub = ge ? run : sub;
            wi = ge ? i + 1 : wi;
        }
        rem -= sub;
        const uint32_t wsel = w;
        int pos = 0;
        if (lut) {
            // the half, the byte, then the byte's table
#pragma unroll
            for (int s = 16; s > 4; s >>= 1) {
                const int cw = __popc(w & ((1u << s) - 1u));
                const bool g2 = rem >= cw;
                rem -= g2 ? cw : 0;
                w = g2 ? (w >> s) : w;
                pos += g2 ? s : 0;
            }
            pos += (int)((lut[w & 0xFFu] >> (3 * rem)) & 7u);
        } else {
#pragma unroll
            for (int s = 16; s > 0; s >>= 1) {
                const int cw = __popc(w & ((1u << s) - 1u));
                const bool g2 = rem >= cw;
                rem -= g2 ? cw : 0;
                w = g2 ? (w >> s) : w;
                pos += g2 ? s : 0;
            }
        }
        const int wbase = L * WPL * 32 + wi * 32;  // place of bit 0 of the word
        const int lrank = wbase + pos;
        // the next member: above `pos` in the same word, else the first one of the two words behind (a gap of 65+ places has a
        // chance of ~1e-3 per lane at one member per 12 places: the wave-uniform branch below is rarely taken)
        const uint32_t rest = (wsel >> pos) >> 1;
        int hrank = rest ? lrank + __ffs((int)rest) : (n1 ? wbase + 31 + __ffs((int)n1) : wbase + 63 + __ffs((int)n2));
        const bool found = (rest | n1 | n2) != 0u;
        const bool need_slow = casek == 0 && !found && sid < 3;
        if (__any(need_slow)) {
            // first the rest of the words read above, ...
            int nw = 0;
            uint32_t nwv = 0u;
#pragma unroll
            for (int i = WPL + 1; i >= 3; i--) {
                const bool cnd = wv[i] != 0u && i > wi + 2;
                nw = cnd ? i : nw;
                nwv = cnd ? wv[i] : nwv;
            }
            int hr = L * WPL * 32 + nw * 32 + __ffs((int)nwv) - 1;
            const bool still = need_slow && nw == 0;
            if (__any(still)) {
                // ... then the first member of the next lane that holds any: that lane from the ballot of the non-empty lanes of
                // this stream, its first set bit from its words
                const unsigned long long balA = __ballot((clAB & FM) != 0), balB = __ballot(((clAB >> FB) & FM) != 0),
                                         balC = __ballot(P10 ? ((clAB >> 20) & FM) != 0 : clC != 0);
                const unsigned long long bal = sid == 0 ? balA : (sid == 1 ? balB : balC);
                const unsigned long long restm = L < 63 ? (bal >> (L + 1)) : 0ull;
                const int L2 = min(restm ? L + __ffsll(restm) : L, NL - 1);
                const uint32_t *wp2 = bits_s + L2 * WPL;
                uint32_t w2[WPL];
#pragma unroll
                for (int i = 0; i < WPL; i++) w2[i] = wp2[i];
                int hr2 = 0;
#pragma unroll
                for (int i = WPL - 1; i >= 0; i--) hr2 = w2[i] != 0u ? i * 32 + __ffs((int)w2[i]) - 1 : hr2;
                if (still) hr = L2 * WPL * 32 + hr2;
            }
            if (need_slow) hrank = hr;
        }
        if (rd == NR - 1) next_flips();  // the vectors have been read for the last time in this step
        const int rs0 = min(max(lrank, 0), TPAD - 1), rs1 = min(max(casek == 0 ? hrank : lrank, 0), TPAD - 1);
        R v, vhi;
        vals.at_ranks(rs0, rs1, v, vhi);
        if (DQM) { v = apply_corr<R>(v, inv, kind); vhi = apply_corr<R>(vhi, inv, kind); }
        R res;
        if (casek == 3) res = nan_of<R>();
        else if (casek != 0) res = v;
        else {
            const R diff = vhi - v;
            if (gamma >= (R)0.5) res = (R)((double)vhi - (double)diff * (1.0 - (double)gamma));
            else res = v + diff * gamma;
            if (res != res && act) flag_cell = true;  // inf - inf: the generic kernel redoes the cell
        }
        if (act) {
            if (!ref_pass) hq_c[o + QL * rd] = res;
            else af_c[o + QL * rd] = get_corr<R>(hq_prev[rd], res, kind);
        }
        if (keep && !ref_pass) { hqk_up(); hqk[0] = res; }
        }  // (rounds)
        if (DQM && vs && sid < 3 && k == 0) {
            if (arr == 0) scaling[(size_t)c * G + g] = mu;  // mean(hist), replaced by the factor in the ref pass
            else scaling[(size_t)c * G + g] = get_corr<R>(scaling[(size_t)c * G + g], mu, kind);
        }
    }
    if (keep && !ref_pass)
        for (int s_ = nA; s_ < 8; s_++) hqk_up();  // step 0 to the head of the queue
}

template <int EPT, bool P10, bool DQM, typename Vals, bool PRESUM = false, int STOP = 0, bool PADT = false, int NR = 1>
__device__ __forceinline__ void cell_group_quantiles3(const Vals &vals, const uint16_t *rank, uint32_t *bits_all, int nvalid, int lane, int wid,
                                                      const int *__restrict__ row_ptr, const int *__restrict__ row_sorted,
                                                      const int *__restrict__ tog64, int G, const typename Vals::value_type *__restrict__ q, int Q,
                                                      int kind, int mode, int arr, int64_t c, typename Vals::value_type *__restrict__ af,
                                                      typename Vals::value_type *out_hq, typename Vals::value_type *scaling, bool &flag_cell,
                                                      const double *gsum = nullptr, const double *guard = nullptr, int n_time = 0,
                                                      const uint32_t *lut = nullptr, typename Vals::value_type *hqk = nullptr)
{
    cell_group_quantiles3g<CellCfg3<EPT>, const uint16_t *, P10, DQM, Vals, PRESUM, STOP, PADT, NR>(
        vals, rank, bits_all, nvalid, lane, wid, row_ptr, row_sorted, tog64, G, q, Q, kind, mode, arr, c, af, out_hq, scaling, flag_cell, gsum, guard,
        n_time, lut, hqk);
}

template <int EPT, bool P10, bool DQM, int STOP = 0, int NR = 1>
__global__ __launch_bounds__(1024) void k_train_cell3(const float *__restrict__ x0, const float *__restrict__ x1, int64_t T, int64_t sc,
                                                       const int *__restrict__ row_ptr, const int *__restrict__ row_sorted, int G,
                                                       const float *__restrict__ q, int Q, int kind, int mode, float *__restrict__ af,
                                                       float *out_hq, float *scaling, int *__restrict__ flag_count,
                                                       int *__restrict__ flag_list, int64_t C, const int *__restrict__ tog64 /* padded with n_time, n_time < TPAD */,
                                                       const int *__restrict__ mem_ptr, const int *__restrict__ mem_idx, const int *__restrict__ mem_t16, int W)
{
    using Cfg = CellCfg<EPT>;
    using Cfg3 = CellCfg3<EPT>;
    constexpr int TPAD = Cfg::TPAD;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *keys = reinterpret_cast<uint32_t *>(smem + Cfg::off_keys);
    uint16_t *idx0 = reinterpret_cast<uint16_t *>(smem + Cfg::off_idx0), *idx1 = reinterpret_cast<uint16_t *>(smem + Cfg::off_idx1);
    uint32_t *bits_all = reinterpret_cast<uint32_t *>(smem + Cfg3::off_bits);
    uint32_t *red = reinterpret_cast<uint32_t *>(smem + Cfg3::off_misc);
    double *gsum = reinterpret_cast<double *>(smem + Cfg3::lds_bytes);  // DQM: [G] group sums + [1] guard (the launch adds (G + 1) * 8 bytes)
    const int tid = threadIdx.x, wid = tid >> 6;
    uint32_t *lut = reinterpret_cast<uint32_t *>(smem + Cfg3::off_lut);
    byte_select_lut(lut, tid);  // (the first barrier of the first load phase comes before any use)
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        const int npass_arr = mode == 0 ? 1 : 2;
        bool flag_cell = false;
        float hqk[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // the hist quantiles of this lane's steps, kept for the ref pass
        for (int arr = 0; arr < npass_arr; arr++) {
            int lane = tid & 63;
            __asm__ volatile("" : "+v"(lane));
            const float *x = ((mode == 0 || arr == 1) ? x0 : x1) + c * sc;  // hist first, then ref (see k_train_cell)
            int nvalid;
            const uint16_t *sorted;  // (no rank[] in the union region)
            // (stop_after 3: rank[] is written by the sort's last pass into the buffer that pass read from)
            // (NaN / padding steps get the rank NAN_RANK: a bit of the scratch word behind every bit-vector)
            if constexpr (DQM) sorted = cell_load_sort<EPT>(x, T, smem, red, 3, nvalid, GroupSumHook<EPT>{smem, gsum, gsum + G, mem_ptr, mem_idx, mem_t16, G, W}, Cfg3::NAN_RANK, true);
            else sorted = cell_load_sort<EPT>(x, T, smem, red, STOP == 1 ? 1 : 3, nvalid, NoLoadHook(), Cfg3::NAN_RANK, true);
            if constexpr (STOP == 1) continue;  // (diagnostic: the load alone)
            if (!sorted) {
                // no valid value in this series (block-uniform): every quantile of the pass is NaN — written with the very
                // expressions of the walk (NaN quantile; af = correction(hist_q, NaN); the DQM means and their correction)
                const float nanq = nan_of<float>();
                const size_t base = (size_t)c * G * (size_t)Q;
                for (int i = tid; i < G * Q; i += 1024) {
                    if (mode == 0 || arr == 0) out_hq[base + i] = nanq;
                    else af[base + i] = get_corr<float>(out_hq[base + i], nanq, kind);
                }
                if constexpr (DQM)
                    for (int g = tid; g < G; g += 1024) {
                        float &s_ = scaling[(size_t)c * G + g];
                        s_ = arr == 0 ? nanq : get_corr<float>(s_, nanq, kind);
                    }
#pragma unroll
                for (int j = 0; j < 8; j++) hqk[j] = nanq;  // (what the walk of a first pass would have kept)
                __syncthreads();
                continue;
            }
            if constexpr (STOP == 2) { if (tid == 0 && sorted[nvalid / 2] == 0xFFFF) out_hq[0] = 1.f; __syncthreads(); continue; }  // (load + sort)
            uint16_t *rank = sorted == idx0 ? idx1 : idx0;  // the buffer the sort left free
            const LdsVals vals{keys, sorted, TPAD};
            cell_group_quantiles3<EPT, P10, DQM, LdsVals, DQM, STOP, true, NR>(vals, rank, bits_all, nvalid, lane, wid, row_ptr, row_sorted, tog64, G, q, Q,
                                                                     kind, mode, arr, c, af, out_hq, scaling, flag_cell, gsum, gsum + G, (int)T, lut, hqk);
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
