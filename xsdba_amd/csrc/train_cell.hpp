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

// wave total of a double on the DPP cross-lane path (the steps of wave_incl_scan on both halves of the value; lanes without
// a source add +0.0), read from lane 63 into scalar registers: wave-uniform
__device__ __forceinline__ double wave_total_dpp(double v)
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
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// Sum of the per-wave totals tot[0 .. wid) of a 16-wave block: every lane reads one total, a row scan on the DPP path, one
// readlane — instead of a loop of up to 15 dependent LDS reads in the last wave, on which all the others wait at the barrier.
__device__ __forceinline__ int waves_before(const uint32_t *tot, int lane, int wid)
{
    int v = (int)tot[lane & 15];
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    const int w = __builtin_amdgcn_readfirstlane(wid);
    const int x = __builtin_amdgcn_readlane(v, (w + 15) & 15);
    return w ? x : 0;
}

__device__ __forceinline__ void lds_wave_fence() { __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// NW words (NW even) from an 8-byte aligned LDS address as ds_read_b64 + their wait, in ONE asm statement: left to itself the
// compiler fuses neighbouring reads into ds_read2_b32 / ds_read2_b64, which take four times the LDS cycles of ds_read_b64 on
// gfx950 (scripts/ubench/issue_rate.hip: 27 vs 6.8 cycles per wave-instruction at a lane stride of 24 bytes).
template <int NW> __device__ __forceinline__ void lds_read_words64(const uint32_t *p, uint32_t (&w)[NW])
{
    static_assert(NW == 2 || NW == 4 || NW == 6 || NW == 8 || NW == 16, "1 to 4, or 8, ds_read_b64");
    const uint32_t a = (uint32_t)(uintptr_t)p;  // (an LDS pointer: the low half of the flat address is the LDS offset)
    uint64_t v0, v1, v2, v3;
    if constexpr (NW == 2) __asm__ volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v0) : "v"(a) : "memory");
    if constexpr (NW == 4) __asm__ volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:8\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v0), "=&v"(v1) : "v"(a) : "memory");
    if constexpr (NW == 6)
        __asm__ volatile("ds_read_b64 %0, %3\n\tds_read_b64 %1, %3 offset:8\n\tds_read_b64 %2, %3 offset:16\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2) : "v"(a) : "memory");
    if constexpr (NW == 8)
        __asm__ volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:8\n\tds_read_b64 %2, %4 offset:16\n\tds_read_b64 %3, %4 offset:24\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(a) : "memory");
    if constexpr (NW == 16) {
        uint64_t v4, v5, v6, v7;
        __asm__ volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:8\n\tds_read_b64 %2, %8 offset:16\n\tds_read_b64 %3, %8 offset:24\n\t"
                         "ds_read_b64 %4, %8 offset:32\n\tds_read_b64 %5, %8 offset:40\n\tds_read_b64 %6, %8 offset:48\n\tds_read_b64 %7, %8 offset:56\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(a) : "memory");
        const uint64_t vv[8] = {v0, v1, v2, v3, v4, v5, v6, v7};
#pragma unroll
        for (int i = 0; i < 8; i++) { w[2 * i] = (uint32_t)vv[i]; w[2 * i + 1] = (uint32_t)(vv[i] >> 32); }
        return;
    }
    w[0] = (uint32_t)v0; w[1] = (uint32_t)(v0 >> 32);
    if constexpr (NW >= 4) { w[2] = (uint32_t)v1; w[3] = (uint32_t)(v1 >> 32); }
    if constexpr (NW >= 6) { w[4] = (uint32_t)v2; w[5] = (uint32_t)(v2 >> 32); }
    if constexpr (NW >= 8) { w[6] = (uint32_t)v3; w[7] = (uint32_t)(v3 >> 32); }
}



template <int EPT> struct CellCfg {
    static constexpr int NT = 1024;
    static constexpr int TPAD = EPT * NT;
    static constexpr int NWORDS = TPAD / 32;
    static constexpr int WPL = (NWORDS + 63) / 64;  // bit-vector words per lane
    static constexpr int BITW = WPL * 64;
    static constexpr int NWAVE = NT / 64;
    static constexpr int WROW = 256 + 8;             // dwords per wave-private counter row of the sort's stable passes (<= 256
                                                     // digits; + 8: the scan's reads across the rows spread over the banks)
    static constexpr size_t off_keys = 0;
    static constexpr size_t off_idx0 = off_keys + (size_t)TPAD * 4;
    static constexpr size_t off_idx1 = off_idx0 + (size_t)TPAD * 2;
    static constexpr size_t off_union = off_idx1 + (size_t)TPAD * 2;
    // sort view of the union: the first pass's histogram (2^14 u16 bins), behind it the stable passes' counter rows
    static constexpr size_t off_hist = off_union;
    static constexpr size_t off_wcnt = off_hist + (size_t)(1 << 14) * 2;
    static constexpr size_t sort_end = off_wcnt + (size_t)NWAVE * WROW * 4;
    // select view of the union
    static constexpr size_t off_rank = off_union;
    static constexpr size_t off_bits = off_rank + (size_t)TPAD * 2;
    static constexpr size_t sel_end = off_bits + (size_t)2 * NWAVE * BITW * 4;  // two group streams per wave
    static constexpr size_t off_misc = ((sort_end > sel_end ? sort_end : sel_end) + 15) & ~(size_t)15;
    static constexpr size_t lds_bytes = off_misc + 64 * 8;  // reduction scratch
};

// Phase B: LSD radix sort of the time indexes by the keys in LDS (valid keys lie in [kmin, kmin + range], NaN / padding
// are 0xFFFFFFFF), then rank[t] = position of t.  Returns the sorted index array (the upper u16 half of the handle buffer;
// the lower half is free afterwards).  stop_after == 2: no rank[]; 3: rank[] goes into that lower half, with `nan_rank`
// (when >= 0) in place of the positions >= nvalid: the three-stream kernel points NaN / padding steps at a scratch word of
// its bit-vectors and toggles without looking at nvalid; else rank[] goes to off_rank.
//
// What is sorted are 32-bit HANDLES, (r >> 14) << 14 | time index with r = key - kmin (NaN / padding: range + 1): the time
// index needs 14 bits (TPAD <= 16384), the 18 bits above it hold every key bit the first pass has not consumed — so the
// passes after the first never go back to keys[] (eleven random LDS reads per thread and pass in round 3's sort) and read
// their input as consecutive dwords.
//   pass 1: the 14 LOW bits in one go.  The first LSD pass need not be stable (its input order carries no information): a
//           counting sort with LDS atomics — 16384 u16 bins packed two per dword, histogram by ds_add (fused into the
//           series' load by cell_load_sort), exclusive scan (16 bins per thread), slots by returning ds_add;
//   then:   the remaining R = nbits - 14 bits in ceil(R / 8) STABLE counting passes of up to 8 bits.  Wave w owns the 64 EPT
//           consecutive places [w 64 EPT, (w + 1) 64 EPT) and takes them in EPT rounds of 64 consecutive places, one per lane;
//           a returning ds_add on a counter row PRIVATE to the wave gives every element the number of equal digits in front
//           of it inside the wave's range — the LDS serves the lanes of one instruction that hit the same word in ascending
//           lane order, and the instructions of a wave in program order (scripts/ubench/atomic_order.hip: 1.4e11 returning
//           atomics over every kind of lane -> address map, 16 waves at once, not one out of order), which is exactly
//           stability.  A scan over (digit, wave) turns the rows into base slots; slot = base + that number.
//           Round 3's stable passes took 4 bits each with per-thread packed-nibble counts (410 VALU instructions per thread
//           and pass, two passes for the 7 bits left of a temperature series); one 7-bit pass now, the handles in registers
//           across it, so that it sorts in place.
#ifndef XS_SORT_STOP
#define XS_SORT_STOP 0  // (scripts/ubench/train3_bench.hip builds with 1..6 to time the sort's phases; the library never does)
#endif
// fused: the load phase has already counted the valid keys into the first pass's histogram — on the low 14 bits of the KEYS,
// which are the digits of (key - kmin) when kmin is a multiple of 2^14 (cell_load_sort rounds it down).
template <int EPT>
__device__ __forceinline__ const uint16_t *radix_sort_indexes(unsigned char *smem, uint32_t *red, int tid, int nvalid,
                                                              uint32_t kmin, uint32_t range, int stop_after, bool hist_zeroed = false,
                                                              int nan_rank = -1, bool fused = false)
{
    using Cfg = CellCfg<EPT>;
    constexpr int NT = Cfg::NT, TPAD = Cfg::TPAD;
    static_assert(TPAD <= (1 << 14), "a handle keeps the time index in 14 bits");
    const uint32_t *keys = reinterpret_cast<const uint32_t *>(smem + Cfg::off_keys);
    uint32_t *H = reinterpret_cast<uint32_t *>(smem + Cfg::off_idx0);       // [TPAD] handles (the two u16 buffers' space)
    uint16_t *idx0 = reinterpret_cast<uint16_t *>(smem + Cfg::off_idx0);
    uint16_t *idx1 = reinterpret_cast<uint16_t *>(smem + Cfg::off_idx1);
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem + Cfg::off_hist);    // pass 1: 16384 u16 bins
    uint32_t *wcnt = reinterpret_cast<uint32_t *>(smem + Cfg::off_wcnt);    // later passes: [16][WROW] per-wave counters
    uint16_t *rank = stop_after == 3 ? idx0 : reinterpret_cast<uint16_t *>(smem + Cfg::off_rank);
    uint16_t *sorted = idx1;
    const int lane = tid & 63, wid = tid >> 6;
    static_assert(Cfg::NWAVE == 16, "waves_before() scans one DPP row of per-wave totals");
    constexpr int D1 = 14, NB1 = 1 << D1, WROW = Cfg::WROW;
    const int nbits = 32 - __clz((int)(range + 1u) | 0);
    const int R = nbits > D1 ? nbits - D1 : 0;           // key bits left after pass 1 (<= 18)
    const int npass_rest = (R + 7) >> 3;                 // 0..3
    const int db = npass_rest ? (R + npass_rest - 1) / npass_rest : 0;  // digit bits of the stable passes (<= 8)
    const int ND = 1 << db;
    uint32_t *wrow = wcnt + wid * WROW;                  // this wave's counter row
    if (npass_rest)
        for (int i = lane; i < ND; i += 64) wrow[i] = 0u;  // (only this wave touches the row before the first scan)
    // ---------------- pass 1 ----------------
    const uint32_t d_nan = (range + 1u) & (NB1 - 1);  // the digit of NaN / padding keys
    constexpr int NL = (EPT + 1) / 2;                 // elements in the load's order: the pair 2 tid + j 2 NT, j < NL
    if (!fused) {
        if (!hist_zeroed) {
            uint4 *hz = reinterpret_cast<uint4 *>(hist);
            for (int i = tid; i < NB1 / 8; i += NT) hz[i] = make_uint4(0u, 0u, 0u, 0u);
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < NL; j++) {
            const int i = 2 * tid + j * 2 * NT;
            if ((j + 1) * 2 * NT <= TPAD || i < TPAD) {
                const uint2 kk = *reinterpret_cast<const uint2 *>(keys + i);
                const uint32_t d0 = (kk.x == 0xFFFFFFFFu) ? d_nan : ((kk.x - kmin) & (NB1 - 1));
                const uint32_t d1 = (kk.y == 0xFFFFFFFFu) ? d_nan : ((kk.y - kmin) & (NB1 - 1));
                atomicAdd(&hist[d0 >> 1], 1u << (16 * (d0 & 1u)));
                atomicAdd(&hist[d1 >> 1], 1u << (16 * (d1 & 1u)));
            }
        }
        __syncthreads();
    }
    if constexpr (XS_SORT_STOP == 1) return idx1;
    // Heavily tied series (zero-inflated precipitation, coarsely rounded values): after pass 1 equal keys are neighbours, so that all
    // 64 lanes of a stable pass's atomic hit ONE counter and are served one after the other — the sort of such a series took
    // twice its time.  Detected from pass 1's histogram (block-uniform, for the price of one compare per bin); the stable passes
    // then hand a wave whose 64 digits are equal its 64 slots with one atomic.
    bool heavy = false;
    {
        constexpr int BPT = NB1 / NT;  // 16 bins = 8 dwords per thread
        uint32_t wv[BPT / 2];
        int run = 0;
        uint32_t bigm = 0u;  // nonzero: one of this thread's bins holds 256 keys or more — a heavily tied series
#pragma unroll
        for (int i = 0; i < BPT / 2; i++) {
            wv[i] = hist[tid * (BPT / 2) + i];
            bigm |= wv[i] & 0xFF00FF00u;  // (the NaN / padding keys are not in yet when the histogram came with the load; else they may
                                          //  switch the aggregation on, which is right for any series)
            // fused: the NaN / padding keys were not counted by the load (their digit needs the range): the owner of their bin adds them
            if (fused && (int)(d_nan >> 4) == tid && (int)((d_nan & 15u) >> 1) == i) wv[i] += (uint32_t)(TPAD - nvalid) << (16 * (d_nan & 1u));
            run += (int)(wv[i] & 0xFFFFu) + (int)(wv[i] >> 16);
        }
        int incl = wave_incl_scan(run);
        // (words 64.. of the scratch: with the histogram fused into the load no barrier lies between the all-reduce of
        //  cell_load_sort, which reads words 0 .. 47, and this store — a slow wave may still be reading them)
        const bool wbig = __any(bigm != 0u);
        if (lane == 63) { red[64 + wid] = (uint32_t)incl; red[80 + wid] = wbig ? 1u : 0u; }
        __syncthreads();
        heavy = __any(red[80 + (lane & 15)] != 0u);
        const int woff = waves_before(red + 64, lane, wid);
        int pre = woff + incl - run;  // exclusive prefix of this thread's first bin
#pragma unroll
        for (int i = 0; i < BPT / 2; i++) {
            int c0 = (int)(wv[i] & 0xFFFFu), c1 = (int)(wv[i] >> 16);
            hist[tid * (BPT / 2) + i] = (uint32_t)pre | ((uint32_t)(pre + c0) << 16);
            pre += c0 + c1;
        }
    }
    __syncthreads();
    if constexpr (XS_SORT_STOP == 2) return idx1;
    {
        // slots by returning atomics; what is stored: the handle, or — narrow keys, no further pass — the final arrays
        const bool last = npass_rest == 0;
        const bool rank_here = last && stop_after == 3;   // (rank[] elsewhere overlays the histogram: a pass of its own below)
#pragma unroll
        for (int j = 0; j < NL; j++) {
            const int i = 2 * tid + j * 2 * NT;
            if ((j + 1) * 2 * NT <= TPAD || i < TPAD) {
                const uint2 kk = *reinterpret_cast<const uint2 *>(keys + i);
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const uint32_t k = h ? kk.y : kk.x;
                    const uint32_t r = (k == 0xFFFFFFFFu) ? range + 1u : k - kmin;
                    const uint32_t d = r & (NB1 - 1);
                    const int sh = 16 * (d & 1);
                    const uint32_t old = atomicAdd(&hist[d >> 1], 1u << sh);
                    const int pos = (int)((old >> sh) & 0xFFFFu);
                    if (!last) H[pos] = ((r >> D1) << D1) | (uint32_t)(i + h);
                    else {
                        sorted[pos] = (uint16_t)(i + h);
                        if (rank_here) rank[i + h] = (uint16_t)((nan_rank >= 0 && pos >= nvalid) ? nan_rank : pos);
                    }
                }
            }
        }
        __syncthreads();
    }
    if constexpr (XS_SORT_STOP == 3) return idx1;
    // ---------------- the stable passes ----------------
    for (int p = 0; p < npass_rest; p++) {
        const int shift = D1 + db * p;
        const bool last = p == npass_rest - 1;
        uint32_t hd[EPT], od[EPT];  // handles and, per element, the number of equal digits in front of it in this wave's range
        {
            const uint32_t *src = H + wid * (64 * EPT) + lane;
#pragma unroll
            for (int j = 0; j < EPT; j++) hd[j] = src[64 * j];
            if (heavy) {
#pragma unroll
                for (int j = 0; j < EPT; j++) {
                    const uint32_t d = (hd[j] >> shift) & (uint32_t)(ND - 1);
                    const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
                    if (__all(d == d0)) {  // (the lanes in lane order, like the atomics of one instruction on one word)
                        uint32_t b = 0u;
                        if (lane == 0) b = atomicAdd(&wrow[d0], 64u);
                        od[j] = (uint32_t)__builtin_amdgcn_readfirstlane((int)b) + (uint32_t)lane;
                    } else od[j] = atomicAdd(&wrow[d], 1u);
                }
            } else {
#pragma unroll
                for (int j = 0; j < EPT; j++) od[j] = atomicAdd(&wrow[(hd[j] >> shift) & (uint32_t)(ND - 1)], 1u);
            }
        }
        __syncthreads();
        if constexpr (XS_SORT_STOP == 4) { if (p == 0) return idx1; }
        {   // scan over the counters in digit-major order (digit d of wave w at d 16 + w): thread t owns `per` consecutive ones
            const int NDW = ND << 4, per = NDW > NT ? NDW >> 10 : 1;  // (1, 2 or 4)
            uint32_t c[4];
            int run = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int i = tid * per + k;
                c[k] = (k < per && i < NDW) ? wcnt[(i & 15) * WROW + (i >> 4)] : 0u;
                run += (int)c[k];
            }
            const int incl = wave_incl_scan(run);
            if (lane == 63) red[wid] = (uint32_t)incl;
            __syncthreads();
            int pre = waves_before(red, lane, wid) + incl - run;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int i = tid * per + k;
                if (k < per && i < NDW) wcnt[(i & 15) * WROW + (i >> 4)] = (uint32_t)pre;
                pre += (int)c[k];
            }
        }
        __syncthreads();
        if constexpr (XS_SORT_STOP == 5) { if (p == 0) return idx1; }
        {
            uint32_t bs[EPT];
#pragma unroll
            for (int j = 0; j < EPT; j++) bs[j] = wrow[(hd[j] >> shift) & (uint32_t)(ND - 1)];
            if (!last) {
#pragma unroll
                for (int j = 0; j < EPT; j++) H[bs[j] + od[j]] = hd[j];  // in place: every handle sits in a register
                for (int i = lane; i < ND; i += 64) wrow[i] = 0u;         // (this wave's row, behind its own reads: the next pass)
            } else {
                // the last pass leaves the final arrays where the handles were (nobody reads those any more)
#pragma unroll
                for (int j = 0; j < EPT; j++) {
                    const int pos = (int)(bs[j] + od[j]);
                    const uint32_t t = hd[j] & (uint32_t)((1 << D1) - 1);
                    sorted[pos] = (uint16_t)t;
                    if (stop_after != 2) rank[t] = (uint16_t)((nan_rank >= 0 && pos >= nvalid) ? nan_rank : pos);
                }
            }
        }
        __syncthreads();
        if constexpr (XS_SORT_STOP == 6) { if (p == 0) return idx1; }
    }
    if (npass_rest == 0 && stop_after != 2 && stop_after != 3) {  // narrow keys, rank[] behind the histogram's space
        for (int r = tid; r < TPAD; r += NT) rank[sorted[r]] = (uint16_t)r;
        __syncthreads();
    }
    return sorted;
}

// Phases A + B, shared by k_train_cell and k_qdm_cell: the cell's series -> order-preserving keys in LDS, the time
// indexes sorted by key (NaN / padding last), rank[t] = position of t in that order.  Returns the sorted index array
// (one of the two u16 buffers; the other one is free afterwards).  All 1024 threads must call it; it ends on a barrier.
// stop_after (timing aid): 1 = return after the load, 2 = after the sort without building rank[].
// skip_empty: return nullptr (after the load, nvalid_out = 0) instead of sorting a series that holds no valid value.
struct NoLoadHook {
    static constexpr bool uses_sort_scratch = false;
    __device__ __forceinline__ void operator()(const uint32_t *, int, int) const {}
};

// `hook(keys, T, tid)` runs between the load and the sort: the keys of the series are in LDS, everything from off_idx0 up
// to off_misc is still free (the DQM group sums use it, train_cell3.hpp); it must end on a block barrier if it touches LDS
template <int EPT, typename Hook = NoLoadHook>
__device__ __forceinline__ const uint16_t *cell_load_sort(const float *__restrict__ x, int64_t T, unsigned char *smem,
                                                          uint32_t *red, int stop_after, int &nvalid_out, const Hook &hook = Hook(),
                                                          int nan_rank = -1, bool skip_empty = false)
{
    using Cfg = CellCfg<EPT>;
    constexpr int NT = Cfg::NT, TPAD = Cfg::TPAD;
    uint32_t *keys = reinterpret_cast<uint32_t *>(smem + Cfg::off_keys);
    // the thread index is made opaque per call: everything derived from it (addresses, element indexes) is then
    // recomputed here instead of being hoisted out of the caller's cell loop and spilled to scratch across it
    int tid = threadIdx.x;
    __asm__ volatile("" : "+v"(tid));
    const int lane = tid & 63, wid = tid >> 6;
        // ---------------- A. load, keys, range ----------------
        int nv = 0;
        uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
        bool fused = false;                 // the load has counted the histogram of the sort's first pass
        const int Ti = (int)T;  // <= TPAD
        if ((reinterpret_cast<uintptr_t>(x) & 7) == 0) {  // 8-byte aligned rows: two elements per load
            // all the loads of the series are issued before the first one is consumed (one HBM latency per array
            // instead of one per 2048 elements).  They are unconditional — addresses clamped to the last full pair,
            // out-of-range results replaced by NaN afterwards — so that nothing orders them behind a branch.
            constexpr int NL = (EPT + 1) / 2;
            const int Te = Ti & ~1;                 // elements covered by whole pairs (T >= 64)
            const float xlast = x[Ti - 1];          // the odd tail element, uniform address
            float2 v2[NL];
#pragma unroll
            for (int j = 0; j < NL; j++) {
                const int i = 2 * tid + j * 2 * NT;
                v2[j] = *reinterpret_cast<const float2 *>(x + min(i, Te - 2));
            }
            if constexpr (!Hook::uses_sort_scratch) {  // the histogram of the sort's first pass, cleared while the loads are in flight
                uint4 *hz = reinterpret_cast<uint4 *>(smem + Cfg::off_hist);
                for (int i = tid; i < (1 << 14) / 8; i += NT) hz[i] = make_uint4(0u, 0u, 0u, 0u);
                __syncthreads();  // ... and counted into as the loads arrive (see radix_sort_indexes: fused)
                fused = true;
            }
            uint32_t *hist1 = reinterpret_cast<uint32_t *>(smem + Cfg::off_hist);
            // keys and their statistics in integer arithmetic (to_key() through compares and selects costs three times the
            // instructions, and selects on VCC stall each other: scripts/ubench/issue_rate.hip): `ok` is all ones for a number
            // and 0 for a NaN; a NaN's key is 0xFFFFFFFF — no effect on the minimum — and it enters the maximum as 0
            auto key_of = [&](float f) {
                const uint32_t u = __float_as_uint(f);
                const uint32_t ok = (uint32_t)((int32_t)((u & 0x7FFFFFFFu) - 0x7F800001u) >> 31);
                const uint32_t vk = (u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u)) & ok;  // = to_key(f) for a number, 0 for a NaN
                nv -= (int)ok;
                kmax = max(kmax, vk);
                const uint32_t k = vk | ~ok;
                kmin = min(kmin, k);
                return k;
            };
#pragma unroll
            for (int j = 0; j < NL; j++) {
                const int i = 2 * tid + j * 2 * NT;
                if (i < TPAD) {
                    float a = v2[j].x, b = v2[j].y;
                    if ((j + 1) * 2 * NT > Te) {  // (wave-uniform: only the last rounds can reach past the series)
                        if (i >= Te) { a = (i == Ti - 1) ? xlast : __builtin_nanf(""); b = __builtin_nanf(""); }
                    }
                    const uint32_t k0 = key_of(a), k1 = key_of(b);
                    *reinterpret_cast<uint2 *>(keys + i) = make_uint2(k0, k1);
                    if constexpr (!Hook::uses_sort_scratch) {
                        // the first pass of the sort counts its 14-bit digits here: no second walk over the keys, no barrier of its own
                        const uint32_t d0 = k0 & 0x3FFFu, d1 = k1 & 0x3FFFu;
                        if (k0 != 0xFFFFFFFFu) atomicAdd(&hist1[d0 >> 1], 1u << (16 * (d0 & 1u)));
                        if (k1 != 0xFFFFFFFFu) atomicAdd(&hist1[d1 >> 1], 1u << (16 * (d1 & 1u)));
                    }
                }
            }
        } else {
            for (int i = tid; i < TPAD; i += NT) {
                uint32_t k = i < Ti ? to_key(x[i]) : 0xFFFFFFFFu;
                keys[i] = k;
                if (k != 0xFFFFFFFFu) { nv++; kmin = min(kmin, k); kmax = max(kmax, k); }
            }
            if constexpr (!Hook::uses_sort_scratch) {
                uint4 *hz = reinterpret_cast<uint4 *>(smem + Cfg::off_hist);
                for (int i = tid; i < (1 << 14) / 8; i += NT) hz[i] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
        // wave reduction on the DPP path (the steps of wave_incl_scan; lanes without a source see the identity): lane 63 holds
        // the wave's count / min / max — no LDS crossbar round trips in front of the barrier
#define XS_RED_STEP(ctrl, rmask)                                                                              \
        nv += __builtin_amdgcn_update_dpp(0, nv, ctrl, rmask, 0xf, false);                                    \
        kmin = min(kmin, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)kmin, ctrl, rmask, 0xf, false));      \
        kmax = max(kmax, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)kmax, ctrl, rmask, 0xf, false));
        XS_RED_STEP(0x111, 0xf) XS_RED_STEP(0x112, 0xf) XS_RED_STEP(0x114, 0xf) XS_RED_STEP(0x118, 0xf)
        XS_RED_STEP(0x142, 0xa) XS_RED_STEP(0x143, 0xc)
#undef XS_RED_STEP
        if (lane == 63) { red[3 * wid] = (uint32_t)nv; red[3 * wid + 1] = kmin; red[3 * wid + 2] = kmax; }
        __syncthreads();
        {   // all-reduce over the 16 waves: a lane per wave (in every row of 16), rotations inside the row
            static_assert(Cfg::NWAVE == 16, "one DPP row holds the per-wave results");
            nv = (int)red[3 * (lane & 15)]; kmin = red[3 * (lane & 15) + 1]; kmax = red[3 * (lane & 15) + 2];
#define XS_ROW_ROR(x, n) __builtin_amdgcn_update_dpp((int)(x), (int)(x), 0x120 + (n), 0xf, 0xf, false)
            nv += XS_ROW_ROR(nv, 8); kmin = min(kmin, (uint32_t)XS_ROW_ROR(kmin, 8)); kmax = max(kmax, (uint32_t)XS_ROW_ROR(kmax, 8));
            nv += XS_ROW_ROR(nv, 4); kmin = min(kmin, (uint32_t)XS_ROW_ROR(kmin, 4)); kmax = max(kmax, (uint32_t)XS_ROW_ROR(kmax, 4));
            nv += XS_ROW_ROR(nv, 2); kmin = min(kmin, (uint32_t)XS_ROW_ROR(kmin, 2)); kmax = max(kmax, (uint32_t)XS_ROW_ROR(kmax, 2));
            nv += XS_ROW_ROR(nv, 1); kmin = min(kmin, (uint32_t)XS_ROW_ROR(kmin, 1)); kmax = max(kmax, (uint32_t)XS_ROW_ROR(kmax, 1));
#undef XS_ROW_ROR
            // (equal in every lane: back into scalar registers, the pass count and the shifts of the sort derive from them)
            nv = __builtin_amdgcn_readfirstlane(nv);
            kmin = (uint32_t)__builtin_amdgcn_readfirstlane((int)kmin);
            kmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)kmax);
        }
        const int nvalid = nv;
        nvalid_out = nv;
        if (fused) kmin &= ~0x3FFFu;  // the digits counted by the load are those of (key - kmin) only for such a kmin
        const uint32_t range = nvalid ? kmax - kmin : 0u;  // NaN / padding are remapped to range + 1
        // red[] is next written behind the first barrier of the sort, which no wave passes before every wave has read it here:
        // a barrier of its own only when a hook (or the timing aid's early return) comes in between
        if (Hook::uses_sort_scratch || stop_after == 1) __syncthreads();

        if (stop_after == 1) return nullptr;  // timing aid (XSDBA_HIP_DEBUG_PHASE): stop after load/keys
        // skip_empty: a series without a single valid value (an ocean cell of a land-only variable, a masked cell) is not sorted
        // — every key is the same, the counting pass would serialise 11 000 atomics on one LDS word —: the caller gets nullptr
        // (wave-uniform) and writes its all-NaN results directly.  keys[] is complete; the barrier protects red[].
        if (skip_empty && nvalid == 0) { __syncthreads(); return nullptr; }
        hook(keys, Ti, tid);
        return radix_sort_indexes<EPT>(smem, red, tid, nvalid, kmin, range, stop_after, !Hook::uses_sort_scratch, nan_rank, fused);
}

// Where phase C finds values: the series' keys and the sorted index array in LDS (k_train_cell); train_long.hpp has
// the global-memory counterpart
struct LdsVals {
    using value_type = float;
    const uint32_t *keys;      // [TPAD] order-preserving keys by time index
    const uint16_t *sorted;    // [TPAD] time index by rank
    int tpad;
    __device__ __forceinline__ float at_time(int t) const { return from_key(keys[t]); }
    // both neighbours are looked up together: two independent (index, key) LDS round trips in flight
    __device__ __forceinline__ void at_ranks(int r0, int r1, float &v0, float &v1) const
    {
        const int id0 = min((int)sorted[r0], tpad - 1), id1 = min((int)sorted[r1], tpad - 1);
        v0 = from_key(keys[id0]);
        v1 = from_key(keys[id1]);
    }
};
// Phase C of the per-cell training kernels: the quantiles of every group of one series from rank[] (LDS, rank of each
// time step in the sorted order, >= nvalid for NaN) — membership bit-vectors in rank space + select.
// Geo: TPAD / WPL / BITW / NWAVE of the vectors.  Called by all 1024 threads; no block-level barrier inside.
// the padded toggle table a geometry walks has 64 * togn_of<Geo>() columns (Geo::TOGN when the geometry names it: the long
// series of train_long.hpp toggle up to 128 steps per group)
template <typename Geo, typename = void> struct togn_of { static constexpr int value = 1; };
template <typename Geo> struct togn_of<Geo, std::void_t<decltype(Geo::TOGN)>> { static constexpr int value = Geo::TOGN; };

// (RankT: `const uint16_t *` into the LDS, or train_xlong.hpp's view of a rank table in global memory)
template <typename Geo, bool DIAG, bool DQM, bool TOG64, typename Vals, typename Stamp, typename RankT>
__device__ __forceinline__ void cell_group_quantiles(
    const Vals &vals, const RankT rank, uint32_t *bits_all, int nvalid, int lane, int wid,
    const int *__restrict__ row_ptr, const int *__restrict__ row_sorted, const int *__restrict__ tog_ptr,
    const int *__restrict__ tog_idx, const int *__restrict__ tog64, int G, const typename Vals::value_type *__restrict__ q, int Q, int kind,
    int mode, int arr, int64_t c, typename Vals::value_type *__restrict__ af, typename Vals::value_type *out_hq,
    typename Vals::value_type *scaling, bool &flag_cell,
    int debug_phase, Stamp &stamp)
{
    using R = typename Vals::value_type;  // float or double series
    constexpr int TPAD = Geo::TPAD, WPL = Geo::WPL, BITW = Geo::BITW;
        // ---------------- C. membership bit-vectors in rank space + select ----------------
        // Every wave owns a contiguous range [r0, r1) of groups and walks it OUTWARD from its middle gc with two
        // bit-vectors in lock-step, one per half-wave: lanes 0-31 (stream A) take gc, gc+1, ... upward, lanes 32-63
        // (stream B) take gc-1, gc-2, ... downward.  The vector of group gc is built once from its row list and
        // copied to B (XOR lists work in both directions), which halves the number of builds.  Lane k of a half
        // serves quantile k: it selects the lower neighbour, then finds the upper one as the next set bit.
        const int gpw = (G + Geo::NWAVE - 1) / Geo::NWAVE;
        const int half = lane >> 5;
        const unsigned hshift = 16u * (unsigned)half;
        const int r0 = min(G, wid * gpw), r1 = min(G, r0 + gpw);
        const int gc = r0 + (r1 - r0) / 2;
        const int cntA = r1 - gc, cntB = gc - r0;  // cntA >= cntB
        uint32_t *bitsA = bits_all + (2 * wid) * BITW, *bitsB = bitsA + BITW;
        for (int i = lane; i < BITW; i += 64) bitsA[i] = 0u;
        lds_wave_fence();
        const uint32_t *bits_h = half ? bitsB : bitsA;
        double gsumA = 0.0, gsumB = 0.0;
        R amax = (R)0;  // DQM: largest magnitude that went through the running sums of this wave (in the series' own type: exact)
        if (DQM && kind == '+' && wid == 0 && lane == 0 && nvalid > 1) {
            // additive kind: the running sums leave an absolute error of ~2e-16 * (largest |value|) on a group mean.
            // Harmless unless the series holds wild outliers (unmasked fill values, 1e300 ...): compare the extremes
            // with the interquartile range of the series, both read off the sorted order, once per array.
            R lo, hi, q1, q3;
            vals.at_ranks(0, nvalid - 1, lo, hi);
            vals.at_ranks(nvalid / 4, (3 * nvalid) / 4, q1, q3);
            constexpr double LIM_ADD = sizeof(R) == 8 ? 1.0e4 : 1.0e8;
            if (!(fmax(fabs((double)lo), fabs((double)hi)) <= LIM_ADD * ((double)q3 - (double)q1))) flag_cell = true;
        }
        // software prefetch of the next toggle lists (first 64 entries each): L2 latency hides under the select.
        // "list r" is the symmetric difference of rows r-1 and r: row r of the padded [G][64] table tog64 (when every
        // list has <= 64 entries, e.g. doy windows: one independent vector load, no pointer chase through SMEM), else
        // tog_idx[tog_ptr[r-1] .. tog_ptr[r])
        constexpr int TOGN = TOG64 ? togn_of<Geo>::value : 1;
        int preA[TOGN], preB[TOGN], preA_len = 0, preA_base = 0, preB_len = 0, preB_base = 0;
#pragma unroll
        for (int u = 0; u < TOGN; u++) preA[u] = preB[u] = -1;
        auto load_list = [&](int r, int (&pre)[TOGN], int &base, int &len) {
            if (TOG64) {
#pragma unroll
                for (int u = 0; u < TOGN; u++) pre[u] = tog64[(size_t)r * (64 * TOGN) + 64 * u + lane];
            } else { base = tog_ptr[r - 1]; len = tog_ptr[r] - base; pre[0] = lane < len ? tog_idx[base + lane] : -1; }
        };
        if (cntA > 1) load_list(gc + 1, preA, preA_base, preA_len);  // A, it = 1: gc -> gc+1
        if (cntB > 0) load_list(gc, preB, preB_base, preB_len);      // B, it = 0: gc -> gc-1
        // quantile-index math depends on (n, k) only: cached while both halves keep their n
        int nA_prev = -1, nB_prev = -1, c_case = 3, c_lo = 0;
        R c_gamma = (R)0;
        const int n_iter = cntA;
        for (int it = 0; it < n_iter; it++) {
            const int gA = gc + it, gB = gc - 1 - it;
            const bool vA = true, vB = it < cntB;
            stamp(-1);
            double dlocA = 0.0, dlocB = 0.0;
            // Toggles are issued in two waves of LDS traffic — all the rank reads, then all the XORs — because the
            // compiler must otherwise order every rank read behind the previous atomic (both live in the same LDS
            // array): one LDS round trip per batch instead of two per element.
            // Only DQM needs the old bit back.
            auto flip = [&](uint32_t *bits, int t, int r, double &dloc) {
                if (t >= 0 && r < nvalid) {
                    const uint32_t m = 1u << (r & 31);
                    if (DQM) {
                        uint32_t old = atomicXor(&bits[r >> 5], m);
                        R v = vals.at_time(t);
                        dloc += (old & m) ? -(double)v : (double)v;
                        amax = fmax(amax, (R)fabs(v));
                    } else {
                        (void)__hip_atomic_fetch_xor(&bits[r >> 5], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            };
            if (it == 0) {
                // the vector of the middle group from its row list, 8 rounds (512 entries) at a time, then copied to B
                const int *l = row_sorted + row_ptr[gc];
                const int len = row_ptr[gc + 1] - row_ptr[gc];
                for (int i0 = 0; i0 < len; i0 += 512) {
                    int ts[8], rs[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) { int i = i0 + 64 * u + lane; ts[u] = i < len ? l[i] : -1; }
#pragma unroll
                    for (int u = 0; u < 8; u++) rs[u] = ts[u] >= 0 ? (int)rank[ts[u]] : 0x7FFFFFFF;
#pragma unroll
                    for (int u = 0; u < 8; u++) flip(bitsA, ts[u], rs[u], dlocA);
                }
                lds_wave_fence();
#pragma unroll
                for (int j = 0; j < WPL; j++) bitsB[lane + 64 * j] = bitsA[lane + 64 * j];
                lds_wave_fence();
            }
            if (debug_phase != 4) {
                // consume the lists prefetched during the previous group FIRST, then issue the next prefetch: the
                // wait for tA/tB then never covers the loads just issued.  A: list gA (gA-1 -> gA), none for its
                // first group; B: list gB+1 (gB+1 -> gB)
                const bool doA = it > 0;
                int tA[TOGN], tB[TOGN], rA[TOGN], rB[TOGN];
#pragma unroll
                for (int u = 0; u < TOGN; u++) { tA[u] = doA ? preA[u] : -1; tB[u] = vB ? preB[u] : -1; }
                const int lenA = doA ? preA_len : 0, lenB = vB ? preB_len : 0, lbA = preA_base, lbB = preB_base;
#pragma unroll
                for (int u = 0; u < TOGN; u++) { rA[u] = tA[u] >= 0 ? (int)rank[tA[u]] : 0x7FFFFFFF; rB[u] = tB[u] >= 0 ? (int)rank[tB[u]] : 0x7FFFFFFF; }
#pragma unroll
                for (int u = 0; u < TOGN; u++) { flip(bitsA, tA[u], rA[u], dlocA); flip(bitsB, tB[u], rB[u], dlocB); }
                if (it + 1 < cntA) load_list(gA + 1, preA, preA_base, preA_len);
                if (it + 1 < cntB) load_list(gB, preB, preB_base, preB_len);
                if (!TOG64) {  // lists longer than 64 entries (coarse groups)
                    for (int i = lane + 64; i < lenA; i += 64) { int t = tog_idx[lbA + i]; flip(bitsA, t, (int)rank[t], dlocA); }
                    for (int i = lane + 64; i < lenB; i += 64) { int t = tog_idx[lbB + i]; flip(bitsB, t, (int)rank[t], dlocB); }
                }
            }
            lds_wave_fence();
            stamp(it == 0 ? 7 : 0);  // toggles (seg 0) / initial builds (seg 7)
            if (debug_phase == 6) { if (lane == 0 && bitsA[3] == 12345u) out_hq[0] = 1.f; break; }
            // per-lane member counts of both vectors, packed (A low 16 bits, B high 16 bits): ONE prefix scan
            int clp = 0;
#pragma unroll
            for (int j = 0; j < WPL; j++) clp += __popc(bitsA[lane * WPL + j]) + (__popc(bitsB[lane * WPL + j]) << 16);
            const int inclp = wave_incl_scan(clp);
            const int exclp = inclp - clp;
            const int totp = __builtin_amdgcn_readlane(inclp, 63);
            const int nA = totp & 0xFFFF, nB = (int)((unsigned)totp >> 16);
            const unsigned long long ballotA = __ballot((clp & 0xFFFF) != 0), ballotB = __ballot(((unsigned)clp >> 16) != 0u);
            stamp(1);  // popcount + scan
            const int n = half ? nB : nA;
            const int g = half ? gB : gA;
            const bool vh = half ? vB : vA;
            R mu = (R)0, inv = (R)0;
            if (DQM) {
                // at it == 0 dlocA holds the sum of the built vector, which B starts from, and dlocB B's first toggles
                const double sA = wave_total_dpp(dlocA), sB = wave_total_dpp(dlocB);
                if (it == 0) gsumB = sA;
                gsumA += sA;
                gsumB += sB;
                const double gs = half ? gsumB : gsumA;
                mu = n ? (R)(gs / (double)n) : nan_of<R>();
                inv = invert<R>(mu, kind);
                if (vh && n && (!isfinite(mu) || (kind == '*' && !(inv > (R)0 && isfinite(inv))))) flag_cell = true;
                {
                    // The running float64 sums carry a rounding error of about eps * |largest value seen| per toggle.
                    // Keep them only while that stays below ~1e-12 (float64 series) / ~5e-9 (float32 series) of the
                    // group sums; otherwise (huge dynamic range, e.g. 1e-20 drizzle next to 1e2 storms, or a group mean
                    // next to zero) the generic kernel redoes the cell with fresh sums.  Every lane tests the values IT
                    // pushed through the sums against both streams' sums; the cell flag ORs the lanes.  Only the
                    // multiplicative kind divides by the mean; for the additive kind the absolute error of the mean,
                    // ~1e-17 of the largest value, is far below the data's own resolution whatever the mean is.
                    constexpr double LIM = sizeof(R) == 8 ? 200.0 : 1.0e6;
                    const double gmin = fmin((vA && nA) ? fabs(gsumA) : __builtin_inf(), (vB && nB) ? fabs(gsumB) : __builtin_inf());
                    if (kind == '*' && !((double)amax <= LIM * gmin)) flag_cell = true;
                }
            }
            const size_t obase = ((size_t)c * G + (vh ? g : 0)) * (size_t)Q;
            // rank-space positions of the j-th member of this half's vector (lrank) and of the member after it (hrank;
            // found = it lies in the same lane's words)
            auto select_pair = [&](int j, int &L, int &lrank, int &hrank, bool &found) {
                // owner lane: the last lane whose exclusive count is <= j.  Branch-free bisection on the raw
                // ds_bpermute (byte-addressed lane index kept pre-shifted), this half's 16-bit field by one v_bfe
                int L4 = 0, eL = 0;
#pragma unroll
                for (int s = 32; s > 0; s >>= 1) {
                    const int cand4 = L4 + 4 * s;  // <= 4 * 63: the steps sum to 63
                    const int ep = __builtin_amdgcn_ds_bpermute(cand4, exclp);
                    const int e = (int)(((unsigned)ep >> hshift) & 0xFFFFu);
                    const bool ok = e <= j;
                    L4 = ok ? cand4 : L4;
                    eL = ok ? e : eL;
                }
                L = L4 >> 2;
                int rem = j - eL;
                const uint32_t *wp = bits_h + L * WPL;
                uint32_t wv[WPL];
#pragma unroll
                for (int i = 0; i < WPL; i++) wv[i] = wp[i];
                // the word holding the rem-th member: running popcounts, monotone selects
                bool ge[WPL];
                int run = 0, sub = 0, wi = 0;
                uint32_t w = wv[0];
#pragma unroll
                for (int i = 0; i < WPL - 1; i++) {
                    run += __popc(wv[i]);
                    ge[i] = rem >= run;
                    w = ge[i] ? wv[i + 1] : w;
                    sub = ge[i] ? run : sub;
                    wi = ge[i] ? i + 1 : wi;
                }
                rem -= sub;
                // first non-empty word after it (scanned backwards, the nearest one wins)
                int nw = 0;
                uint32_t nwv = 0u;
#pragma unroll
                for (int i = WPL - 1; i >= 1; i--) {
                    const bool c = wv[i] != 0u && !ge[i - 1];  // i > wi
                    nw = c ? i : nw;
                    nwv = c ? wv[i] : nwv;
                }
                const uint32_t wsel = w;
                int pos = 0;
#pragma unroll
                for (int s = 16; s > 0; s >>= 1) {
                    int cw = __popc(w & ((1u << s) - 1u));
                    bool g2 = rem >= cw;
                    rem -= g2 ? cw : 0;
                    w = g2 ? (w >> s) : w;
                    pos += g2 ? s : 0;
                }
                const int wbase = L * WPL * 32;
                lrank = wbase + wi * 32 + pos;
                const uint32_t rest = (wsel >> pos) >> 1;  // members above pos in the same word
                const bool inword = rest != 0u;
                found = inword || nw != 0;
                hrank = inword ? lrank + __ffs((int)rest) : wbase + nw * 32 + __ffs((int)nwv) - 1;
            };
            if (debug_phase == 3 || debug_phase == 4) { if (n == 12345 && lane == 0) out_hq[obase] = (R)exclp; continue; }
            for (int kb = 0; kb < Q; kb += 32) {
                const int k = kb + (lane & 31);
                const bool act = vh && k < Q;
                int casek, lo;  // casek 0: general lerp, 1: clamp to a[n-1], 2: clamp to a[0], 3: empty
                R gamma;
                if (kb == 0 && nA == nA_prev && nB == nB_prev) { casek = c_case; lo = c_lo; gamma = c_gamma; }
                else {
                    const double qk = (double)q[k < Q ? k : 0];
                    const double vi = (double)n * qk + (1.0 + qk * (1.0 - 1.0 - 1.0)) - 1.0;
                    lo = 0; gamma = (R)0;
                    if (n == 0) casek = 3;
                    else if (vi != vi || vi >= (double)(n - 1)) { casek = 1; lo = n - 1; }
                    else if (vi < 0.0) { casek = 2; lo = 0; }
                    else { casek = 0; lo = (int)floor(vi); gamma = (R)(vi - (double)lo); }
                    if (kb == 0) { c_case = casek; c_lo = lo; c_gamma = gamma; }
                }
                stamp(2);  // DQM mean, quantile index math
                int L, lrank, hrank;
                bool found;
                select_pair(lo, L, lrank, hrank, found);
                stamp(3);  // select(lo) + next member in the owner lane's words
                const bool need_slow = casek == 0 && !found;  // the next member lives in another lane's words
                if (__any(need_slow)) {
                    // (nearly every round has such a lane) it is the first member of the next lane that holds any:
                    // that lane from the ballot of non-empty lanes, its first set bit from its words — no second select
                    const unsigned long long restm = L < 63 ? ((half ? ballotB : ballotA) >> (L + 1)) : 0ull;
                    const int L2 = restm ? L + __ffsll(restm) : L;
                    const uint32_t *wp2 = bits_h + L2 * WPL;
                    uint32_t w2[WPL];
#pragma unroll
                    for (int i = 0; i < WPL; i++) w2[i] = wp2[i];
                    int hr2 = 0;
#pragma unroll
                    for (int i = WPL - 1; i >= 0; i--) hr2 = w2[i] != 0u ? i * 32 + __ffs((int)w2[i]) - 1 : hr2;
                    if (need_slow) hrank = L2 * WPL * 32 + hr2;
                }
                stamp(5);  // next set bit (+ second select)
                // both neighbours are looked up together: two independent (index, key) LDS round trips in flight
                const int rs0 = min(lrank, TPAD - 1), rs1 = min(casek == 0 ? hrank : lrank, TPAD - 1);
                R v, vhi;
                vals.at_ranks(rs0, rs1, v, vhi);
                stamp(4);  // value lookups
                if (DQM) { v = apply_corr<R>(v, inv, kind); vhi = apply_corr<R>(vhi, inv, kind); }
                R res;
                if (casek == 3) res = nan_of<R>();
                else if (casek != 0) res = v;
                else {
                    R left = v, right = vhi;
                    R diff = right - left;
                    if (gamma >= (R)0.5) res = (R)((double)right - (double)diff * (1.0 - (double)gamma));
                    else res = left + diff * gamma;
                    if (res != res && act) flag_cell = true;  // inf - inf: needs a[n-1]; the generic kernel redoes the cell
                }
                if (act) {
                    if (mode == 0) out_hq[obase + k] = res;
                    else if (arr == 0) out_hq[obase + k] = res;  // hist_q
                    else af[obase + k] = get_corr<R>(out_hq[obase + k], res, kind);  // res = ref_q
                }
            }
            stamp(6);  // upper value, interpolation, stores
            nA_prev = nA; nB_prev = nB;
            if (DQM && vh && (lane & 31) == 0) {
                if (arr == 0) scaling[(size_t)c * G + g] = mu;  // mean(hist), replaced below
                else scaling[(size_t)c * G + g] = get_corr<R>(scaling[(size_t)c * G + g], mu, kind);
            }
        }
}

// DQM: dqm_train (group-mean normalisation + scaling); TOG64: the padded [G][64] toggle table is used.  Both are
// template parameters (as are the timing aids under DIAG) so that the group loop carries no mode branches.
template <int EPT, bool DIAG = false, bool DQM = false, bool TOG64 = false>
__global__ __launch_bounds__(1024) void k_train_cell(
    const float *__restrict__ x0, const float *__restrict__ x1, int64_t T, int64_t sc, const int *__restrict__ row_ptr,
    const int *__restrict__ row_sorted, const int *__restrict__ tog_ptr, const int *__restrict__ tog_idx, int G,
    const float *__restrict__ q, int Q, int kind, int mode, float *__restrict__ af, float *out_hq, float *scaling,
    int *__restrict__ flag_count, int *__restrict__ flag_list, int debug_phase_arg, int64_t C,
    const int *__restrict__ tog64)
{
    const int debug_phase = DIAG ? debug_phase_arg : 0;  // timing aids exist in the diagnostic build only
    using Cfg = CellCfg<EPT>;
    constexpr int NT = Cfg::NT, TPAD = Cfg::TPAD;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *keys = reinterpret_cast<uint32_t *>(smem + Cfg::off_keys);
    uint16_t *rank = reinterpret_cast<uint16_t *>(smem + Cfg::off_rank);
    uint32_t *bits_all = reinterpret_cast<uint32_t *>(smem + Cfg::off_bits);
    uint32_t *red = reinterpret_cast<uint32_t *>(smem + Cfg::off_misc);

    const int tid = threadIdx.x, wid = tid >> 6;
    // persistent-style grid: each block walks its cells grid-stride
    // diagnostic build (XSDBA_HIP_DEBUG_PHASE=9): cycle shares of the phase-C segments, written to out_hq[0..7]
    unsigned long long tacc[DIAG ? 8 : 1] = {0}, tlast = 0;
    auto stamp = [&](int seg) { if constexpr (DIAG) { unsigned long long now = __builtin_readcyclecounter(); if (seg >= 0) tacc[seg] += now - tlast; tlast = now; } };
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
    const int npass_arr = mode == 0 ? 1 : 2;
    bool flag_cell = false;

    for (int arr = 0; arr < npass_arr; arr++) {
        // opaque per pass: lane-derived addresses are recomputed here, not kept live (spilled) across the sort
        int lane = tid & 63;
        __asm__ volatile("" : "+v"(lane));
        // hist goes first: its quantiles are an output anyway and are read back (L2 / Infinity Cache) by the ref
        // pass to form af, so nothing is parked in scratch and every output byte is written exactly once
        const float *x = ((mode == 0 || arr == 1) ? x0 : x1) + c * sc;
        // ---------------- A + B. load, keys, LSD radix sort of the time indexes, rank[] ----------------
        int nvalid;
        const uint16_t *sorted = cell_load_sort<EPT>(x, T, smem, red, debug_phase == 1 ? 1 : (debug_phase == 2 ? 2 : 0), nvalid);
        if (debug_phase == 1) continue;
        if (debug_phase == 2) { if (tid == 0) out_hq[(size_t)c * G * Q] = (float)sorted[nvalid / 2]; __syncthreads(); continue; }
        if (debug_phase == 5) { if (tid == 0) out_hq[(size_t)c * G * Q] = (float)rank[nvalid / 2]; __syncthreads(); continue; }

        // ---------------- C0. a single group holding every time step (group="time"): the sorted order IS the row
        if (G == 1 && !DQM && row_ptr[1] - row_ptr[0] == (int)T) {
            const int n = nvalid;
            for (int k = tid; k < Q; k += NT) {
                auto load = [&](int i) { return from_key(keys[sorted[i]]); };
                float res = n ? quantile_sorted<float>(load, n, q[k]) : __builtin_nanf("");
                const size_t o = (size_t)c * Q + k;
                if (mode == 0 || arr == 0) out_hq[o] = res;
                else af[o] = get_corr<float>(out_hq[o], res, kind);
            }
            __syncthreads();
            continue;
        }
        // ---------------- C. membership bit-vectors in rank space + select ----------------
        {
            const LdsVals vals{keys, sorted, TPAD};
            cell_group_quantiles<Cfg, DIAG, DQM, TOG64>(vals, rank, bits_all, nvalid, lane, wid, row_ptr, row_sorted, tog_ptr,
                                                         tog_idx, tog64, G, q, Q, kind, mode, arr, c, af, out_hq, scaling,
                                                         flag_cell, debug_phase, stamp);
        }
        __syncthreads();  // all waves are done with keys / sorted / rank before the next array reuses them
    }
    // cells whose shortcut assumptions failed are queued for the generic kernel
    int any = __syncthreads_or(flag_cell ? 1 : 0);
    if (any && tid == 0 && flag_list) {
        int slot = atomicAdd(flag_count, 1);
        flag_list[slot] = (int)c;
    }
    flag_cell = false;
    }  // cell loop
    if constexpr (DIAG) { if (blockIdx.x == 0 && tid == 0) for (int i = 0; i < 8; i++) out_hq[i] = (float)((double)tacc[i] * 1e-3); }
}

}  // namespace xs
