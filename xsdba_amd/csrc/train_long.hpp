// train_long.hpp — the per-cell training kernel for series of 11 265 .. 22 528 time steps (e.g. 60 years daily).
//
// One series no longer fits the LDS as keys + handles + sort scratch, so the keys never go there:
//   1. the series goes from HBM into REGISTERS (2 EPT keys per thread) and is sorted as ONE sequence by the handle sort of
//      train_cell.hpp — pass 1: the 14 low key bits by a counting sort with LDS atomics (histogram fused into the load), the
//      handles ((key bits above 14) | time step) scattered into the LDS; then stable counting passes of up to 8 bits on
//      wave-private counter rows, the handles in registers across each pass (in place).  The last pass leaves rank[t] in LDS
//      (the walk's toggles need it) and the time steps in rank order in a per-block global scratch slice (45 KB: 11.5 MB for
//      the 256 resident blocks, L2-resident);
//   2. phase C is the one of k_train_cell (cell_group_quantiles) with 704-word vectors; an order statistic is fetched as
//      x[sorted[r]]: the time step from the scratch slice (L2), the value from the series itself (read-only: L1 / L2 hits).
// Rounds 1-3 sorted the two halves of the series with the short kernel's sort, parked their (key, time) sequences in 225 KB
// of global scratch per block, brought them back and merged them by bisection: 11.8 of 23.0 ms per 32 400 cells x 21 900
// days and 8.4 x the algorithmic bytes in FETCH + WRITE (the parked halves and the rank-ordered keys overflowed the L2).
// Same results as k_train_cell / k_train_rows: the rank order is a total order of the same keys, ties in any order (which
// never changes a quantile value).
#pragma once
#include "train_cell.hpp"

namespace xs {

// EPT: pairs of elements per thread — 11 (series up to 22 528 steps) or 8 (up to 16 384: 31 to 44 years of daily steps pay for
// 16 384 places instead of 22 528)
template <int EPT> struct LongCfg {
    static constexpr int NT = 1024, TPAD = 2 * EPT * NT;
    static constexpr int NWORDS = TPAD / 32, WPL = (NWORDS + 63) / 64, BITW = WPL * 64, NWAVE = NT / 64;
    static constexpr int TOGN = 2;  // the padded toggle table of this geometry: [G][128] (60 years of day-of-year windows: 120 steps)
    static constexpr int WROW = 256 + 8;  // dwords per wave-private counter row (see CellCfg)
    static_assert(TPAD <= (1 << 15), "a handle keeps the time step in 15 bits");
    // sort view of the LDS: [handles | histogram of pass 1 ... | counter rows | bit 31 of the reduced keys]; phase-C view:
    // [2 * NWAVE vectors | rank]: the histogram lies where rank[] will (dead before the last pass writes rank[])
    static constexpr size_t off_h = 0, off_rank = (size_t)TPAD * 4, off_bits = 0;
    static constexpr size_t off_hist = off_rank;
    static constexpr size_t off_wcnt = off_rank + ((size_t)TPAD * 2 > (size_t)(1 << 14) * 2 ? (size_t)TPAD * 2 : (size_t)(1 << 14) * 2);
    static constexpr size_t off_b31 = off_wcnt + (size_t)NWAVE * WROW * 4;
    static constexpr size_t off_misc = (off_b31 + (size_t)TPAD / 8 + 15) & ~(size_t)15;
    static_assert((size_t)2 * NWAVE * BITW * 4 <= off_rank, "the vectors must end before rank[]");
    static constexpr size_t lds_bytes = off_misc + 64 * 8;
    // per-block global scratch: the time steps in rank order
    static constexpr size_t scratch_bytes = (size_t)TPAD * 2;
};

// loads that must see what other threads of this workgroup stored to global memory: bypass the per-CU L1
__device__ __forceinline__ uint32_t load_l2(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint16_t load_l2(const uint16_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ... after the storing waves have drained their stores (write-through to L2) and the workgroup has met at a barrier.
// An agent-scope fence would also write back / invalidate caches for other workgroups, which nobody needs here.
__device__ __forceinline__ void stores_done()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

struct GlobalValsIdx {
    using value_type = float;
    const float *x;              // the series (read-only for the whole launch: plain cached loads)
    const uint16_t *sorted;      // time steps in rank order, written by this block's sort: read past the L1
    int tlast;                   // the last time step of the series (ranks past the valid ones hold padding steps beyond it)
    __device__ __forceinline__ float at_time(int t) const { return x[t]; }
    __device__ __forceinline__ int idx_load(int r) const { return (int)load_l2(sorted + r); }
    __device__ __forceinline__ float val_load(int t) const { return x[min(t, tlast)]; }  // (lanes without a quantile look up anything)
    __device__ __forceinline__ void at_ranks(int r0, int r1, float &v0, float &v1) const
    {
        const int t0 = idx_load(r0), t1 = idx_load(r1);
        v0 = val_load(t0);
        v1 = val_load(t1);
    }
};

// Phase C behind a call: the walk's register allocation is then its own (inlined, its loops shared the 128 registers of a
// 1024-thread block with whatever the sorts and the merge keep alive across the cell loop, and spilled in its group loop).
template <bool DQM, bool TOG64, int EPT>
__device__ __attribute__((noinline)) void long_phase_c(const float *x, int n_time, const uint16_t *sorted_g, const uint16_t *rank, uint32_t *bits_all, int nvalid,
                                                       int lane, int wid, const int *row_ptr, const int *row_sorted, const int *tog_ptr,
                                                       const int *tog_idx, const int *tog64, int G, const float *q, int Q, int kind, int mode,
                                                       int arr, int64_t c, float *af, float *out_hq, float *scaling, bool &flag_cell)
{
    auto nostamp = [](int) {};
    const GlobalValsIdx vals{x, sorted_g, n_time - 1};
    cell_group_quantiles<LongCfg<EPT>, false, DQM, TOG64>(vals, rank, bits_all, nvalid, lane, wid, row_ptr, row_sorted, tog_ptr, tog_idx, tog64, G, q, Q,
                                                     kind, mode, arr, c, af, out_hq, scaling, flag_cell, 0, nostamp);
}

// The sort of one series of up to TPAD steps: keys in registers, handles in LDS (see the header).  Returns the number of valid
// values (0: nothing was sorted); leaves rank[t] in LDS and the time steps in rank order in `sorted_g`.  All 1024 threads; the
// caller must drain the stores and meet at a barrier before anyone reads either.
// (COPY_OUT = false: the order stays in LDS only — train_xlong.hpp reads it there)
template <int EPT, bool COPY_OUT = true>
__device__ __forceinline__ int long_load_sort(const float *__restrict__ x, int Ti, unsigned char *smem, uint32_t *red, uint16_t *rank,
                                              uint16_t *sorted_g)
{
    using L = LongCfg<EPT>;
    constexpr int NT = L::NT, TPAD = L::TPAD, NE = 2 * EPT, D1 = 14, NB1 = 1 << D1, WROW = L::WROW, RND = TPAD / NT;  // RND = NE
    uint32_t *H = reinterpret_cast<uint32_t *>(smem + L::off_h);
    uint16_t *sorted_l = reinterpret_cast<uint16_t *>(smem + L::off_h);  // the final order, staged where the handles were
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem + L::off_hist);
    uint32_t *wcnt = reinterpret_cast<uint32_t *>(smem + L::off_wcnt);
    uint32_t *b31 = reinterpret_cast<uint32_t *>(smem + L::off_b31);
    int tid = threadIdx.x;
    __asm__ volatile("" : "+v"(tid));  // (see cell_load_sort)
    const int lane = tid & 63, wid = tid >> 6;
    uint32_t *wrow = wcnt + wid * WROW;
    // ---- A. the series -> keys in registers (element e of a thread: time step 2 tid + (e / 2) 2 NT + (e & 1)), range, histogram
    float v[NE];
    const int Te = Ti & ~1;
    if ((reinterpret_cast<uintptr_t>(x) & 7) == 0) {
        const float xlast = x[Ti - 1];
#pragma unroll
        for (int j = 0; j < EPT; j++) {
            const int i = 2 * tid + j * 2 * NT;
            const float2 p = *reinterpret_cast<const float2 *>(x + min(i, Te - 2));
            v[2 * j] = p.x; v[2 * j + 1] = p.y;
        }
#pragma unroll
        for (int j = 0; j < EPT; j++) {
            const int i = 2 * tid + j * 2 * NT;
            if ((j + 1) * 2 * NT > Te) {  // (wave-uniform: only the last rounds can reach past the series)
                if (i >= Te) { v[2 * j] = (i == Ti - 1) ? xlast : __builtin_nanf(""); v[2 * j + 1] = __builtin_nanf(""); }
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < NE; e++) {
            const int i = 2 * tid + (e >> 1) * 2 * NT + (e & 1);
            v[e] = x[min(i, Ti - 1)];
        }
#pragma unroll
        for (int e = 0; e < NE; e++) {
            const int i = 2 * tid + (e >> 1) * 2 * NT + (e & 1);
            if (i >= Ti) v[e] = __builtin_nanf("");
        }
    }
    {   // the tables are cleared while the loads are in flight
        uint4 *hz = reinterpret_cast<uint4 *>(hist);
        for (int i = tid; i < NB1 / 8; i += NT) hz[i] = make_uint4(0u, 0u, 0u, 0u);
        for (int i = lane; i < WROW; i += 64) wrow[i] = 0u;
        for (int i = tid; i < TPAD / 32; i += NT) b31[i] = 0u;
    }
    __syncthreads();
    uint32_t k[NE];
    int nv = 0;
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
    for (int e = 0; e < NE; e++) {  // (integer arithmetic: see cell_load_sort)
        const uint32_t u = __float_as_uint(v[e]);
        const uint32_t ok = (uint32_t)((int32_t)((u & 0x7FFFFFFFu) - 0x7F800001u) >> 31);
        const uint32_t vk = (u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u)) & ok;
        nv -= (int)ok;
        kmax = max(kmax, vk);
        k[e] = vk | ~ok;
        kmin = min(kmin, k[e]);
        // pass 1 counts the 14 low bits of the KEYS: the digits of (key - kmin) for a kmin that is a multiple of 2^14 (below)
        const uint32_t d = k[e] & (NB1 - 1);
        if (k[e] != 0xFFFFFFFFu) atomicAdd(&hist[d >> 1], 1u << (16 * (d & 1u)));
    }
#define XS_RED_STEP(ctrl, rmask)                                                                          \
    nv += __builtin_amdgcn_update_dpp(0, nv, ctrl, rmask, 0xf, false);                                    \
    kmin = min(kmin, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)kmin, ctrl, rmask, 0xf, false));      \
    kmax = max(kmax, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)kmax, ctrl, rmask, 0xf, false));
    XS_RED_STEP(0x111, 0xf) XS_RED_STEP(0x112, 0xf) XS_RED_STEP(0x114, 0xf) XS_RED_STEP(0x118, 0xf)
    XS_RED_STEP(0x142, 0xa) XS_RED_STEP(0x143, 0xc)
#undef XS_RED_STEP
    if (lane == 63) { red[3 * wid] = (uint32_t)nv; red[3 * wid + 1] = kmin; red[3 * wid + 2] = kmax; }
    __syncthreads();
    {
        static_assert(L::NWAVE == 16, "one DPP row holds the per-wave results");
        nv = (int)red[3 * (lane & 15)]; kmin = red[3 * (lane & 15) + 1]; kmax = red[3 * (lane & 15) + 2];
#define XS_ROW_ROR(x, n) __builtin_amdgcn_update_dpp((int)(x), (int)(x), 0x120 + (n), 0xf, 0xf, false)
        nv += XS_ROW_ROR(nv, 8); kmin = min(kmin, (uint32_t)XS_ROW_ROR(kmin, 8)); kmax = max(kmax, (uint32_t)XS_ROW_ROR(kmax, 8));
        nv += XS_ROW_ROR(nv, 4); kmin = min(kmin, (uint32_t)XS_ROW_ROR(kmin, 4)); kmax = max(kmax, (uint32_t)XS_ROW_ROR(kmax, 4));
        nv += XS_ROW_ROR(nv, 2); kmin = min(kmin, (uint32_t)XS_ROW_ROR(kmin, 2)); kmax = max(kmax, (uint32_t)XS_ROW_ROR(kmax, 2));
        nv += XS_ROW_ROR(nv, 1); kmin = min(kmin, (uint32_t)XS_ROW_ROR(kmin, 1)); kmax = max(kmax, (uint32_t)XS_ROW_ROR(kmax, 1));
#undef XS_ROW_ROR
        nv = __builtin_amdgcn_readfirstlane(nv);
        kmin = (uint32_t)__builtin_amdgcn_readfirstlane((int)kmin);
        kmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)kmax);
    }
    const int nvalid = nv;
    if (nvalid == 0) { __syncthreads(); return 0; }  // (block-uniform; the barrier protects red[])
    kmin &= ~(uint32_t)(NB1 - 1);
    const uint32_t range = kmax - kmin;      // NaN / padding are remapped to range + 1
    const int nbits = 32 - __clz((int)(range + 1u) | 0);
    const int R = nbits > D1 ? nbits - D1 : 0;           // key bits left after pass 1 (<= 18)
    const int npass_rest = (R + 7) >> 3;
    const int db = npass_rest ? (R + npass_rest - 1) / npass_rest : 0;
    const int ND = 1 << db;
    // a handle holds bits 14 .. 30 of the reduced key above the 15-bit time step; bit 31 (a series that spans zero: the keys of
    // negative and positive floats lie 2^31 apart) lives in a bit per time step and joins the digit of the last pass
    const bool wide = nbits == 32;
    const uint32_t d_nan = (range + 1u) & (NB1 - 1);
    // ---- pass 1: scan of the histogram (16 bins per thread), slots by returning atomics
    {
        constexpr int BPT = NB1 / NT;
        uint32_t wv[BPT / 2];
        int run = 0;
#pragma unroll
        for (int i = 0; i < BPT / 2; i++) {
            wv[i] = hist[tid * (BPT / 2) + i];
            if ((int)(d_nan >> 4) == tid && (int)((d_nan & 15u) >> 1) == i) wv[i] += (uint32_t)(TPAD - nvalid) << (16 * (d_nan & 1u));
            run += (int)(wv[i] & 0xFFFFu) + (int)(wv[i] >> 16);
        }
        const int incl = wave_incl_scan(run);
        // (words 64.. of the scratch: a slow wave may still be reading the words of the all-reduce above)
        if (lane == 63) red[64 + wid] = (uint32_t)incl;
        __syncthreads();
        int pre = waves_before(red + 64, lane, wid) + incl - run;
#pragma unroll
        for (int i = 0; i < BPT / 2; i++) {
            const int c0 = (int)(wv[i] & 0xFFFFu), c1 = (int)(wv[i] >> 16);
            hist[tid * (BPT / 2) + i] = (uint32_t)pre | ((uint32_t)(pre + c0) << 16);
            pre += c0 + c1;
        }
    }
    __syncthreads();
    {
        const bool last = npass_rest == 0;
#pragma unroll
        for (int e = 0; e < NE; e++) {
            const int i = 2 * tid + (e >> 1) * 2 * NT + (e & 1);
            const uint32_t r = (k[e] == 0xFFFFFFFFu) ? range + 1u : k[e] - kmin;
            const uint32_t d = r & (NB1 - 1);
            const int sh = 16 * (d & 1);
            const uint32_t old = atomicAdd(&hist[d >> 1], 1u << sh);
            const int pos = (int)((old >> sh) & 0xFFFFu);
            if (!last) {
                H[pos] = (((r >> D1) & 0x1FFFFu) << 15) | (uint32_t)i;
                if (wide && (r >> 31)) atomicOr(&b31[i >> 5], 1u << (i & 31));
            } else sorted_l[pos] = (uint16_t)i;  // narrow keys: the order is final (rank[] below: it overlays the histogram)
        }
    }
    __syncthreads();
    // ---- the stable passes (train_cell.hpp: radix_sort_indexes), RND rounds of 64 consecutive places per wave
    for (int p = 0; p < npass_rest; p++) {
        const int shift = db * p + 15;  // (bit b >= 14 of the reduced key is bit b + 1 of the handle)
        const bool last = p == npass_rest - 1;
        const bool with31 = last && wide;   // the top digit is short of its highest bit: it comes from b31[]
        const uint32_t dmask = with31 ? (uint32_t)(ND >> 1) - 1u : (uint32_t)(ND - 1);
        const int topsh = db - 1;
        uint32_t hd[RND], od[RND];
        if (p > 0)
            for (int i = lane; i < ND; i += 64) wrow[i] = 0u;
        {
            const uint32_t *src = H + wid * (64 * RND) + lane;
#pragma unroll
            for (int j = 0; j < RND; j++) hd[j] = src[64 * j];
        }
        auto digit = [&](uint32_t h) {
            uint32_t d = (h >> shift) & dmask;
            if (with31) { const uint32_t t = h & 0x7FFFu; d |= ((b31[t >> 5] >> (t & 31)) & 1u) << topsh; }
            return d;
        };
#pragma unroll
        for (int j = 0; j < RND; j++) od[j] = atomicAdd(&wrow[digit(hd[j])], 1u);
        __syncthreads();
        {
            const int NDW = ND << 4, per = NDW > NT ? NDW >> 10 : 1;
            uint32_t c[4];
            int run = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int i = tid * per + q;
                c[q] = (q < per && i < NDW) ? wcnt[(i & 15) * WROW + (i >> 4)] : 0u;
                run += (int)c[q];
            }
            const int incl = wave_incl_scan(run);
            if (lane == 63) red[wid] = (uint32_t)incl;
            __syncthreads();
            int pre = waves_before(red, lane, wid) + incl - run;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int i = tid * per + q;
                if (q < per && i < NDW) wcnt[(i & 15) * WROW + (i >> 4)] = (uint32_t)pre;
                pre += (int)c[q];
            }
        }
        __syncthreads();
        {
            uint32_t bs[RND];
#pragma unroll
            for (int j = 0; j < RND; j++) bs[j] = wrow[digit(hd[j])];
            if (!last) {
#pragma unroll
                for (int j = 0; j < RND; j++) H[bs[j] + od[j]] = hd[j];
            } else {
#pragma unroll
                for (int j = 0; j < RND; j++) {
                    const int pos = (int)(bs[j] + od[j]);
                    const uint32_t t = hd[j] & 0x7FFFu;
                    sorted_l[pos] = (uint16_t)t;  // (where the handles were: every handle sits in a register)
                    rank[t] = (uint16_t)pos;      // (NaN / padding steps: pos >= nvalid, never a member)
                }
            }
        }
        __syncthreads();
    }
    if (npass_rest == 0) {
        for (int r = tid; r < TPAD; r += NT) rank[sorted_l[r]] = (uint16_t)r;
    }
    // the time steps in rank order leave the LDS (the walk's vectors take their place) in 16-byte pieces: scattered straight
    // to global memory they were 22 stores of 64 two-byte pieces per thread — a third of the sort's time
    if constexpr (COPY_OUT) {
        const uint4 *sl = reinterpret_cast<const uint4 *>(sorted_l);
        uint4 *sg = reinterpret_cast<uint4 *>(sorted_g);
        for (int i = tid; i < TPAD / 8; i += NT) sg[i] = sl[i];
    }
    return nvalid;
}

template <bool DQM, bool TOG64, int EPT = 11>
__global__ __launch_bounds__(1024) void k_train_long(
    const float *__restrict__ x0, const float *__restrict__ x1, int64_t T, int64_t sc, const int *__restrict__ row_ptr,
    const int *__restrict__ row_sorted, const int *__restrict__ tog_ptr, const int *__restrict__ tog_idx, int G,
    const float *__restrict__ q, int Q, int kind, int mode, float *__restrict__ af, float *out_hq, float *scaling,
    int *__restrict__ flag_count, int *__restrict__ flag_list, int64_t C, const int *__restrict__ tog64,
    unsigned char *__restrict__ gscratch, int debug_phase)
{
    using L = LongCfg<EPT>;
    constexpr int NT = L::NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t *rank = reinterpret_cast<uint16_t *>(smem + L::off_rank);
    uint32_t *bits_all = reinterpret_cast<uint32_t *>(smem + L::off_bits);
    uint32_t *red = reinterpret_cast<uint32_t *>(smem + L::off_misc);
    uint16_t *sorted_g = reinterpret_cast<uint16_t *>(gscratch + (size_t)blockIdx.x * L::scratch_bytes);

    const int tid = threadIdx.x, wid = tid >> 6;
    const int Ti = (int)T;
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        const int npass_arr = mode == 0 ? 1 : 2;
        bool flag_cell = false;
        for (int arr = 0; arr < npass_arr; arr++) {
            int lane = tid & 63;
            __asm__ volatile("" : "+v"(lane));
            const float *x = ((mode == 0 || arr == 1) ? x0 : x1) + c * sc;
            // ---------------- 1. sort ----------------
            const int nvalid = long_load_sort<EPT>(x, Ti, smem, red, rank, sorted_g);
            if (debug_phase == 1) { __syncthreads(); continue; }  // timing aid: the sort only
            if (nvalid == 0) {
                // no valid value in the series (block-uniform): every quantile of the pass is NaN — the expressions of the walk
                const float nanq = nan_of<float>();
                const size_t base = (size_t)c * G * (size_t)Q;
                for (int i = tid; i < G * Q; i += NT) {
                    if (mode == 0 || arr == 0) out_hq[base + i] = nanq;
                    else af[base + i] = get_corr<float>(out_hq[base + i], nanq, kind);
                }
                if constexpr (DQM)
                    for (int g = tid; g < G; g += NT) {
                        float &s_ = scaling[(size_t)c * G + g];
                        s_ = arr == 0 ? nanq : get_corr<float>(s_, nanq, kind);
                    }
                __syncthreads();
                continue;
            }
            stores_done();
            __syncthreads();  // rank[] and the rank-ordered time steps are complete; the handles are dead: the vectors take their place
            // ---------------- 2. group quantiles ----------------
            long_phase_c<DQM, TOG64, EPT>(x, Ti, sorted_g, rank, bits_all, nvalid, lane, wid, row_ptr, row_sorted, tog_ptr, tog_idx, tog64, G, q, Q, kind, mode,
                                     arr, c, af, out_hq, scaling, flag_cell);
            __syncthreads();  // everyone is done with rank[] / the vectors before the next sort reuses the LDS
        }
        const int any = __syncthreads_or(flag_cell ? 1 : 0);
        if (any && tid == 0 && flag_list) {
            const int slot = atomicAdd(flag_count, 1);
            flag_list[slot] = (int)c;
        }
    }
}

}  // namespace xs
