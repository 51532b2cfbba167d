// train_xlong.hpp — the per-cell training kernel for series of 22 529 .. 65 535 time steps (hourly data over a few years,
// more than 61 years of daily steps): nbutils.py:113-148 has no length limit, k_train_long ends where one series' handles
// fill the LDS.
//
//   1. the series is cut into K = ceil(T / 22 528) time chunks of equal length; each chunk is sorted by k_train_long's sort
//      (keys in registers, handles in LDS) and leaves the LDS as a run of 64-bit (key, position of the time step) pairs in the
//      block's global scratch;
//   2. the runs are merged pairwise (K - 1 merges) by output tiles of 16 384 pairs: the tile's slices of both runs come from a
//      merge-path search on its diagonals (a wave per diagonal), go into the LDS with coalesced loads (32-bit keys + 16-bit
//      steps), and every thread merges sixteen consecutive places (its own diagonal by bisection in LDS, then a pair per step);
//      the last merge writes rank[t] and the VALUES in rank order instead of a run;
//   3. rank[] and the rank-ordered steps live in global scratch (L2): the LDS belongs to the walk's bit vectors, which are
//      5.6 KB (45 056 steps) or 8 KB (65 536) each;
//   4. phase C is cell_group_quantiles (train_cell.hpp) on 14 (or 9) waves — the vectors of more do not fit —, rank[] read
//      through the L2 (a few hundred toggles per group step), an order statistic one load from the values in rank order (the
//      last merge writes them: a key gives its float back).
// Same results as k_train_rows on the same series: the rank order is a total order of the same keys.
#pragma once
#include "train_long.hpp"

namespace xs {

// geometry of the walk: WPL words per lane and vector — 22 (series up to 45 056 steps) or 32 (65 536)
template <int WPL_> struct XLongGeo {
    using L = LongCfg<11>;
    static constexpr int NT = 1024, WPL = WPL_, BITW = WPL * 64, TPAD = BITW * 32, NWORDS = BITW;
    static constexpr int TOGN = 4;  // the padded toggle table: [G][256] (hourly steps over five years toggle 240 per day-of-year group)
    static constexpr int TILE = 16384;  // pairs per merge tile (keys 64 KB + time steps 32 KB of LDS)
    static constexpr size_t sort_end = L::off_misc;  // the chunk sort's view of the LDS
    static constexpr int NWAVE_FIT = (int)((160 * 1024 - 1024) / ((size_t)2 * BITW * 4));
    static constexpr int NWAVE = NWAVE_FIT < 16 ? NWAVE_FIT : 16;  // waves that walk
    static constexpr size_t walk_end = (size_t)2 * NWAVE * BITW * 4;
    static constexpr size_t off_misc = ((sort_end > walk_end ? sort_end : walk_end) + 15) & ~(size_t)15;
    static constexpr size_t off_split = off_misc + 128 * 4;  // merge-path splits of up to 2 TPAD / TILE + 1 tiles
    static constexpr size_t lds_bytes = off_split + 32 * 4;
    static_assert((size_t)TILE * 6 <= off_misc, "a merge tile fits below the scratch words");
    static_assert(TPAD <= (1 << 16), "ranks and time steps are 16-bit");
    // per-block global scratch: two merged runs (ping-pong), the chunk's run, rank[], the values in rank order
    static constexpr size_t g_run = 0, g_run2 = (size_t)TPAD * 8, g_chunk = g_run2 + (size_t)TPAD * 8;
    static constexpr size_t g_rank = g_chunk + (size_t)L::TPAD * 8, g_sorted = g_rank + (size_t)TPAD * 2;
    static constexpr size_t scratch_bytes = g_sorted + (size_t)TPAD * 4;  // (the VALUES in rank order: float32)
};

__device__ __forceinline__ unsigned long long load_l2(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the order-preserving key of a float and back (train_cell.hpp's transform: positive floats get the sign bit, negative ones are
// complemented)
__device__ __forceinline__ float key_to_float(uint32_t key)
{
    return __uint_as_float((key & 0x80000000u) ? key ^ 0x80000000u : ~key);
}

// the values in rank order, written by this block: an order statistic is ONE load past the L1 (k_train_long's x[sorted[r]] is two
// dependent ones, the second a random 4-byte read of the series)
struct GlobalSortedVals {
    using value_type = float;
    const float *x;   // the series (DQM: the value of a toggled step)
    const float *sv;  // [TPAD] values in rank order (ranks past the valid ones: whatever the scratch holds)
    const int *time_of;  // the lists hold positions (see RankGlobal): the time step of a position
    __device__ __forceinline__ float at_time(int pos) const { return x[time_of[pos]]; }
    __device__ __forceinline__ void at_ranks(int r0, int r1, float &v0, float &v1) const
    {
        v0 = __hip_atomic_load(sv + r0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v1 = __hip_atomic_load(sv + r1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};

// rank[] in global memory, written by this block: read past the L1
// (indexed by POSITION — the plan's (base group, occurrence) order of the time steps, api.hip: d_xl_pos — like the row and toggle
// lists this kernel is given: the steps a group transition toggles are whole base groups, contiguous there, so a list's gathers
// touch a handful of sectors instead of one each; with rank[] in time order the kernel moved 24 x its algorithmic bytes)
struct RankGlobal {
    const uint16_t *p;
    __device__ __forceinline__ int operator[](int pos) const { return (int)load_l2(p + pos); }
};

// the sorted chunk leaves the LDS as a run: pair r = (key of x[t] << 32) | position of t, for the time step t of rank r (positions:
// see RankGlobal — the merged run then says where rank[] is to be written without another lookup)
template <int NT>
__device__ __attribute__((noinline)) void write_run(const float *x, int base, int len, const uint16_t *sorted_l, int nv, unsigned long long *dst,
                                                    const int *__restrict__ pos_of, uint16_t *pos_l, int tid)
{
    using u64 = unsigned long long;
    // the chunk's positions first go into LDS with coalesced loads (pos_l: 2 len bytes beside the order): gathered from the table
    // in global memory they cost as much as the sort's last pass
    for (int i = tid; i < len; i += NT) pos_l[i] = (uint16_t)pos_of[base + i];
    __syncthreads();
    for (int r0 = tid; r0 < nv; r0 += 8 * NT) {  // (eight gathers in flight)
        int t[8];
        uint32_t u[8];
#pragma unroll
        for (int e = 0; e < 8; e++) t[e] = (int)sorted_l[min(r0 + e * NT, nv - 1)];
#pragma unroll
        for (int e = 0; e < 8; e++) u[e] = __float_as_uint(x[base + t[e]]);
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const uint32_t key = u[e] ^ ((uint32_t)((int32_t)u[e] >> 31) | 0x80000000u);
            if (r0 + e * NT < nv) dst[r0 + e * NT] = ((u64)key << 32) | (u64)pos_l[t[e]];
        }
    }
}

// (behind a call, like the walk: inlined, its sixteen loads in flight shared the registers with whatever the kernel keeps
// alive across the cell loop, were spilled, and every reload waited for the loads before it)
template <int NT, int TILE, bool FINAL>
__device__ __attribute__((noinline)) void merge_runs(const unsigned long long *A, int n, const unsigned long long *B, int m, unsigned long long *O,
                                           uint16_t *rank_g, float *sorted_v, uint32_t *tk, uint16_t *tt, int *split, int tid, int dbg = 0)
{
    using u64 = unsigned long long;
    const int tot = n + m, ntile = (tot + TILE - 1) / TILE;
    const int lane = tid & 63, wid = tid >> 6;
    if (wid <= ntile) {
        // the merge path crosses diagonal d at (i, d - i), i the first index with key(A[i]) > key(B[d - 1 - i]): a wave per diagonal,
        // 64 candidates per round (three or four rounds of two independent loads instead of sixteen dependent ones)
        const int d = min(wid * TILE, tot);
        int lo = max(0, d - m), hi = min(d, n);
        while (hi > lo) {
            const int step = (hi - lo + 63) >> 6, mid = lo + lane * step;
            const bool below = mid < hi && (uint32_t)(load_l2(A + min(mid, n - 1)) >> 32) <= (uint32_t)(load_l2(B + max(d - 1 - mid, 0)) >> 32);
            const int cnt = __popcll(__ballot(below));  // (the predicate is monotone: true on the first cnt candidates)
            if (cnt == 0) hi = lo;
            else { hi = min(hi, lo + cnt * step); lo = lo + (cnt - 1) * step + 1; }
        }
        if (lane == 0) split[wid] = lo;
    }
    __syncthreads();
    if (dbg == 5) return;
    for (int k = 0; k < ntile; k++) {
        const int d0 = k * TILE, d1 = min(d0 + TILE, tot);
        const int i0 = split[k], i1 = split[k + 1], j0 = d0 - i0, na = i1 - i0, nt = d1 - d0, nb = nt - na;
        constexpr int PER = TILE / NT;
        {   // the tile's slices of both runs: [A's | B's], coalesced
            u64 v[PER];
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const int e = min(tid + u * NT, nt - 1);
                const u64 *src = e < na ? A + (i0 + e) : B + (j0 + e - na);  // (one load: a branch per slice would serialise them)
                v[u] = load_l2(src);
            }
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const int e = tid + u * NT;
                if (e < nt) { tk[e] = (uint32_t)(v[u] >> 32); tt[e] = (uint16_t)v[u]; }
            }
        }
        __syncthreads();
        if (dbg == 6) continue;
        // a thread merges PER consecutive places of the tile: its own diagonal by bisection, then one pair per step
        const int dd = min(tid * PER, nt);
        int lo = max(0, dd - nb), hi = min(dd, na);
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (tk[mid] <= tk[na + dd - 1 - mid]) lo = mid + 1;
            else hi = mid;
        }
        int i = lo, j = dd - lo;
        uint32_t ka = tk[min(i, nt - 1)], kb = tk[min(na + j, nt - 1)];
        float vout[PER];
        uint32_t tout[PER];
#pragma unroll
        for (int s2 = 0; s2 < PER; s2++) {
            const bool take_a = j >= nb || (i < na && ka <= kb);
            const int e = take_a ? i : na + j;
            const uint32_t t = tt[min(e, nt - 1)], key = take_a ? ka : kb;
            i += take_a ? 1 : 0;
            j += take_a ? 0 : 1;
            const uint32_t nk = tk[min(take_a ? i : na + j, nt - 1)];
            ka = take_a ? nk : ka;
            kb = take_a ? kb : nk;
            vout[s2] = key_to_float(key);
            tout[s2] = t;
            if (!FINAL && dd + s2 < nt) O[d0 + dd + s2] = ((u64)key << 32) | (u64)t;
        }
        if (FINAL) {
            if (dbg != 4) {
#pragma unroll
                for (int s2 = 0; s2 < PER; s2++)
                    if (dd + s2 < nt) rank_g[tout[s2]] = (uint16_t)(d0 + dd + s2);  // (the pairs carry positions)
            }
            if (dd + PER <= nt) {
                static_assert(PER == 16, "four 16-byte stores per thread");
                float4 *dst = reinterpret_cast<float4 *>(sorted_v + d0 + dd);
#pragma unroll
                for (int s4 = 0; s4 < 4; s4++) dst[s4] = make_float4(vout[4 * s4], vout[4 * s4 + 1], vout[4 * s4 + 2], vout[4 * s4 + 3]);
            } else {
#pragma unroll
                for (int s2 = 0; s2 < PER; s2++)
                    if (dd + s2 < nt) sorted_v[d0 + dd + s2] = vout[s2];
            }
        }
        __syncthreads();  // the tile is reused
    }
    stores_done();
    __syncthreads();
}

template <bool DQM, bool TOGP, int WPL>
__device__ __attribute__((noinline)) void xlong_phase_c(const float *x, int n_time, const float *sorted_v, const uint16_t *rank_g, uint32_t *bits_all,
                                                        int nvalid, int lane, int wid, const int *row_ptr, const int *row_sorted, const int *tog_ptr,
                                                        const int *tog_idx, const int *togp, const int *time_of, int G, const float *q, int Q, int kind,
                                                        int mode, int arr, int64_t c, float *af, float *out_hq, float *scaling, bool &flag_cell)
{
    auto nostamp = [](int) {};
    const GlobalSortedVals vals{x, sorted_v, time_of};
    const RankGlobal rk{rank_g};
    cell_group_quantiles<XLongGeo<WPL>, false, DQM, TOGP>(vals, rk, bits_all, nvalid, lane, wid, row_ptr, row_sorted, tog_ptr, tog_idx, togp, G, q, Q, kind,
                                                          mode, arr, c, af, out_hq, scaling, flag_cell, 0, nostamp);
}

// CE: pairs of elements per thread of the chunk sort — 11 (chunks of up to 22 528 steps) or 8 (up to 16 384: series of up to 32 768
// steps are two such chunks, and their sorts pay for 16 384 places instead of 22 528)
template <bool DQM, bool TOGP, int WPL, int CE = 11>
__global__ __launch_bounds__(1024) void k_train_xlong(
    const float *__restrict__ x0, const float *__restrict__ x1, int64_t T, int64_t sc, const int *__restrict__ row_ptr,
    const int *__restrict__ row_sorted, const int *__restrict__ tog_ptr, const int *__restrict__ tog_idx, int G,
    const float *__restrict__ q, int Q, int kind, int mode, float *__restrict__ af, float *out_hq, float *scaling,
    int *__restrict__ flag_count, int *__restrict__ flag_list, int64_t C, const int *__restrict__ togp,
    const int *__restrict__ pos_of, const int *__restrict__ time_of, unsigned char *__restrict__ gscratch, int debug_phase)
{
    // (row_sorted, tog_idx, togp: in POSITIONS, see RankGlobal)
    using L = LongCfg<CE>;
    using X = XLongGeo<WPL>;
    using u64 = unsigned long long;
    constexpr int NT = X::NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t *rank_l = reinterpret_cast<uint16_t *>(smem + L::off_rank);
    const uint16_t *sorted_l = reinterpret_cast<const uint16_t *>(smem + L::off_h);
    uint32_t *bits_all = reinterpret_cast<uint32_t *>(smem);
    uint32_t *tile_k = reinterpret_cast<uint32_t *>(smem);
    uint16_t *tile_t = reinterpret_cast<uint16_t *>(smem + (size_t)X::TILE * 4);
    uint32_t *red = reinterpret_cast<uint32_t *>(smem + X::off_misc);
    int *split = reinterpret_cast<int *>(smem + X::off_split);
    unsigned char *gs = gscratch + (size_t)blockIdx.x * X::scratch_bytes;
    u64 *run_a = reinterpret_cast<u64 *>(gs + X::g_run), *run_b = reinterpret_cast<u64 *>(gs + X::g_run2);
    u64 *chunk = reinterpret_cast<u64 *>(gs + X::g_chunk);
    uint16_t *rank_g = reinterpret_cast<uint16_t *>(gs + X::g_rank);
    float *sorted_v = reinterpret_cast<float *>(gs + X::g_sorted);

    const int tid = threadIdx.x, wid = tid >> 6;
    const int Ti = (int)T;
    const int K = (Ti + L::TPAD - 1) / L::TPAD;
    const int CL = (((Ti + K - 1) / K) + 1) & ~1;  // chunk length: even (the sort loads pairs), (K - 1) CL < T <= K CL
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        const int npass_arr = mode == 0 ? 1 : 2;
        bool flag_cell = false;
        for (int arr = 0; arr < npass_arr; arr++) {
            int lane = tid & 63;
            __asm__ volatile("" : "+v"(lane));
            const float *x = ((mode == 0 || arr == 1) ? x0 : x1) + c * sc;
            // ---------------- 1. chunk sorts and merges ----------------
            int n = 0;
            bool ranked = false;
            u64 *S = run_a, *O = run_b;
            for (int j = 0; j < K; j++) {
                const int base = j * CL, len = min(CL, Ti - base);
                const int nvj = long_load_sort<CE, false>(x + base, len, smem, red, rank_l, nullptr);
                // steps without a value are no members: rank 0xFFFF (behind the valid ranks of the chunk's order, among padding steps)
                if (nvj == 0) {  // (block-uniform; the sort has met at a barrier and left no order)
                    for (int t = tid; t < len; t += NT) rank_g[pos_of[base + t]] = 0xFFFFu;
                    continue;
                }
                for (int r = nvj + tid; r < L::TPAD; r += NT) {
                    const int t = (int)sorted_l[r];
                    if (t < len) rank_g[pos_of[base + t]] = 0xFFFFu;
                }
                u64 *dst = n == 0 ? S : chunk;
                write_run<NT>(x, base, len, sorted_l, nvj, dst, pos_of, reinterpret_cast<uint16_t *>(smem + (size_t)L::TPAD * 2 + 64), tid);
                stores_done();
                __syncthreads();  // the run is complete; the sort's LDS is free
                if (n == 0 || debug_phase == 2) { n += nvj; continue; }  // (debug_phase: timing aids of scripts/ubench/xlong_bench.hip)
                if (j == K - 1) {  // the last merge leaves rank[] and the rank-ordered steps instead of a run
                    merge_runs<NT, X::TILE, true>(S, n, chunk, nvj, O, rank_g, sorted_v, tile_k, tile_t, split, tid, debug_phase);
                    ranked = true;
                } else {
                    merge_runs<NT, X::TILE, false>(S, n, chunk, nvj, O, rank_g, sorted_v, tile_k, tile_t, split, tid, debug_phase);
                    u64 *t_ = S; S = O; O = t_;
                }
                n += nvj;
            }
            const int nvalid = n;
            if (debug_phase == 1 || debug_phase == 2 || debug_phase >= 4) { __syncthreads(); continue; }  // timing aids: sorts (2) and merges (1) only
            if (nvalid == 0) {
                // no valid value in the series (block-uniform): every quantile of the pass is NaN — the expressions of the walk
                const float nanq = nan_of<float>();
                const size_t ob = (size_t)c * G * (size_t)Q;
                for (int i = tid; i < G * Q; i += NT) {
                    if (mode == 0 || arr == 0) out_hq[ob + i] = nanq;
                    else af[ob + i] = get_corr<float>(out_hq[ob + i], nanq, kind);
                }
                if constexpr (DQM)
                    for (int g = tid; g < G; g += NT) {
                        float &s_ = scaling[(size_t)c * G + g];
                        s_ = arr == 0 ? nanq : get_corr<float>(s_, nanq, kind);
                    }
                __syncthreads();
                continue;
            }
            if (!ranked) {  // (the last chunk held no valid value, or only one did)
                for (int r0 = tid; r0 < nvalid; r0 += 8 * NT) {
                    u64 pr[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) pr[e] = load_l2(S + min(r0 + e * NT, nvalid - 1));
#pragma unroll
                    for (int e = 0; e < 8; e++)
                        if (r0 + e * NT < nvalid) {
                            rank_g[(uint32_t)pr[e] & 0xFFFFu] = (uint16_t)(r0 + e * NT);
                            sorted_v[r0 + e * NT] = key_to_float((uint32_t)(pr[e] >> 32));
                        }
                }
                stores_done();
                __syncthreads();
            }
            // rank[] and the rank-ordered time steps are complete; the LDS is the vectors' now
            if (debug_phase == 3) continue;
            // ---------------- 2. group quantiles ----------------
            if (wid < X::NWAVE)
                xlong_phase_c<DQM, TOGP, WPL>(x, Ti, sorted_v, rank_g, bits_all, nvalid, lane, wid, row_ptr, row_sorted, tog_ptr, tog_idx, togp, time_of, G, q, Q,
                                              kind, mode, arr, c, af, out_hq, scaling, flag_cell);
            __syncthreads();  // everyone is done with the vectors before the next sort reuses the LDS
        }
        const int any = __syncthreads_or(flag_cell ? 1 : 0);
        if (any && tid == 0 && flag_list) {
            const int slot = atomicAdd(flag_count, 1);
            flag_list[slot] = (int)c;
        }
    }
}

}  // namespace xs
