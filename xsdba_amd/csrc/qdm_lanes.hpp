// qdm_lanes.hpp — qdm_adjust for SHORT ranking rows (exact day-of-year groups: one member per year, <= 32 members), float32
// and float64 (template on the series type; the 64-bit keys of a double travel beside their payload instead of in one word).
//
// _adjustment.qdm_adjust (_adjustment.py:783-886) ranks every value of sim inside its group (utils.rank, utils.py:573-638:
// average ranks, NaN -> NaN, percent ranks rescaled with the extreme ranks) and looks the percent rank up on the quantile
// axis of the group's factor row.  k_qdm_adjust sorts those rows on the cross-lane path (one row per 32 lanes) and lets
// every element bisect its row in LDS; that is latency-bound (11.3 ms per 64 800 cells x 10 950 days).
//
// Here a LANE owns a group: its <= 32 members sit in 32 registers (coalesced loads — consecutive lanes, consecutive days of
// the year), Batcher's odd-even merge network sorts the (key, time index) pairs without leaving the registers (191
// compare-exchanges, no LDS, no cross-lane traffic), one forward and one backward sweep over the sorted registers give the
// tie runs, i.e. 2 * (average rank) = start + end + 1 of each element.  The map (N, 2r) -> table node is the same for every
// cell when the group's extreme values are not tied (pmin, pmax of the rescaling depend on N only): it is tabulated once per
// block with the very float64 expressions the generic kernel evaluates per element, so an element costs one table byte, one
// factor read and one multiply.  Groups with tied extremes evaluate those expressions directly (same results by
// construction).  The adjusted values go through a per-wave [member][lane] LDS buffer back to member order and every lane
// stores where it loaded from.  A wave works alone on (cell, 64 groups) items: no block barrier after the prologue.
//
// A wave whose factor rows hold a NaN (the nearest node of the element's own row is then not the 2-D nearest neighbour)
// appends the cell to flag_list; k_qdm_adjust redoes those cells, like the cells k_qdm_cell cannot finish.  Time steps in
// no group are not touched here: the host pre-fills the output with NaN when the grouping has any.
#pragma once
#include "kernels.hpp"

namespace xs {

constexpr int QL_M = 32;               // members per ranking row (registers per lane)
constexpr int QL_TW = 2 * QL_M + 2;    // row pitch of the node table: 2r lies in [2, 2N]; NaN rows reach N + 33
constexpr int QL_NAN = 255;            // node code: the factor is NaN (extrapolation = "nan", undefined percent rank)
constexpr int QL_CODES = ((QL_M + 1) * QL_TW + 15) & ~15;

// dynamic LDS of a block of `nwave` waves: the quantile axis and the node table (shared), then per wave the factor rows of its
// 64 groups and a [max_mem][64] buffer that takes the adjusted values back from sorted order to member order
__host__ __device__ inline size_t qdm_lanes_wave_bytes(int Q, int max_mem, size_t esz = 4) { return ((size_t)64 * Q + (size_t)max_mem * 64) * esz; }
__host__ __device__ inline size_t qdm_lanes_lds_bytes(int nwave, int Q, int max_mem, size_t esz = 4)
{
    return (((size_t)Q + 3) & ~(size_t)3) * esz + QL_CODES + (size_t)nwave * qdm_lanes_wave_bytes(Q, max_mem, esz);
}

// the node of the group's own row nearest to the rescaled percent rank: utils.rank's formula and interp2d_nearest
// (device_math.hpp) on a NaN-free row of the shared quantile axis, expression for expression
template <typename R>
__device__ __forceinline__ int ql_code(int twor, int N, double pmin, double pmax, const R *xq, int Q, int extrap)
{
    const double r = (double)twor / 2.0;  // = less + (equal + 1) / 2, exact
    const double pr = r / (double)N;
    const double sq = pmax * (pr - pmin) / (pmax - pmin);
    if (sq != sq) return QL_NAN;
    if (sq < (double)xq[0]) return extrap == 0 ? 0 : QL_NAN;
    if (sq > (double)xq[Q - 1]) return extrap == 0 ? Q - 1 : QL_NAN;
    int w;
    (void)row_nearest<R, double>(sq, xq, Q, &w);
    return w;
}

// every element NaN (time steps in no group are never touched by k_qdm_lanes)
template <typename R> __global__ void k_fill_nan(R *__restrict__ dst, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = nan_of<R>();
}

__device__ __forceinline__ void lds_wave_fence_ql() { __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Work item = (cell, chunk of 64 groups); a wave owns the items of ONE chunk (its member -> time table stays in registers,
// two 16-bit indexes per register) and walks the cells; waves never wait for each other after the prologue.
template <typename R>
__global__ __launch_bounds__(256, 2) void k_qdm_lanes(const R *__restrict__ sim, int64_t T, int64_t C, int64_t sc,
                                                       const R *__restrict__ af, const R *__restrict__ quantiles, int G, int Q,
                                                       const int *__restrict__ mem_t32, int max_mem, int extrap, int kind,
                                                       R *__restrict__ scen, int64_t osc, int *__restrict__ flag_count,
                                                       int *__restrict__ flag_list)
{
    using U = typename KeyOf<R>::U;
    constexpr U MAXK = KeyOf<R>::MAXK;
    constexpr U SIGN = (U)1 << (8 * sizeof(R) - 1);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    R *xq = reinterpret_cast<R *>(smem);                                                  // [Q] the quantile axis
    unsigned char *codes = smem + (((size_t)Q + 3) & ~(size_t)3) * sizeof(R);             // [QL_M + 1][QL_TW]
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wid = tid >> 6, nwave = nt >> 6;
    R *ysw = reinterpret_cast<R *>(codes + QL_CODES + (size_t)wid * qdm_lanes_wave_bytes(Q, max_mem, sizeof(R)));  // [64][Q] factor rows
    R *trw = ysw + (size_t)64 * Q;                                                        // [max_mem][64]

    // ---- once per block: the quantile axis and the (N, 2r) -> node table of groups whose extremes are not tied
    bool bad_axis = false;
    for (int i = tid; i < Q; i += nt) { const R v = quantiles[i]; xq[i] = v; bad_axis |= v != v; }
    __syncthreads();
    // (the own row holds the 2-D nearest node only while no node of it is farther than one group step)
    if (tid == 0 && !((double)xq[Q - 1] - (double)xq[0] <= 1.0)) bad_axis = true;
    bad_axis = __syncthreads_or(bad_axis ? 1 : 0) != 0;
    for (int i = tid; i < (QL_M + 1) * QL_TW; i += nt) {
        const int N = i / QL_TW, tw = i - N * QL_TW;
        int code = QL_NAN;
        if (!bad_axis && N >= 1 && tw >= 2 && tw <= 2 * N) {
            const double Nd = (double)N;
            const double pmin = ((1.0 + 1.0) / 2.0) / Nd, pmax = ((Nd - 1.0) + (1.0 + 1.0) / 2.0) / Nd;
            code = ql_code<R>(tw, N, pmin, pmax, xq, Q, extrap);
        }
        codes[i] = (unsigned char)code;
    }
    __syncthreads();

    const int nchunk = (G + 63) / 64;
    const int n_items = (int)C * nchunk, n_waves = (int)gridDim.x * nwave;  // (n_cell * n_chunk < 2^31: host-checked)
    uint32_t tt16[QL_M / 2];  // member slots -> time indexes of this lane's group, two 16-bit indexes per word, 0xFFFF: no such member
    // The words are needed when an item's loads are issued and again when its results are stored, not in between: they are
    // fetched anew (16 coalesced L2 hits) after the sort instead of occupying 16 registers through it — the pointer is made
    // opaque so that the second fetch is not folded into the first and spilled.
    auto fetch_slots = [&](int g) __attribute__((always_inline)) {
        const int *mp = mem_t32;
        __asm__ volatile("" : "+s"(mp));
#pragma unroll
        for (int y = 0; y < QL_M / 2; y++) tt16[y] = g < G ? (uint32_t)mp[(size_t)y * G + g] : 0xFFFFFFFFu;
    };
    int cur_chunk = -1;
    for (int item = (int)blockIdx.x * nwave + wid; item < n_items; item += n_waves) {
        const int ci = item / nchunk;
        const int chunk = item - ci * nchunk;
        const int64_t c = ci;
        // (the lane index is made opaque per item: what derives from it — table and row addresses — is recomputed here instead
        //  of being hoisted out of the item loop and spilled across the sort)
        int ln = lane;
        __asm__ volatile("" : "+v"(ln));
        const int g0 = chunk * 64, g = g0 + ln;
        if (chunk != cur_chunk) { cur_chunk = chunk; fetch_slots(g); }  // (else: still there from the previous item's stores)
        auto slot_time = [&](int y) { return (tt16[y / 2] >> (16 * (y & 1))) & 0xFFFFu; };
        // ---- loads: the members of the lane's group (consecutive lanes, consecutive days: coalesced) and the factor rows of
        // the wave's groups (one contiguous piece of the cell's table), all issued before anything is looked at
        const char *xrow = reinterpret_cast<const char *>(sim + c * sc);
        R x[QL_M];
#pragma unroll
        for (int y = 0; y < QL_M; y++) {
            const uint32_t t = slot_time(y);
            x[y] = *reinterpret_cast<const R *>(xrow + (t == 0xFFFFu ? 0u : t * (uint32_t)sizeof(R)));
        }
        const int nrow = min(64, G - g0), nfac = nrow * Q;
        bool has_nan = bad_axis;
        {
            const R *yq = af + ((size_t)c * G + g0) * Q;
            constexpr int NB = 20;  // (20 nodes: the wave's rows in one batch)
            for (int i0 = ln; i0 < nfac; i0 += NB * 64) {
                R yv[NB];
#pragma unroll
                for (int kq = 0; kq < NB; kq++) yv[kq] = yq[min(i0 + kq * 64, nfac - 1)];
#pragma unroll
                for (int kq = 0; kq < NB; kq++) {
                    const int i = i0 + kq * 64;
                    if (i < nfac) { ysw[i] = yv[kq]; has_nan |= yv[kq] != yv[kq]; }
                }
            }
        }
        // (every load is issued before the first value is looked at: pinned, or each one sinks into its `present` branch)
#pragma unroll
        for (int y = 0; y < QL_M; y++) __asm__ volatile("" : "+v"(x[y]));
        lds_wave_fence_ql();  // the rows were written across lanes
        if (__any(has_nan)) {
            // a NaN factor among the wave's rows: the own-row node need not be the 2-D nearest one — the generic kernel redoes
            // the cell (other waves of the cell may list it again: the list holds n_cell * n_chunk entries, the redo is idempotent)
            if (lane == 0) { const int slot = atomicAdd(flag_count, 1); flag_list[slot] = (int)c; }
            continue;
        }
        if constexpr (sizeof(R) == 8) {
            // ---- float64: ranks by COUNTING over the 496 pairs of the lane's members, all in registers (two compares and four
            // carry-adds per pair) — no sort network on 64-bit keys with a payload beside them (which the compiler either tears
            // apart or keeps in memory), no key transform (the comparisons treat -0.0 and +0.0 as equal and ignore NaN by
            // themselves), and the results come out in member order: no way back through LDS either.
            uint32_t less[QL_M], more[QL_M];
#pragma unroll
            for (int y = 0; y < QL_M; y++) {
                if (slot_time(y) == 0xFFFFu) x[y] = nan_of<R>();  // no such member: counts for nobody
                less[y] = 0u; more[y] = 0u;
            }
#pragma unroll
            for (int i = 0; i < QL_M; i++)
#pragma unroll
                for (int j = i + 1; j < QL_M; j++) {
                    const uint32_t c1 = x[i] < x[j] ? 1u : 0u, c2 = x[j] < x[i] ? 1u : 0u;
                    less[j] += c1; more[i] += c1;
                    less[i] += c2; more[j] += c2;
                }
            int N = 0;
#pragma unroll
            for (int y = 0; y < QL_M; y++) N += x[y] == x[y] ? 1 : 0;
            // tie counts of the smallest / largest value: every member with nothing below (above) it shares them
            int nmin = 1, nmax = 1;
#pragma unroll
            for (int y = 0; y < QL_M; y++) {
                const bool ok = x[y] == x[y];
                const int eq = N - (int)less[y] - (int)more[y];
                nmin = (ok && less[y] == 0u) ? eq : nmin;
                nmax = (ok && more[y] == 0u) ? eq : nmax;
            }
            const bool untied = nmin == 1 && nmax == 1;
            int ln2 = lane;
            __asm__ volatile("" : "+v"(ln2));
            const int gq = ln2 * Q;
            char *orow = reinterpret_cast<char *>(scen + c * osc);
            auto finish = [&](R xv, uint32_t t, int code) __attribute__((always_inline)) {
                const R a = code == QL_NAN ? nan_of<R>() : ysw[gq + (code == QL_NAN ? 0 : code)];
                if (t != 0xFFFFu) *reinterpret_cast<R *>(orow + t * (uint32_t)sizeof(R)) = apply_corr<R>(xv, a, kind);
            };
            if (!__any(!untied)) {
                const unsigned char *crow = codes + N * QL_TW;
#pragma unroll
                for (int y = 0; y < QL_M; y++) {
                    const int twor = 2 * (int)less[y] + (N - (int)less[y] - (int)more[y]) + 1;
                    finish(x[y], slot_time(y), x[y] == x[y] ? (int)crow[min(twor, QL_TW - 1)] : QL_NAN);
                }
            } else {
                const double Nd = (double)N;
                const double pmin = (((double)nmin + 1.0) / 2.0) / Nd;
                const double pmax = ((Nd - (double)nmax) + ((double)nmax + 1.0) / 2.0) / Nd;
#pragma unroll 1
                for (int j = 0; j < QL_M; j++) {
                    int jo = j;
                    __asm__ volatile("" : "+s"(jo));
                    R xv = (R)0;
                    uint32_t le = 0u, mo = 0u, t = 0xFFFFu;
#pragma unroll
                    for (int u = 0; u < QL_M; u++) { xv = u == jo ? x[u] : xv; le = u == jo ? less[u] : le; mo = u == jo ? more[u] : mo; t = u == jo ? slot_time(u) : t; }
                    const int twor = 2 * (int)le + (N - (int)le - (int)mo) + 1;
                    finish(xv, t, (xv == xv && twor <= 2 * N) ? ql_code<R>(twor, N, pmin, pmax, xq, Q, extrap) : QL_NAN);
                }
            }
            continue;
        }
        U k[QL_M];
        uint32_t p[QL_M];
#pragma unroll
        for (int y = 0; y < QL_M; y++) {
            const bool present = slot_time(y) != 0xFFFFu;
            U bits;
            if constexpr (sizeof(R) == 4) bits = __float_as_uint(x[y]); else bits = (U)__double_as_longlong(x[y]);
            // -0.0 ranks as +0.0 (scipy compares values); its sign rides in bit 31 of the payload and returns at the end
            k[y] = present ? to_key(x[y] + (R)0) : MAXK;
            p[y] = present ? ((uint32_t)y | (bits == SIGN ? 0x80000000u : 0u)) : 0xFFFFFFu;  // payload: the member slot
        }
        // ---- Batcher's odd-even merge sort of the (key, payload) pairs, in registers.  A pair travels as ONE 64-bit word,
        // key on top (equal keys are then ordered by payload, which no result depends on): a compare-exchange is one
        // 64-bit compare and four selects, and the key and payload networks cannot drift apart in the schedule (as two
        // 32-bit networks the compiler ran all the key min / max first and kept every intermediate key alive for the
        // payload selects: 500 live registers)
        if constexpr (sizeof(R) == 4) {
            uint64_t kp[QL_M];
#pragma unroll
            for (int y = 0; y < QL_M; y++) kp[y] = ((uint64_t)k[y] << 32) | p[y];
#pragma unroll
            for (int pp = 1; pp < QL_M; pp *= 2)
#pragma unroll
                for (int kk = pp; kk >= 1; kk /= 2)
#pragma unroll
                    for (int j = kk % pp; j <= QL_M - 1 - kk; j += 2 * kk)
#pragma unroll
                        for (int i = 0; i <= (kk - 1 < QL_M - j - kk - 1 ? kk - 1 : QL_M - j - kk - 1); i++)
                            if ((i + j) / (2 * pp) == (i + j + kk) / (2 * pp)) {
                                const uint64_t va = kp[i + j], vb = kp[i + j + kk];
                                const bool sw = vb < va;
                                kp[i + j] = sw ? vb : va;
                                kp[i + j + kk] = sw ? va : vb;
                                // (two exchanges in flight at most: unbounded, the sixteen of a layer keep ~64 more registers busy)
                                if (((i + j) & 1) == 1 || kk == 1) __builtin_amdgcn_sched_barrier(0);
                            }
#pragma unroll
            for (int y = 0; y < QL_M; y++) { k[y] = (U)(kp[y] >> 32); p[y] = (uint32_t)kp[y]; }
        } else {
            // 64-bit keys: key and payload side by side, ONE compare per exchange decides both (no min / max on the keys alone,
            // which would let the two networks drift apart)
#pragma unroll
            for (int pp = 1; pp < QL_M; pp *= 2)
#pragma unroll
                for (int kk = pp; kk >= 1; kk /= 2)
#pragma unroll
                    for (int j = kk % pp; j <= QL_M - 1 - kk; j += 2 * kk)
#pragma unroll
                        for (int i = 0; i <= (kk - 1 < QL_M - j - kk - 1 ? kk - 1 : QL_M - j - kk - 1); i++)
                            if ((i + j) / (2 * pp) == (i + j + kk) / (2 * pp)) {
                                const U ka = k[i + j], kb = k[i + j + kk];
                                const uint32_t pa = p[i + j], pb = p[i + j + kk];
                                const bool sw = kb < ka;
                                U nka = sw ? kb : ka, nkb = sw ? ka : kb;
                                uint32_t npa = sw ? pb : pa, npb = sw ? pa : pb;
                                // (all four results are pinned here — as locals: an array element in an asm operand keeps the array
                                //  in memory —: the payload selects may not be deferred behind the key network)
                                __asm__ volatile("" : "+v"(nka), "+v"(nkb), "+v"(npa), "+v"(npb));
                                k[i + j] = nka; k[i + j + kk] = nkb; p[i + j] = npa; p[i + j + kk] = npb;
                            }
        }
        int ln2 = lane;
        __asm__ volatile("" : "+v"(ln2));
        fetch_slots(g0 + ln2);  // (in flight under the sweeps)
        // ---- forward sweep: start of every tie run (bits 24-28 of the payload), valid count, start of the last valid run
        int N = 0, st = 0, last_start = 0;
#pragma unroll
        for (int j = 0; j < QL_M; j++) {
            if (j > 0) st = (k[j] == k[j - 1]) ? st : j;
            p[j] |= (uint32_t)st << 24;
            const bool valid = k[j] != MAXK;
            N += valid ? 1 : 0;
            last_start = valid ? st : last_start;
        }
        const bool untied = (N < 2 || k[1] != k[0]) && (N < 1 || last_start == N - 1);
        const int gq = ln2 * Q;  // the lane's row in the wave's factor rows
        // ---- backward sweep: end of every tie run; 2r = start + end + 1 replaces the start in bits 24-30 of the payload
        {
            int e = QL_M;
#pragma unroll
            for (int j = QL_M - 1; j >= 0; j--) {
                if (j < QL_M - 1) e = (k[j] == k[j + 1]) ? e : j + 1;
                const uint32_t twor = ((p[j] >> 24) & 31u) + (uint32_t)e + 1u;
                p[j] = (p[j] & 0x80FFFFFFu) | (twor << 24);  // (an absent slot keeps its 24 low bits set)
            }
        }
        auto park = [&](U kj, uint32_t pj, int code) __attribute__((always_inline)) {
            const R a = code == QL_NAN ? nan_of<R>() : ysw[gq + (code == QL_NAN ? 0 : code)];
            R v;
            if constexpr (sizeof(R) == 4) v = __uint_as_float(__float_as_uint(from_key(kj)) | (pj & 0x80000000u));
            else v = __longlong_as_double(__double_as_longlong(from_key(kj)) | (long long)((uint64_t)(pj & 0x80000000u) << 32));
            const R out = apply_corr<R>(v, a, kind);
            if ((pj & 0xFFFFFFu) != 0xFFFFFFu) trw[(pj & 31u) * 64u + (uint32_t)ln2] = out;
        };
        if (!__any(!untied)) {
            // 2r -> node from the block's table -> factor -> adjusted value, parked by member slot
            const unsigned char *crow = codes + N * QL_TW;
#pragma unroll
            for (int j = 0; j < QL_M; j++) park(k[j], p[j], (int)crow[(p[j] >> 24) & 127u]);
        } else {
            // some group of this wave has tied extremes: its rescaling differs, the float64 expressions are evaluated per
            // element (a rolled loop over the places; the place is wave-uniform, the registers are picked by a select chain)
            int nmin = 1;
            bool eq = true;
#pragma unroll
            for (int j = 1; j < QL_M; j++) { eq = eq && k[j] == k[0]; nmin += eq ? 1 : 0; }
            const int nmax = N - last_start;
            const double Nd = (double)N;
            const double pmin = (((double)nmin + 1.0) / 2.0) / Nd;
            const double pmax = ((Nd - (double)nmax) + ((double)nmax + 1.0) / 2.0) / Nd;
#pragma unroll 1
            for (int j = 0; j < QL_M; j++) {
                U kj = 0;
                uint32_t pj = 0u;
                int jo = j;
                __asm__ volatile("" : "+s"(jo));  // (opaque: with a visible induction variable the chain below is folded into k[j] and the arrays go to memory)
#pragma unroll
                for (int u = 0; u < QL_M; u++) { kj = u == jo ? k[u] : kj; pj = u == jo ? p[u] : pj; }
                const int twor = (int)((pj >> 24) & 127u);
                const int code = (kj != MAXK && twor <= 2 * N) ? ql_code<R>(twor, N, pmin, pmax, xq, Q, extrap) : QL_NAN;
                park(kj, pj, code);
            }
        }
        // ---- back in member order: every lane stores where it loaded from (coalesced the same way)
        lds_wave_fence_ql();
        {
            char *orow = reinterpret_cast<char *>(scen + c * osc);
            R o[QL_M];
#pragma unroll
            for (int y = 0; y < QL_M; y++) o[y] = trw[(y < max_mem ? y : 0) * 64 + ln2];
#pragma unroll
            for (int y = 0; y < QL_M; y++) {
                const uint32_t t = slot_time(y);
                if (t != 0xFFFFu) *reinterpret_cast<R *>(orow + t * (uint32_t)sizeof(R)) = o[y];
            }
        }
        lds_wave_fence_ql();  // (the next item refills the wave's rows)
    }
}

}  // namespace xs
