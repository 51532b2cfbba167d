// qdm_cell.hpp — fast QDM adjust / rank kernel: one 1024-thread workgroup per grid cell, float32 and float64.
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
#include "train_cell_f64.hpp"

// timing aid for experiment builds only (-DXS_QDM_STOP=n: 1 = stop a cell after the load + sort, 2 = after the tie bits / tables,
// 3 = after the group walk); the release library is built without it
#ifndef XS_QDM_STOP
#define XS_QDM_STOP 0
#endif

namespace xs {

template <int EPT> struct QdmCfg {
    using B = CellCfg<EPT>;
    static constexpr int GMAX = 512, QMAX = 64;
    static constexpr size_t off_bits = B::off_rank + (size_t)B::TPAD * 2;          // per wave: BITW dwords
    static constexpr size_t off_pw = off_bits + (size_t)B::NWAVE * B::BITW * 4;     // per wave: BITW u16 prefix counts
    static constexpr size_t sel_end = off_pw + (size_t)B::NWAVE * B::BITW * 2;
    static constexpr size_t off_gstat = ((B::sort_end > sel_end ? B::sort_end : sel_end) + 15) & ~(size_t)15;
    static constexpr size_t off_tab = off_gstat + (size_t)GMAX * 3 * 8;             // N, pmin, pmax per group (float64)
    static constexpr size_t off_red = off_tab + (size_t)(2 * QMAX + 2) * 8 + 16;    // nodes, 1-D table, its metadata (float64-sized)
    static constexpr size_t lds_bytes = off_red + 64 * 8;
};

// How the kernel sees the values of the cell's series after the sort.  float32: the order-preserving keys in LDS ARE the
// values.  float64: the LDS keys are 31-bit range keys (train_cell_f64.hpp); equal keys do not imply equal values, so
// equality is decided on the doubles (read from global memory / L2) whenever the keys are within one step of each other
// (-0.0 and +0.0 are one key step apart but equal), and values are read from the series itself.
// glob: the keys have left the LDS (the second group stream of the walk lives there, see below): values and equality come from
// the series itself (global memory / L2) — the same floats the keys were made of.
template <typename R> struct QdmSeries;
template <> struct QdmSeries<float> {
    const float *x;
    const uint32_t *keys;
    const uint16_t *sorted;
    bool glob;
    __device__ __forceinline__ float at_time(int t) const { return glob ? x[t] : from_key(keys[t]); }
    __device__ __forceinline__ bool eq_ranks(int a, int b) const
    {
        if (glob) return x[sorted[a]] == x[sorted[b]];
        return from_key(keys[sorted[a]]) == from_key(keys[sorted[b]]);
    }
};
template <> struct QdmSeries<double> {
    const double *x;
    const uint32_t *keys;
    const uint16_t *sorted;
    bool glob;
    __device__ __forceinline__ double at_time(int t) const { return x[t]; }
    __device__ __forceinline__ bool eq_ranks(int a, int b) const
    {
        const int ta = sorted[a], tb = sorted[b];
        if (!glob) {
            const uint32_t ka = keys[ta], kb = keys[tb];
            if (ka != kb && ka + 1u != kb && kb + 1u != ka) return false;
        }
        return x[ta] == x[tb];
    }
};

template <int EPT, typename R = float>
__global__ __launch_bounds__(1024) void k_qdm_cell(
    const R *__restrict__ sim, int64_t T, int64_t C, int64_t sc, const R *__restrict__ af,
    const R *__restrict__ quantiles, int G, int Q, const int *__restrict__ gidx,
    const int *__restrict__ grp_ptr, const int *__restrict__ grp_idx,  // ranking rows (windowed rows or exact groups)
    const int *__restrict__ mem_ptr, const int *__restrict__ mem_idx,  // exact groups: the elements each row ranks
    const int *__restrict__ tog_ptr, const int *__restrict__ tog_idx, const int *__restrict__ tog64,
    int use_tog,  // bit 0: consecutive ranking rows differ by the toggle lists (0: rows are the (disjoint) exact groups);
                  // bit 1: the ranking rows ARE the exact groups (no window), at most 64 of them: ranks by the group-keyed pass (C3)
    int interp, int extrap, int kind, R *__restrict__ scen, double *__restrict__ sim_q,
    int *__restrict__ flag_count, int *__restrict__ flag_list, int max_mem /* most elements any row ranks */,
    const int *__restrict__ mem_t32 /* [16][G] member slots, two 16-bit time steps per word (0xFFFF: none), or null */)
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
    R *qn = reinterpret_cast<R *>(smem + QC::off_tab);  // quantile nodes (grouped) / compacted nodes (1-D)
    R *ty = qn + QC::QMAX;                              // compacted factors of the 1-D table
    R *tyfl = ty + QC::QMAX;                            // [0] yfirst, [1] ylast of the 1-D table
    int *tmeta = reinterpret_cast<int *>(tyfl + 2);     // [0] pairs, [3] bad nodes
    uint32_t *red = reinterpret_cast<uint32_t *>(smem + QC::off_red);

    const int tid = threadIdx.x, wid = tid >> 6;
    uint32_t *bits = reinterpret_cast<uint32_t *>(smem + QC::off_bits) + wid * BITW;
    uint16_t *pw = reinterpret_cast<uint16_t *>(smem + QC::off_pw) + wid * BITW;

    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        int lane = tid & 63;  // opaque per cell: lane-derived addresses are recomputed, not spilled across the sort
        __asm__ volatile("" : "+v"(lane));
        int nvalid;
        const uint16_t *sorted;
        if constexpr (sizeof(R) == 8) sorted = cell_load_sort_f64<EPT>(sim + c * sc, T, smem, red, nvalid);
        else {
            sorted = cell_load_sort<EPT>(sim + c * sc, T, smem, red, 0, nvalid, NoLoadHook(), -1, true);
            if (!sorted) sorted = idx1;  // (no valid value: not sorted; nothing below reads the order when nvalid == 0)
        }
        if constexpr (XS_QDM_STOP == 1) { if (sorted[0] == 0xFFFF && tid == 0) flag_count[1] = 1; __syncthreads(); continue; }
        uint16_t *r2buf = (sorted == idx1) ? idx0 : idx1;  // 2 * average rank per time step, 0 = no rank
        // two group streams per wave (see C2 below) when the rows walk by the padded toggle table, every wave has groups of its own
        // and a row ranks at most 30 elements (30 years of a day-of-year window): the second stream's vectors take the place of the
        // keys, which are dead once the tie bits are formed — what still needs a value reads the series itself
        const bool have_tb = (size_t)(TPAD / 8) <= (size_t)(QC::GMAX - G) * 24;
        const bool two = have_tb && G >= NWAVE && (use_tog & 1) && tog64 != nullptr && max_mem <= 30 && mem_t32 != nullptr;
        const QdmSeries<R> ser{sim + c * sc, keys, sorted, two}, ser_lds{sim + c * sc, keys, sorted, false};
        // the run of equal values around rank r (the order is non-decreasing, so equality with r is monotone on each side):
        // first rank in [0, r] whose value equals the one at r ...
        auto run_begin = [&](int r) {
            int lo = 0, hi = r;
            while (lo < hi) { int mid = (lo + hi) >> 1; if (ser.eq_ranks(mid, r)) hi = mid; else lo = mid + 1; }
            return lo;
        };
        // ... and first rank in (r, nvalid] whose value differs (nvalid if none)
        auto run_end = [&](int r) {
            int lo = r + 1, hi = nvalid;
            while (lo < hi) { int mid = (lo + hi) >> 1; if (ser.eq_ranks(mid, r)) lo = mid + 1; else hi = mid; }
            return lo;
        };
        // tie bits: bit r set when the values at ranks r - 1 and r are equal — formed once per cell, in parallel, in the unused
        // tail of gstat (shared with pass D's node table, which is built later): a tie test in the group walk is then one LDS
        // read instead of two dependent index -> key round trips per side (the walk waits on LDS latency, not on issue)
        uint32_t *tieb = reinterpret_cast<uint32_t *>(gstat + 3 * G);
        auto tbit = [&](int r) { return (tieb[r >> 5] >> (r & 31)) & 1u; };
        auto tie_run = [&](int r, int &rs, int &re) {
            rs = r; re = r + 1;
            if (have_tb) {
                const bool tp = r > 0 && tbit(r), tn = re < nvalid && tbit(re);
                if (tp) rs = run_begin(r);
                if (tn) re = run_end(r);
            } else {
                if (r > 0 && ser.eq_ranks(r - 1, r)) rs = run_begin(r);
                if (re < nvalid && ser.eq_ranks(re, r)) re = run_end(r);
            }
        };
        // extreme percent ranks of a group (utils.rank :631-635: mn / mx of rank / N with average ties), once per group
        // (phase C parks the three counts of a group; the float64 divisions are done for all groups at once after it, by a
        //  lane per group, instead of once per group step by a whole wave)
        // (the counts are parked as integers in the low words of the group's three float64 slots: the conversions wait for the
        //  lane-per-group pass after the walk instead of running, wave-wide, in every group step)
        auto put_stats = [&](int g, int n, int nmin, int nmax) {
            int *gi = reinterpret_cast<int *>(gstat + 3 * g);
            gi[0] = n;
            gi[2] = nmin;
            gi[4] = nmax;
        };
        {
            uint4 *z = reinterpret_cast<uint4 *>(r2buf);  // (16-byte aligned: both index buffers are)
            for (int i = tid; i < TPAD / 8; i += NT) z[i] = make_uint4(0u, 0u, 0u, 0u);
        }
        if (have_tb) {
            // all the index -> key reads of a thread's places first (independent), then the ballots and the stores
            uint32_t eqm = 0u;
#pragma unroll
            for (int j = 0; j < TPAD / NT; j++) {
                const int r = j * NT + tid;
                eqm |= (uint32_t)(r > 0 && r < nvalid && ser_lds.eq_ranks(r - 1, r)) << j;
            }
#pragma unroll
            for (int j = 0; j < TPAD / NT; j++) {
                const int r = j * NT + tid;
                const unsigned long long b = __ballot((eqm >> j) & 1u);
                if (lane == 0) { tieb[r >> 5] = (uint32_t)b; tieb[(r >> 5) + 1] = (uint32_t)(b >> 32); }
            }
        }
        if (tid == 0) *reinterpret_cast<int *>(gstat + 3 * (G / 2)) = 0;  // (the middle group seeds the node table of pass D: 0 = no valid value this cell)
        // interpolation table: the nodes (grouped) or the compacted (node, factor) pairs of the only row (1-D)
        if (af && wid == 0) {
            const R xq = lane < Q ? quantiles[lane] : nan_of<R>();
            if (G > 1) {
                if (lane < Q) qn[lane] = xq;
                const unsigned long long bad = __ballot(lane < Q && xq != xq);
                const R xlo = __shfl(xq, 0, 64), xhi = __shfl(xq, Q - 1, 64);
                if (lane == 0) tmeta[3] = (bad != 0ull) || !(xhi - xlo <= (R)1);
            } else {
                const R yq = lane < Q ? af[(size_t)c * Q + lane] : nan_of<R>();
                const unsigned long long bx = __ballot(xq == xq), by = __ballot(yq == yq), bxy = bx & by;
                const int pos = __popcll(bxy & ((1ull << lane) - 1ull));
                if ((bxy >> lane) & 1ull) { qn[pos] = xq; ty[pos] = yq; }
                const R yf = __shfl(yq, by ? __ffsll((long long)by) - 1 : 63, 64);
                const R yl = __shfl(yq, by ? 63 - __clzll((long long)by) : 63, 64);
                if (lane == 0) { tmeta[0] = __popcll(bxy); tyfl[0] = yf; tyfl[1] = yl; tmeta[3] = 0; }
            }
        }
        __syncthreads();

        if constexpr (XS_QDM_STOP == 2) { if (r2buf[5] == 0xFFFF && tid == 0) flag_count[1] = 1; __syncthreads(); continue; }
        const bool all_time = G == 1 && grp_ptr[1] - grp_ptr[0] == (int)T && mem_ptr[1] - mem_ptr[0] == (int)T;
        if (all_time) {
            // ---------------- C0. one group holding every time step: the sorted order is the ranking ----------------
            for (int r = tid; r < nvalid; r += NT) {
                int rs, re;
                tie_run(r, rs, re);
                r2buf[sorted[r]] = (uint16_t)(rs + re + 1);
            }
            if (tid == 0 && nvalid) {
                put_stats(0, nvalid, run_end(0), nvalid - run_begin(nvalid - 1));
            }
        } else if (nvalid && (use_tog & 2)) {
            // ---------------- C3. un-windowed groups (time.month, time.season, ...): a group-keyed stable pass ----------------
            // The ranking rows are the groups themselves and every valid step belongs to one: ONE more stable counting pass of the
            // sorted order, keyed by the group, lays the ranks out group by group in value order — an element's place in its
            // group's segment IS the number of members below it.  No bit-vector is built or walked (the walk built 912-member
            // vectors per wave and ranked by prefix popcounts: 4.8 of 14.2 ms per 64 800 monthly cells).  Ties: a run of equal
            // values inside a segment is a run of unflagged places, a place being flagged when a segment starts there or its
            // value differs from its predecessor's; run start / run end of every place by a forward max-scan / backward
            // min-scan of the flagged places over the block, the segment start the same way.
            constexpr int RND = TPAD / NT, GS = 72;
            // (LDS: behind rank[], where the walks keep their vectors and prefix tables and the sort kept its histogram — all dead here)
            static_assert(QC::off_bits + (size_t)TPAD * 2 + (size_t)(NWAVE * GS + TPAD / 32 + 72) * 4 <= QC::off_gstat, "ranks by place, counters, flags, segment starts");
            uint16_t *GR = reinterpret_cast<uint16_t *>(smem + QC::off_bits);   // [place] -> global rank
            uint32_t *wc = reinterpret_cast<uint32_t *>(GR + TPAD);             // [NWAVE][GS] wave-private counters -> base places
            uint32_t *SB = wc + NWAVE * GS;                                     // [TPAD / 32] a segment starts at this place
            int *seg = reinterpret_cast<int *>(SB + TPAD / 32);                 // [G + 1] first place of every group's segment
            for (int i = tid; i < NWAVE * GS + TPAD / 32; i += NT) wc[i] = 0u;
            __syncthreads();
            uint32_t *wrow = wc + wid * GS;
            int gq[RND];
            uint32_t od[RND];
#pragma unroll
            for (int j = 0; j < RND; j++) {
                const int r = wid * (64 * RND) + 64 * j + lane;
                gq[j] = r < nvalid ? gidx[sorted[r]] : -1;
            }
            // (a wave's returning atomics are served in program order, the lanes of one instruction that hit the same word in
            //  ascending lane order — train_cell.hpp: the pass is stable)
#pragma unroll
            for (int j = 0; j < RND; j++) od[j] = gq[j] >= 0 ? atomicAdd(&wrow[gq[j]], 1u) : 0u;
            __syncthreads();
            {   // exclusive scan of the counters in (group, wave) order: thread i owns counter (group i >> 4, wave i & 15)
                const bool own = tid < NWAVE * G;
                const uint32_t cv = own ? wc[(tid & 15) * GS + (tid >> 4)] : 0u;
                const int incl = wave_incl_scan((int)cv);
                if (lane == 63) red[wid] = (uint32_t)incl;
                __syncthreads();
                const int pre = waves_before(red, lane, wid) + incl - (int)cv;
                if (own) {
                    wc[(tid & 15) * GS + (tid >> 4)] = (uint32_t)pre;
                    if ((tid & 15) == 0) { seg[tid >> 4] = pre; atomicOr(&SB[pre >> 5], 1u << (pre & 31)); }
                    if (tid == NWAVE * G - 1) { seg[G] = pre + (int)cv; atomicOr(&SB[(pre + (int)cv) >> 5], 1u << ((pre + (int)cv) & 31)); }
                }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < RND; j++)
                if (gq[j] >= 0) GR[wrow[gq[j]] + od[j]] = (uint16_t)(wid * (64 * RND) + 64 * j + lane);
            __syncthreads();
            const int ntot = seg[G];  // the ranked steps (valid and in a group)
            // a thread owns RND consecutive places
            const int p0 = tid * RND;
            int rr[RND];
            uint32_t fl = 0u, sbm = 0u;  // bit j: place p0 + j is flagged / starts a segment
            {
                int rprev = (p0 > 0 && p0 <= ntot) ? (int)GR[p0 - 1] : -1;
#pragma unroll
                for (int j = 0; j < RND; j++) {
                    const int p = p0 + j;
                    rr[j] = p < ntot ? (int)GR[p] : -1;
                    const bool sb = p <= ntot && ((SB[p >> 5] >> (p & 31)) & 1u);
                    bool f = sb || p == ntot;
                    if (!f && p < ntot) f = !ser_lds.eq_ranks(rprev, rr[j]);
                    fl |= (uint32_t)f << j;
                    sbm |= (uint32_t)sb << j;
                    rprev = rr[j];
                }
            }
            // block-wide scans of "the last flagged place at or before" (max, forward) and "the first flagged place behind" (min,
            // backward): per thread over its places, per wave on the DPP path (values + 1 so that 0 is "none"), across the waves
            // through `red`
            auto wave_max_scan = [&](int v) {  // inclusive, v >= 0
                v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false));
                v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false));
                v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false));
                v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false));
                v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false));
                v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false));
                return v;
            };
            auto block_excl_max = [&](int mine, uint32_t *scr) {  // max of `mine` (>= 0) over the threads before this one
                const int incl = wave_max_scan(mine);
                int ex = __builtin_amdgcn_ds_bpermute(((lane + 63) & 63) * 4, incl);
                ex = lane ? ex : 0;
                if (lane == 63) scr[wid] = (uint32_t)incl;
                __syncthreads();
                int carry = 0;
                for (int w = 0; w < NWAVE; w++) carry = w < wid ? max(carry, (int)scr[w]) : carry;
                __syncthreads();
                return max(ex, carry);
            };
            int lastf = 0, lasts = 0;  // (place + 1) of the last flagged place / segment start among this thread's, 0: none
#pragma unroll
            for (int j = 0; j < RND; j++) {
                if ((fl >> j) & 1u) lastf = p0 + j + 1;
                if ((sbm >> j) & 1u) lasts = p0 + j + 1;
            }
            const int carf = block_excl_max(lastf, red), cars = block_excl_max(lasts, red);
            // backward: the first flagged place of this thread's (as BIG - place so that the max machinery serves), mirrored lanes
            constexpr int BIG = 1 << 20;
            int firstf = 0;
#pragma unroll
            for (int j = RND - 1; j >= 0; j--)
                if ((fl >> j) & 1u) firstf = BIG - (p0 + j);
            int nextf;  // the first flagged place behind this thread's places
            {
                const int mir = __builtin_amdgcn_ds_bpermute((63 - lane) * 4, firstf);
                const int inclm = wave_max_scan(mir);                       // lane l: max over mirrored lanes <= l = original lanes >= 63 - l
                int exm = __builtin_amdgcn_ds_bpermute(((lane + 63) & 63) * 4, inclm);
                exm = lane ? exm : 0;                                       // mirrored lane l: original lanes > 63 - l
                const int ex = __builtin_amdgcn_ds_bpermute((63 - lane) * 4, exm);
                const int wtot = __builtin_amdgcn_readlane(inclm, 63);
                if (lane == 0) red[wid] = (uint32_t)wtot;
                __syncthreads();
                int carry = 0;
                for (int w = 0; w < NWAVE; w++) carry = w > wid ? max(carry, (int)red[w]) : carry;
                __syncthreads();
                const int best = max(ex, carry);
                nextf = best ? BIG - best : ntot;
            }
            {
                int curf = carf, curs = cars, rs[RND], sg[RND];
#pragma unroll
                for (int j = 0; j < RND; j++) {
                    if ((fl >> j) & 1u) curf = p0 + j + 1;
                    if ((sbm >> j) & 1u) curs = p0 + j + 1;
                    rs[j] = curf - 1; sg[j] = curs - 1;
                }
                int nx = nextf;
#pragma unroll
                for (int j = RND - 1; j >= 0; j--) {
                    const int p = p0 + j;
                    if (p < ntot) {
                        const int re = nx;  // the run of place p ends where the next flagged place is
                        const int t = (int)sorted[rr[j]];
                        r2buf[t] = (uint16_t)((rs[j] - sg[j]) + (re - sg[j]) + 1);
                        const bool first = (sbm >> j) & 1u, lastp = (SB[(p + 1) >> 5] >> ((p + 1) & 31)) & 1u;
                        if (first || lastp) {
                            // the group of this place: the last one whose segment starts at or before it
                            int lo = 0, hi = G - 1;
                            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (seg[mid] <= p) lo = mid; else hi = mid - 1; }
                            int *gi = reinterpret_cast<int *>(gstat + 3 * lo);
                            if (first) { gi[0] = seg[lo + 1] - seg[lo]; gi[2] = re - p; }
                            if (lastp) gi[4] = p + 1 - rs[j];
                        }
                    }
                    if ((fl >> j) & 1u) nx = p;
                }
            }
        } else if (nvalid && two) {
            // ---------------- C2. the walk of C with TWO group streams per wave ----------------
            // A step of the walk is a chain of dependent LDS round trips (toggle -> recount -> prefix table -> member ranks) that
            // keeps a wave's issue slot empty most of the time (SQ counters: 40 % VALU issue).  The wave's groups [g0, g1) are
            // walked OUTWARD from gm with two vectors in lock-step: stream A takes gm, gm + 1, ... (lanes 0-31 rank its
            // elements), stream B gm - 1, gm - 2, ... (lanes 32-63).  One build (group gm) + a copy; the XOR lists work in both
            // directions.  Lanes 30 / 31 of a half rank the first / last member of the row (the extreme tie counts).
            const int gpc = (G + NWAVE - 1) / NWAVE;
            const int g0 = min(G, wid * gpc), g1 = min(G, g0 + gpc);
            const int nB = (g1 - g0) / 2, nA = (g1 - g0) - nB, gm = g0 + nB;  // nA >= nB
            // LDS: A's vectors where the one-stream walk keeps its own; the per-word prefix counts of both streams packed in one
            // dword (A low, B high 16 bits: one store per word, one bit-field extract to read) in the key array; B's vectors in
            // the one-stream walk's prefix tables (waves 0-7) and behind the packed table in the key array (waves 8-15)
            uint32_t *bitsA = bits;
            uint32_t *pw2 = reinterpret_cast<uint32_t *>(smem + Cfg::off_keys) + wid * BITW;
            uint32_t *bitsB = wid < NWAVE / 2 ? reinterpret_cast<uint32_t *>(smem + QC::off_pw) + wid * BITW
                                              : reinterpret_cast<uint32_t *>(smem + Cfg::off_keys + (size_t)NWAVE * BITW * 4) + (wid - NWAVE / 2) * BITW;
            static_assert((size_t)NWAVE * BITW * 4 + (size_t)(NWAVE / 2) * BITW * 4 <= (size_t)TPAD * 4, "the packed table and half of B's vectors fit in the key array");
            static_assert((size_t)(NWAVE / 2) * BITW * 4 <= (size_t)NWAVE * BITW * 2, "the other half of B's vectors fit in the one-stream prefix tables");
            const int half = lane >> 5, l5 = lane & 31;
            const uint32_t *bits_h = half ? bitsB : bitsA;
            const unsigned pwsh = 16u * (unsigned)half;
            if (nA > 0) {
                for (int i = lane; i < BITW; i += 64) bitsA[i] = 0u;
                lds_wave_fence();
                {   // the vector of group gm from its row list, copied to B
                    const int *l = grp_idx + grp_ptr[gm];
                    const int len = grp_ptr[gm + 1] - grp_ptr[gm];
                    for (int i0 = 0; i0 < len; i0 += 256) {
                        int ts[4], rs[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) { const int i = i0 + 64 * u + lane; ts[u] = i < len ? l[i] : -1; }
#pragma unroll
                        for (int u = 0; u < 4; u++) rs[u] = ts[u] >= 0 ? (int)rank[ts[u]] : 0x7FFFFFFF;
#pragma unroll
                        for (int u = 0; u < 4; u++)
                            if (rs[u] < nvalid)
                                (void)__hip_atomic_fetch_xor(&bitsA[rs[u] >> 5], 1u << (rs[u] & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    lds_wave_fence();
#pragma unroll
                    for (int j = 0; j < WPL; j++) bitsB[lane + 64 * j] = bitsA[lane + 64 * j];
                    lds_wave_fence();
                }
                // lists and member rows a step ahead (global loads; their latency hides under the previous step)
                int preA = -1, preB = -1;                                      // toggle entries of the coming step
                if (nA > 1) preA = tog64[(size_t)(gm + 1) * 64 + lane];        // A, it = 1: gm -> gm + 1
                if (nB > 0) preB = tog64[(size_t)gm * 64 + lane];              // B, it = 0: gm -> gm - 1
                auto member_of = [&](int g, bool valid) {                      // the element lane l5 ranks in row g (-1: none)
                    // (the slot table: one load, no pointer chase through mem_ptr — that one made every step wait for all its loads)
                    const uint32_t w = (uint32_t)mem_t32[(size_t)(l5 >> 1) * G + (valid ? g : 0)];
                    const int t = (int)((w >> (16 * (l5 & 1))) & 0xFFFFu);
                    return (valid && t != 0xFFFF) ? t : -1;
                };
                int memN = member_of(half ? gm - 1 : gm, half ? nB > 0 : true);
                for (int it = 0; it < nA; it++) {
                    const int gA = gm + it, gB = gm - 1 - it;
                    const bool vB = it < nB;
                    // toggles of both streams, all rank reads first, then the XORs
                    const int tA = it > 0 ? preA : -1, tB = vB ? preB : -1;
                    const int rA = tA >= 0 ? (int)rank[tA] : 0x7FFFFFFF, rB = tB >= 0 ? (int)rank[tB] : 0x7FFFFFFF;
                    if (rA < nvalid) (void)__hip_atomic_fetch_xor(&bitsA[rA >> 5], 1u << (rA & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (rB < nvalid) (void)__hip_atomic_fetch_xor(&bitsB[rB >> 5], 1u << (rB & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (it + 1 < nA) preA = tog64[(size_t)(gA + 1) * 64 + lane];
                    if (it + 1 < nB) preB = tog64[(size_t)gB * 64 + lane];
                    // (an explicit copy: left to itself the compiler loads the next element into a fresh register and moves it over at
                    //  the loop's end — behind a wait for every load in flight, which undoes the prefetch of all three lists)
                    int t;
                    __asm__ volatile("v_mov_b32 %0, %1" : "=v"(t) : "v"(memN));
                    memN = member_of(half ? gB - 1 : gA + 1, half ? it + 1 < nB : it + 1 < nA);
                    lds_wave_fence();
                    // member counts of both vectors, packed (A low, B high 16 bits): one prefix scan; the per-word prefix tables
                    uint32_t wa[WPL], wb[WPL];
                    int clp = 0;
#pragma unroll
                    for (int j = 0; j < WPL; j++) {
                        wa[j] = bitsA[lane * WPL + j]; wb[j] = bitsB[lane * WPL + j];
                        clp += __popc(wa[j]) + (__popc(wb[j]) << 16);
                    }
                    const int inclp = wave_incl_scan(clp);
                    const int totp = __builtin_amdgcn_readlane(inclp, 63);
                    const int cntA = totp & 0xFFFF, cntB = (int)((unsigned)totp >> 16);
                    int runp = inclp - clp;
#pragma unroll
                    for (int j = 0; j < WPL; j++) {
                        pw2[lane * WPL + j] = (uint32_t)runp;
                        runp += __popc(wa[j]) + (__popc(wb[j]) << 16);
                    }
                    // first / last member of each row (the extreme tie counts): lanes 30 / 31 of a half read the words of the first /
                    // last lane of their stream that holds any member (wave-uniform lane numbers off the ballots) back from the
                    // LDS and scan them — every lane runs the scan, on broadcast reads, instead of every lane scanning its own
                    // words of both vectors for the sake of two of them (a quarter of the step's vector instructions)
                    const unsigned long long bmA = __ballot((clp & 0xFFFF) != 0), bmB = __ballot(((unsigned)clp >> 16) != 0u);
                    const int LfA = bmA ? __ffsll((long long)bmA) - 1 : 0, LlA = bmA ? 63 - __clzll((long long)bmA) : 0;
                    const int LfB = bmB ? __ffsll((long long)bmB) - 1 : 0, LlB = bmB ? 63 - __clzll((long long)bmB) : 0;
                    const int Lx = l5 == 31 ? (half ? LlB : LlA) : (half ? LfB : LfA);
                    uint32_t wx[WPL];
#pragma unroll
                    for (int j = 0; j < WPL; j++) wx[j] = bits_h[Lx * WPL + j];
                    int pf = 0, pl = 0;
#pragma unroll
                    for (int j = 0; j < WPL; j++)
                        if (wx[j]) pl = j * 32 + 31 - __clz((int)wx[j]);
#pragma unroll
                    for (int j = WPL - 1; j >= 0; j--)
                        if (wx[j]) pf = j * 32 + __ffs((int)wx[j]) - 1;
                    const int px = Lx * (WPL * 32) + (l5 == 31 ? pl : pf);
                    lds_wave_fence();
                    // this half's stream: its row's count, its element (lanes 0-29) or the row's first / last member (lanes 30 / 31)
                    const int n = half ? cntB : cntA;
                    const bool vh = half ? vB : true;
                    int r = t >= 0 ? (int)rank[t] : 0x7FFFFFFF;
                    r = l5 >= 30 ? px : r;
                    int bl = 0, bh = 0;
                    if (vh && n > 0 && r < nvalid) {
                        int rs, re;
                        tie_run(r, rs, re);
                        const int ws = rs >> 5, we = re >> 5;
                        bl = (int)((pw2[ws] >> pwsh) & 0xFFFFu) + __popc(bits_h[ws] & ((1u << (rs & 31)) - 1u));
                        bh = re >= nvalid ? n : (int)((pw2[we] >> pwsh) & 0xFFFFu) + __popc(bits_h[we] & ((1u << (re & 31)) - 1u));
                        if (l5 < 30) r2buf[t] = (uint16_t)(bl + bh + 1);
                    }
                    const int nminA = __builtin_amdgcn_readlane(bh, 30), nmaxA = cntA - __builtin_amdgcn_readlane(bl, 31);
                    const int nminB = __builtin_amdgcn_readlane(bh, 62), nmaxB = cntB - __builtin_amdgcn_readlane(bl, 63);
                    if (lane == 0 && cntA > 0) put_stats(gA, cntA, nminA, nmaxA);
                    if (lane == 32 && vB && cntB > 0) put_stats(gB, cntB, nminB, nmaxB);
                    lds_wave_fence();  // the next toggles rewrite the vectors / tables
                }
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
                else if (use_tog & 1) {
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
                int run = incl - cl;
#pragma unroll
                for (int j = 0; j < WPL; j++) {
                    pw[lane * WPL + j] = (uint16_t)run;
                    run += __popc(wv[j]);
                }
                lds_wave_fence();
                // members of the row with a global rank below r
                auto below = [&](int r) -> int {
                    if (r >= nvalid) return n;
                    const int w = r >> 5;
                    return (int)pw[w] + __popc(bits[w] & ((1u << (r & 31)) - 1u));
                };
                // first / last member of the row: the words of the first / last lane that holds any, read into scalars
                int p0 = -1, p1 = -1;
                if (slice == 0) {
                    const unsigned long long bm = __ballot(cl > 0);
                    const int Lf = __ffsll((long long)bm) - 1, Ll = 63 - __clzll((long long)bm);
#pragma unroll
                    for (int j = WPL - 1; j >= 0; j--) {
                        const uint32_t wf = (uint32_t)__builtin_amdgcn_readlane((int)wv[j], Lf);
                        if (wf) p0 = (Lf * WPL + j) * 32 + __ffs((int)wf) - 1;
                    }
#pragma unroll
                    for (int j = 0; j < WPL; j++) {
                        const uint32_t wl = (uint32_t)__builtin_amdgcn_readlane((int)wv[j], Ll);
                        if (wl) p1 = (Ll * WPL + j) * 32 + 31 - __clz((int)wl);
                    }
                }
                const int *ml = mem_idx + mem_ptr[g];
                const int mlen = mem_ptr[g + 1] - mem_ptr[g];
                if (nslice == 1 && mlen <= 62) {
                    // the usual case (a wave per group, at most 62 members): the extreme tie counts of the row ride in lanes 62 / 63
                    // of the members' own pass — the tie run of the first member gives nmin, the one of the last member nmax — instead
                    // of a chain of dependent LDS reads of their own that the whole wave waits on
                    const int t = lane < mlen ? ml[lane] : -1;
                    int r = t >= 0 ? (int)rank[t] : 0x7FFFFFFF;
                    r = lane == 62 ? p0 : (lane == 63 ? p1 : r);
                    int bl = 0, bh = 0;
                    if (r < nvalid) {
                        int rs, re;
                        tie_run(r, rs, re);
                        bl = below(rs); bh = below(re);
                        if (lane < 62) r2buf[t] = (uint16_t)(bl + bh + 1);
                    }
                    const int nmin = __builtin_amdgcn_readlane(bh, 62), nmax = n - __builtin_amdgcn_readlane(bl, 63);
                    if (lane == 0) put_stats(g, n, nmin, nmax);
                } else {
                    if (slice == 0) {  // extreme tie counts of the row (wave-uniform work)
                        int e0 = p0 + 1, s1 = p1;
                        if (have_tb) {
                            if (e0 < nvalid && tbit(e0)) e0 = run_end(p0);
                            if (s1 > 0 && tbit(s1)) s1 = run_begin(p1);
                        } else {
                            if (e0 < nvalid && ser.eq_ranks(e0, p0)) e0 = run_end(p0);
                            if (s1 > 0 && ser.eq_ranks(s1 - 1, p1)) s1 = run_begin(p1);
                        }
                        const int nmin = below(e0), nmax = n - below(s1);
                        if (lane == 0) put_stats(g, n, nmin, nmax);
                    }
                    // average-tie ranks of the elements this row ranks (the exact group)
                    // (four rounds of member indexes are requested before the first is used: the list is a gather from global
                    //  memory, and one round trip per 64 members was most of a long row's walk — monthly groups: 912 members)
                    for (int i0 = slice * 64; i0 < mlen; i0 += nslice * 64 * 4) {
                        int ts[4], rr[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) { const int i = i0 + u * nslice * 64 + lane; ts[u] = i < mlen ? ml[i] : -1; }
#pragma unroll
                        for (int u = 0; u < 4; u++) rr[u] = ts[u] >= 0 ? (int)rank[ts[u]] : 0x7FFFFFFF;
#pragma unroll
                        for (int u = 0; u < 4; u++)
                            if (rr[u] < nvalid) {
                                int rs, re;
                                tie_run(rr[u], rs, re);
                                r2buf[ts[u]] = (uint16_t)(below(rs) + below(re) + 1);
                            }
                    }
                }
                lds_wave_fence();  // the next toggles rewrite bits / pw
            }
        }
        __syncthreads();
        if constexpr (XS_QDM_STOP == 3) { if (r2buf[5] == 0xFFFF && tid == 0) flag_count[1] = 1; __syncthreads(); continue; }
        for (int g = tid; g < G; g += NT) {
            const int *gi = reinterpret_cast<const int *>(gstat + 3 * g);
            const double N = (double)gi[0], nmin = (double)gi[2], nmax = (double)gi[4];
            gstat[3 * g] = N;
            gstat[3 * g + 1] = ((nmin + 1.0) / 2.0) / N;
            gstat[3 * g + 2] = ((N - nmax) + (nmax + 1.0) / 2.0) / N;
        }
        __syncthreads();

        // ---------------- D. percent ranks, interpolation on the quantile axis, apply (coalesced) ----------------
        bool flag_cell = false;
        const int tab_n = tmeta[0];
        const R tab_yf = tyfl[0], tab_yl = tyfl[1];
        const bool bad_nodes = af && G > 1 && tmeta[3];
        const R *afc = af ? af + (size_t)c * G * Q : nullptr;
        // Grouped tables, no sim_q wanted: the node a percent rank maps to depends on (2r, N, pmin, pmax) only, and most groups
        // share the middle group's triple (full windows, untied extremes).  Its map 2r -> node is tabulated once per cell with
        // the very expressions evaluated per element below (two float64 divisions and a bisection each); an element of such a
        // group costs one table byte.  The table lives in the unused tail of gstat (groups beyond G).
        unsigned char *tabD = reinterpret_cast<unsigned char *>(gstat + 3 * G);
        const double N0 = gstat[3 * (G / 2)], pmin0 = gstat[3 * (G / 2) + 1], pmax0 = gstat[3 * (G / 2) + 2];
        const int n0 = (int)N0;
        const int tab_len = (2 * n0 + 2 + 15) & ~15;
        const bool use_tab = af && G > 1 && !sim_q && !bad_nodes && n0 > 0 && (size_t)(tab_len + G) <= (size_t)(QC::GMAX - G) * 24 && Q <= 63;
        unsigned char *gsame = tabD + tab_len;  // per group: its (N, pmin, pmax) is the middle group's — its elements take the table
        auto node_code = [&](double sq) -> int {  // 255: NaN; else node | 0x40 below the first node | 0x80 above the last
            if (sq != sq) return 255;
            if (sq < (double)qn[0]) return 0x40;
            if (sq > (double)qn[Q - 1]) return (Q - 1) | 0x80;
            int w;
            (void)row_nearest<R, double>(sq, qn, Q, &w);
            return w;
        };
        if (use_tab) {
            for (int r2 = tid; r2 < 2 * n0 + 2; r2 += NT) {
                const double r = (double)r2 / 2.0;
                const double pr = r / N0;
                tabD[r2] = (unsigned char)(r2 ? node_code(pmax0 * (pr - pmin0) / (pmax0 - pmin0)) : 255);
            }
            for (int g = tid; g < G; g += NT) gsame[g] = (unsigned char)(gstat[3 * g] == N0 && gstat[3 * g + 1] == pmin0 && gstat[3 * g + 2] == pmax0);
            __syncthreads();
        }
        if (af && use_tab) {
            // NB elements per thread and round, every load of a stage in flight together: element by element the pass is a chain of
            // dependent round trips (group index -> group triple -> table byte -> factor -> store) and took 4.7 of the kernel's 18.5 ms;
            // four at a time 4.2 ms; all eleven at once spill (19.3 ms for the kernel)
            constexpr int NB = 8;
            for (int t0 = tid; t0 < (int)T; t0 += NB * NT) {
                int gg[NB], rr[NB], code[NB];
                bool in[NB];
                R vv[NB], aa[NB];
#pragma unroll
                for (int u = 0; u < NB; u++) {
                    const int t = t0 + u * NT;
                    in[u] = t < (int)T;
                    const int tc = in[u] ? t : 0;
                    gg[u] = gidx ? gidx[tc] : 0;
                    rr[u] = r2buf[tc];
                    vv[u] = ser.at_time(tc);
                }
#pragma unroll
                for (int u = 0; u < NB; u++) {
                    const bool same = gsame[max(gg[u], 0)] != 0;
                    const int cd = tabD[rr[u]];
                    code[u] = (rr[u] && gg[u] >= 0 && same) ? cd : -1;
                }
#pragma unroll
                for (int u = 0; u < NB; u++)
                    if (in[u] && code[u] < 0) {  // (rare: a group whose triple differs, an element without a rank or a group)
                        double sq = __builtin_nan("");
                        const int g = gg[u];
                        if (rr[u] && g >= 0) {
                            const double N = gstat[3 * g], pmin = gstat[3 * g + 1], pmax = gstat[3 * g + 2];
                            const double r = (double)rr[u] / 2.0;
                            const double pr = r / N;
                            sq = pmax * (pr - pmin) / (pmax - pmin);
                        }
                        code[u] = g >= 0 ? node_code(sq) : 255;
                    }
#pragma unroll
                for (int u = 0; u < NB; u++) {
                    const bool have = in[u] && code[u] != 255;
                    aa[u] = have ? afc[(size_t)gg[u] * Q + (code[u] & 0x3F)] : nan_of<R>();
                }
#pragma unroll
                for (int u = 0; u < NB; u++) {
                    R a = aa[u];
                    if (in[u] && code[u] != 255) {
                        flag_cell |= a != a;
                        if ((code[u] & 0xC0) && extrap != 0) a = nan_of<R>();
                    }
                    // (streaming store: the result must not push this CU's series out of the L2 before pass D has read it again —
                    //  the keys have left the LDS in the two-stream walk)
                    if (in[u]) __builtin_nontemporal_store(apply_corr<R>(vv[u], a, kind), &scen[(size_t)c * T + (t0 + u * NT)]);
                }
            }
        } else  // (sim_q wanted, a single group, or no room for the table: element by element)
        for (int t = tid; t < (int)T; t += NT) {
            const int g = gidx ? gidx[t] : 0;
            const int r2 = r2buf[t];
            const R v = ser.at_time(t);
            double sq = __builtin_nan("");
            if (r2 && g >= 0) {
                const double N = gstat[3 * g], pmin = gstat[3 * g + 1], pmax = gstat[3 * g + 2];
                const double r = (double)r2 / 2.0;
                const double pr = r / N;
                sq = interp == -1 ? r : pmax * (pr - pmin) / (pmax - pmin);  // interp -1 (ranks only): the average ranks themselves
            }
            if (sim_q) sim_q[(size_t)c * T + t] = sq;
            if (af) {
                R a;
                if (G == 1) {
                    int nd_unused;
                    a = interp1d_eval<R>(sq, true, qn, ty, tab_n, tab_yf, tab_yl, interp, extrap, nd_unused);
                } else if (g < 0 || sq != sq) a = nan_of<R>();
                else {
                    const R *row = afc + (size_t)g * Q;
                    if (sq < (double)qn[0]) { a = row[0]; flag_cell |= a != a; if (extrap != 0) a = nan_of<R>(); }
                    else if (sq > (double)qn[Q - 1]) { a = row[Q - 1]; flag_cell |= a != a; if (extrap != 0) a = nan_of<R>(); }
                    else {
                        int w;
                        (void)row_nearest<R, double>(sq, qn, Q, &w);
                        a = row[w];
                        flag_cell |= a != a;
                    }
                }
                scen[(size_t)c * T + t] = apply_corr<R>(v, a, kind);
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
