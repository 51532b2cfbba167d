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

// experiment builds only (-DXS_QDM_CLK=1): wave 0 of block 0 accumulates the shader clock between phase boundaries and prints it
#ifndef XS_QDM_CLK
#define XS_QDM_CLK 0
#endif
#if XS_QDM_CLK
#define XS_CLK(i) do { const long long now_ = clock64(); clk_[i] += now_ - last_; last_ = now_; } while (0)
#else
#define XS_CLK(i) do { } while (0)
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

// The node of the quantile axis a percent rank maps to, as pass D of k_qdm_cell codes it in a byte: 255: NaN; else
// node | 0x40 below the first node | 0x80 above the last.  qn: the Q nodes (LDS).
template <typename R>
__device__ __forceinline__ int qdm_node_code(double sq, const R *qn, int Q)
{
    if (sq != sq) return 255;
    if (sq < (double)qn[0]) return 0x40;
    if (sq > (double)qn[Q - 1]) return (Q - 1) | 0x80;
    int w;
    (void)row_nearest<R, double>(sq, qn, Q, &w);
    return w;
}

// The same byte for the ONE row of a single group (utils._interp_on_quantiles_1D with interp1d 'nearest', as interp1d_eval
// decides it: bounds halfway between the nodes, summed in the table's type, a percent rank ON a bound goes DOWN): complete rows
// only (no NaN among the nodes or the cell's factors — the compacted row of pass D is then the row itself).
template <typename R>
__device__ __forceinline__ int qdm_node_code_1d(double sq, const R *qn, int Q)
{
    if (sq != sq) return 255;
    if (sq < (double)qn[0]) return 0x40;
    if (sq > (double)qn[Q - 1]) return (Q - 1) | 0x80;
    int lo = 0, hi = Q - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const R b = qn[mid] / (R)2.0 + qn[mid + 1] / (R)2.0;
        if ((double)b < sq) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// The map 2 * rank -> node code of a group depends on the group's (N, ties at the minimum, ties at the maximum) only — not on
// the cell.  For the triples (N, 1, 1) — a ranking row of N valid values with untied extremes: every row of a cell without NaNs
// and without ties at its extremes — it is tabulated ONCE per launch, one table per distinct row length of the plan, with the
// very expressions pass D evaluates per element (utils.rank :631-635 rescaling, two float64 divisions, a bisection).  A cell
// whose groups all match reads these tables and builds none of its own (k_qdm_cell spent 1.9 K such evaluations per cell on
// the middle group's table and ~150 instructions per element of every group with another triple: 5 of 12 months).
// one_d: the row of a single group (1-D nearest: qdm_node_code_1d) instead of the rows of a grouped table.
// grid: (blocks, classes).
template <typename R>
__global__ __launch_bounds__(256) void k_qdm_node_tables(const R *__restrict__ quantiles, int Q, const int *__restrict__ cls_N,
                                                         const int *__restrict__ cls_off, unsigned char *__restrict__ out, int one_d)
{
    __shared__ R qn[64];
    if ((int)threadIdx.x < Q) qn[threadIdx.x] = quantiles[threadIdx.x];
    __syncthreads();
    const int n = cls_N[blockIdx.y];
    unsigned char *tab = out + cls_off[blockIdx.y];
    const double N0 = (double)n, nmin = 1.0, nmax = 1.0;
    const double pmin0 = ((nmin + 1.0) / 2.0) / N0, pmax0 = ((N0 - nmax) + (nmax + 1.0) / 2.0) / N0;  // (as k_qdm_cell parks them)
    for (int r2 = blockIdx.x * blockDim.x + threadIdx.x; r2 < 2 * n + 2; r2 += gridDim.x * blockDim.x) {
        const double r = (double)r2 / 2.0;
        const double pr = r / N0;
        const double sq = pmax0 * (pr - pmin0) / (pmax0 - pmin0);
        tab[r2] = (unsigned char)(r2 ? (one_d ? qdm_node_code_1d<R>(sq, qn, Q) : qdm_node_code<R>(sq, qn, Q)) : 255);
    }
}

// k_qdm_node_tables' classes as the cell kernel takes them (by value: scalar registers, no loads in the kernel): a group of N valid
// values with untied extremes takes the table of the class with that N — whatever its row length is (a shorter month's table
// serves a longer month that lost values to NaNs)
struct QdmTabs {
    int K = 0, bytes = 0;  // classes (0: no tables), bytes of all tables
    int N[16] = {}, off[16] = {};  // the class's count of values, where its table starts in gtab (in units of 16 bytes)
};

// PASS: the ranking rows are the groups themselves (one group holding every step, or bit 1 of use_tog) — the kernel is instantiated
// without the walks, which is worth 60 spilled registers to either half
template <int EPT, typename R = float, bool PASS = false>
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
    const int *__restrict__ mem_t32 /* [16][G] member slots, two 16-bit time steps per word (0xFFFF: none), or null */,
    const unsigned char *__restrict__ gtab /* k_qdm_node_tables' tables (never null: any readable address when there are none) */,
    const QdmTabs gt, const unsigned char *__restrict__ gidx8 /* gidx in bytes (255: no group) when G <= 254, else null */)
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

    const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // (wave-uniform: what derives from it lives in scalar registers)
    uint32_t *bits = reinterpret_cast<uint32_t *>(smem + QC::off_bits) + wid * BITW;
    uint16_t *pw = reinterpret_cast<uint16_t *>(smem + QC::off_pw) + wid * BITW;

#if XS_QDM_CLK
    long long clk_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = clock64();
#endif
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        XS_CLK(7);
        int lane = tid & 63;  // opaque per cell: lane-derived addresses are recomputed, not spilled across the sort
        __asm__ volatile("" : "+v"(lane));
        int nvalid;
        const uint16_t *sorted;
        if constexpr (sizeof(R) == 8) sorted = cell_load_sort_f64<EPT>(sim + c * sc, T, smem, red, nvalid);
        else {
            sorted = cell_load_sort<EPT>(sim + c * sc, T, smem, red, 0, nvalid, NoLoadHook(), -1, true);
            if (!sorted) sorted = idx1;  // (no valid value: not sorted; nothing below reads the order when nvalid == 0)
        }
        XS_CLK(0);
        if constexpr (XS_QDM_STOP == 1) { if (sorted[0] == 0xFFFF && tid == 0) flag_count[1] = 1; __syncthreads(); continue; }
        uint16_t *r2buf = (sorted == idx1) ? idx0 : idx1;  // 2 * average rank per time step, 0 = no rank
        // PASS over several groups: the groups of the elements at this thread's ranks (rank (wid * RND + j) * 64 + lane), requested
        // here, cross the tie bits / tables phase in flight.  The gather goes through the BYTE copy of gidx: 11 KB stay in the CU's
        // vector cache, whereas 11 264 scattered reads of the 44 KB of ints pulled a 64-byte sector each from L2 (1.05 of the pass's
        // 2.8 ms per 64 800 cells).  The order beyond nvalid holds the NaN steps: read, not used.
        constexpr int RND = TPAD / NT;
        int gq[PASS ? RND : 1];
        if constexpr (PASS) {
            if (use_tog & 2) {
#pragma unroll
                for (int j = 0; j < RND; j++) gq[j] = (int)gidx8[min((int)sorted[(((wid * RND) << 6) | lane) + 64 * j], (int)T - 1)];
            }
        }
        // two group streams per wave (see C2 below) when the rows walk by the padded toggle table, every wave has groups of its own
        // and a row ranks at most 30 elements (30 years of a day-of-year window): the second stream's vectors take the place of the
        // keys, which are dead once the tie bits are formed — what still needs a value reads the series itself
        const bool have_tb = (size_t)(TPAD / 8) <= (size_t)(QC::GMAX - G) * 24;
        const bool two = !PASS && have_tb && G >= NWAVE && (use_tog & 1) && tog64 != nullptr && max_mem <= 30 && mem_t32 != nullptr;
        const QdmSeries<R> ser{sim + c * sc, keys, sorted, two}, ser_lds{sim + c * sc, keys, sorted, false};
        // tie bits: bit r set when the values at ranks r - 1 and r are equal — formed once per cell, in parallel, in the unused
        // tail of gstat (shared with pass D's node table, which is built later): a tie test in the group walk is then one LDS
        // read instead of two dependent index -> key round trips per side (the walk waits on LDS latency, not on issue)
        uint32_t *tieb = reinterpret_cast<uint32_t *>(gstat + 3 * G);
        auto tbit = [&](int r) { return (tieb[r >> 5] >> (r & 31)) & 1u; };
        // ... and, per word of tie bits, where the run that reaches into the word from below starts (WS) and where the run that
        // leaves it at the top ends (WE): the run of any rank is then two bit scans in its own word, or one of the two entries —
        // no search.  (Round 5: the runs were found by BISECTION over index -> value reads, 14 dependent steps — global loads once
        // the keys have left the LDS in the two-stream walk — in whichever lanes met a tie: measured data are full of ties (values
        // rounded to 0.01: the day-of-year rank_window kernel took 1.8 x its time on continuous values, zero-inflated: 2.2 x).)
        const bool have_runs = !PASS && have_tb && (size_t)(TPAD / 8) + (size_t)(TPAD / 32) * 4 + 16 <= (size_t)(QC::GMAX - G) * 24;
        uint16_t *WS = reinterpret_cast<uint16_t *>(tieb + TPAD / 32 + 2), *WE = WS + TPAD / 32;
        // the run of equal values around rank r (the order is non-decreasing, so equality with r is monotone on each side):
        // first rank in [0, r] whose value equals the one at r ...
        auto run_begin = [&](int r) {
            if (have_runs) {
                const int w = r >> 5;
                const uint32_t zl = ~tieb[w] & (0xFFFFFFFFu >> (31 - (r & 31)));  // run starts at or below r in its word
                const int ws = (int)WS[w];
                return zl ? (w << 5) + 31 - __clz((int)zl) : ws;
            }
            int lo = 0, hi = r;
            while (lo < hi) { int mid = (lo + hi) >> 1; if (ser.eq_ranks(mid, r)) hi = mid; else lo = mid + 1; }
            return lo;
        };
        // ... and first rank in (r, nvalid] whose value differs (nvalid if none)
        auto run_end = [&](int r) {
            if (have_runs) {
                const int w = r >> 5;
                const uint32_t zh = ~tieb[w] & (0xFFFFFFFEu << (r & 31));  // run starts above r in its word (the bits past nvalid are clear)
                const int we = (int)WE[w];
                return zh ? (w << 5) + __ffs((int)zh) - 1 : we;
            }
            int lo = r + 1, hi = nvalid;
            while (lo < hi) { int mid = (lo + hi) >> 1; if (ser.eq_ranks(mid, r)) lo = mid + 1; else hi = mid; }
            return lo;
        };
        auto tie_run = [&](int r, int &rs, int &re) {
            rs = r; re = r + 1;
            if (have_runs) { rs = run_begin(r); re = run_end(r); }
            else if (have_tb) {
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
            // (PASS: the thread index rebuilt from the per-cell opaque lane, so that the RND addresses derived from it are formed
            //  here instead of being kept — spilled — across the whole cell loop)
            const int tidb = (wid << 6) | lane;
            uint32_t eqm = 0u;
#pragma unroll
            for (int j = 0; j < TPAD / NT; j++) {
                const int r = j * NT + tidb;
                eqm |= (uint32_t)(r > 0 && r < nvalid && ser_lds.eq_ranks(r - 1, r)) << j;
            }
#pragma unroll
            for (int j = 0; j < TPAD / NT; j++) {
                const int r = j * NT + tidb;
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
        if (have_runs) {
            // WS / WE from the finished tie words: an exclusive prefix maximum of "the highest run start in the word" (wave 0, the
            // words in ascending order) and an exclusive suffix minimum of "the lowest one" (wave 1, descending: as a maximum of
            // BIG - position), 64 words at a time on the DPP path, the carry in a scalar.  Positions + 1, 0 = none.
            auto wave_max_scan = [&](int v) {  // inclusive, v >= 0
                v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false));
                v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false));
                v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false));
                v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false));
                v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false));
                v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false));
                return v;
            };
            constexpr int NW32 = TPAD / 32, BIG = 1 << 20;
            if (wid == 0) {
                int carry = 0;
                for (int ch = 0; ch < NW32; ch += 64) {
                    const int w = ch + lane;
                    const uint32_t z = w < NW32 ? ~tieb[w] : 0u;
                    const int v = z ? (w << 5) + 31 - __clz((int)z) + 1 : 0;
                    const int incl = wave_max_scan(v);
                    int ex = __builtin_amdgcn_ds_bpermute(((lane + 63) & 63) * 4, incl);
                    ex = max(lane ? ex : 0, carry);
                    if (w < NW32) WS[w] = (uint16_t)max(ex - 1, 0);
                    carry = max(carry, __builtin_amdgcn_readlane(incl, 63));
                }
            } else if (wid == 1) {
                int carry = 0;
                for (int ch = ((NW32 - 1) / 64) * 64; ch >= 0; ch -= 64) {
                    const int w = ch + 63 - lane;  // (descending with the lane)
                    const uint32_t z = w < NW32 ? ~tieb[w] : 0u;
                    const int v = z ? BIG - ((w << 5) + __ffs((int)z) - 1) : 0;
                    const int incl = wave_max_scan(v);
                    int ex = __builtin_amdgcn_ds_bpermute(((lane + 63) & 63) * 4, incl);
                    ex = max(lane ? ex : 0, carry);
                    if (w < NW32) WE[w] = (uint16_t)(ex ? min(BIG - ex, nvalid) : nvalid);
                    carry = max(carry, __builtin_amdgcn_readlane(incl, 63));
                }
            }
            __syncthreads();
        }

        XS_CLK(1);
        if constexpr (XS_QDM_STOP == 2) { if (r2buf[5] == 0xFFFF && tid == 0) flag_count[1] = 1; __syncthreads(); continue; }
        const bool all_time = G == 1 && grp_ptr[1] - grp_ptr[0] == (int)T && mem_ptr[1] - mem_ptr[0] == (int)T;
        if constexpr (PASS) {
            // ---------------- C0 / C3. the ranking rows are the groups themselves: ranks straight from the sorted order ----------------
            // (one group holding every time step; or a few un-windowed groups — time.month, time.season, ... — with every step in
            //  one: bit 1 of use_tog)
            // C3: ONE more stable counting pass of the sorted order, keyed by the group, lays the elements out group by group in
            // value order — an element's place in its group's segment IS the number of members below it.  No bit-vector is built
            // or walked (the walk built 912-member vectors per wave and ranked by prefix popcounts: 4.8 of 14.2 ms per 64 800
            // monthly cells).  C0: the sorted order is the only segment.
            // Ties: a place is flagged where a segment starts or its value differs from its predecessor's; run start / run end
            // of every place by a forward max-scan / backward min-scan of the flagged places over the block, the segment start
            // the same way.  "Differs": C0 reads the tie bits; C3 scatters, with the time step of an element, the index of its
            // run of equal values in the whole series (its rank minus the tie bits below it) — equal in a segment means equal label.
            // A thread's places are consecutive (its reads of PK are contiguous); every load of a stage is unconditional and in
            // flight together (a branch per element made each a round trip of its own: 4.3 ms for this pass instead of 2.8).
            constexpr int SBW = TPAD / 32 + 2;
            // (LDS: rank[] and what lies behind it — the walks' vectors and prefix tables, the sort's histogram — all dead here)
            static_assert(Cfg::off_rank + (size_t)TPAD * 4 + (size_t)(1024 + SBW + 66) * 4 + (size_t)(TPAD / 32) * 2 <= QC::off_gstat,
                          "elements by place, counters, segment bits, segment starts, tie prefix");
            uint32_t *PK = reinterpret_cast<uint32_t *>(smem + Cfg::off_rank);   // [place] -> time step | run label << 16
            uint32_t *wc = PK + TPAD;                                            // [wave][64] counters -> base places
            uint32_t *SB = wc + 1024;                                            // [SBW] a segment starts at this place
            int *seg = reinterpret_cast<int *>(SB + SBW);                        // [G + 1] first place of every group's segment
            uint16_t *TP = reinterpret_cast<uint16_t *>(seg + 66);               // [TPAD / 32] tie bits in the words before this one
            if (nvalid) {
            const int tq = (wid << 6) | lane;  // (lane is opaque per cell: what derives from it is recomputed here, not kept across the sort)
            const bool direct = !(use_tog & 2);  // (the host sends a single all-time group here: all_time holds)
            wc[tq] = 0u;
            for (int i = tq; i < SBW; i += NT) SB[i] = 0u;
            int tp = 0, tincl = 0;
            if (!direct) {
                tp = tq < TPAD / 32 ? __popc(tieb[tq]) : 0;
                tincl = wave_incl_scan(tp);
                if (lane == 63) red[32 + wid] = (uint32_t)tincl;
            }
            __syncthreads();
            int tt[RND];       // the time step of the element at place p0 + j (a thread owns RND consecutive places)
            uint32_t fl = 0u;  // bit j: place p0 + j is flagged
            const int p0 = tq * RND;
            if (direct) {
                if (tq == 0) {
                    seg[0] = 0; seg[1] = nvalid;
                    atomicOr(&SB[0], 1u);
                    atomicOr(&SB[nvalid >> 5], 1u << (nvalid & 31));
                }
                const unsigned long long tw = (((unsigned long long)tieb[(p0 >> 5) + 1] << 32) | tieb[p0 >> 5]) >> (p0 & 31);
#pragma unroll
                for (int j = 0; j < RND; j++) {
                    const int p = p0 + j;
                    tt[j] = (int)sorted[p];
                    fl |= (uint32_t)(p < nvalid && !((tw >> j) & 1ull)) << j;
                }
                __syncthreads();
            } else {
                const int tbefore = waves_before(red + 32, lane, wid);  // (whole waves: cross-lane reads)
                if (tq < TPAD / 32) TP[tq] = (uint16_t)(tbefore + tincl - tp);
                // (counters private to sub-waves of 16 lanes — fewer lanes on one counter — measured: no gain)
                const int rb = ((wid * RND) << 6) | lane;  // this thread's ranks: rb + 64 j
                uint32_t *wrow = wc + (wid << 6);
                uint32_t od[RND];
                // (every load of a stage unconditional and in flight together — a branch per element made each of them a round trip
                //  of its own)
#pragma unroll
                for (int j = 0; j < RND; j++) tt[j] = (int)sorted[rb + 64 * j];
#pragma unroll
                for (int j = 0; j < RND; j++) gq[j] = rb + 64 * j < nvalid ? gq[j] : -1;
                // (a wave's returning atomics are served in program order, the lanes of one instruction that hit the same word in
                //  ascending lane order — train_cell.hpp: the pass is stable)
#pragma unroll
                for (int j = 0; j < RND; j++) od[j] = gq[j] >= 0 ? atomicAdd(&wrow[gq[j]], 1u) : 0u;
                __syncthreads();
                {   // exclusive scan of the counters in (group, wave) order
                    const bool own = tq < (G << 4);
                    const int cg = tq >> 4, cs = tq & 15;
                    const int ca = (cs << 6) + cg;
                    const int cv = own ? (int)wc[ca] : 0;
                    const int incl = wave_incl_scan(cv);
                    if (lane == 63) red[wid] = (uint32_t)incl;
                    __syncthreads();
                    const int pre = waves_before(red, lane, wid) + incl - cv;
                    if (own) {
                        wc[ca] = (uint32_t)pre;
                        if (cs == 0) { seg[cg] = pre; atomicOr(&SB[pre >> 5], 1u << (pre & 31)); }
                        if (tq == (G << 4) - 1) { seg[G] = pre + cv; atomicOr(&SB[(pre + cv) >> 5], 1u << ((pre + cv) & 31)); }
                    }
                }
                __syncthreads();
                {
                    uint32_t bs[RND], twd[RND], tpv[RND];
#pragma unroll
                    for (int j = 0; j < RND; j++) {
                        const int r = rb + 64 * j;
                        bs[j] = wrow[max(gq[j], 0)];
                        twd[j] = tieb[r >> 5];
                        tpv[j] = TP[r >> 5];
                    }
#pragma unroll
                    for (int j = 0; j < RND; j++) {
                        const int r = rb + 64 * j;
                        const int ties = (int)tpv[j] + __popc(twd[j] & (0xFFFFFFFFu >> (31 - (r & 31))));
                        if (gq[j] >= 0) PK[bs[j] + od[j]] = (uint32_t)tt[j] | ((uint32_t)(r - ties) << 16);
                    }
                }
                __syncthreads();
                const int ntot_ = seg[G];
                uint32_t pk[RND + 1];  // (places beyond ntot hold stale words: read, masked by the comparison below)
                pk[0] = PK[max(p0 - 1, 0)];
#pragma unroll
                for (int j = 0; j < RND; j++) pk[j + 1] = PK[p0 + j];
#pragma unroll
                for (int j = 0; j < RND; j++) {
                    tt[j] = (int)(pk[j + 1] & 0xFFFFu);
                    fl |= (uint32_t)(p0 + j < ntot_ && (pk[j + 1] >> 16) != (pk[j] >> 16)) << j;
                }
            }
            const int ntot = seg[G];  // the ranked steps (valid and in a group)
            // bit j: place p0 + j starts a segment (bit RND: the first place of the next thread does); a segment start is flagged,
            // and so is the place behind the last one
            const uint32_t sbm = (uint32_t)(((((unsigned long long)SB[(p0 >> 5) + 1] << 32) | SB[p0 >> 5]) >> (p0 & 31)) & ((2ull << RND) - 1ull));
            fl |= sbm & ((1u << RND) - 1u);
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
            int lastf = 0, lasts = 0;  // (place + 1) of the last flagged place / segment start among this thread's, 0: none
#pragma unroll
            for (int j = 0; j < RND; j++) {
                if ((fl >> j) & 1u) lastf = p0 + j + 1;
                if ((sbm >> j) & 1u) lasts = p0 + j + 1;
            }
            // backward: the first flagged place of this thread's (as BIG - place so that the max machinery serves), mirrored lanes
            constexpr int BIG = 1 << 20;
            int firstf = 0;
#pragma unroll
            for (int j = RND - 1; j >= 0; j--)
                if ((fl >> j) & 1u) firstf = BIG - (p0 + j);
            int carf, cars, nextf;
            {
                const int inf_ = wave_max_scan(lastf), ins_ = wave_max_scan(lasts);
                int exf = __builtin_amdgcn_ds_bpermute(((lane + 63) & 63) * 4, inf_), exs = __builtin_amdgcn_ds_bpermute(((lane + 63) & 63) * 4, ins_);
                exf = lane ? exf : 0; exs = lane ? exs : 0;
                const int mir = __builtin_amdgcn_ds_bpermute((63 - lane) * 4, firstf);
                const int inclm = wave_max_scan(mir);                       // lane l: max over mirrored lanes <= l = original lanes >= 63 - l
                int exm = __builtin_amdgcn_ds_bpermute(((lane + 63) & 63) * 4, inclm);
                exm = lane ? exm : 0;                                       // mirrored lane l: original lanes > 63 - l
                const int exb = __builtin_amdgcn_ds_bpermute((63 - lane) * 4, exm);
                if (lane == 63) { red[wid] = (uint32_t)inf_; red[16 + wid] = (uint32_t)ins_; red[32 + wid] = (uint32_t)inclm; }
                __syncthreads();
                int cf = 0, cs_ = 0, cb = 0;
#pragma unroll
                for (int w = 0; w < NWAVE; w++) {
                    cf = w < wid ? max(cf, (int)red[w]) : cf;
                    cs_ = w < wid ? max(cs_, (int)red[16 + w]) : cs_;
                    cb = w > wid ? max(cb, (int)red[32 + w]) : cb;
                }
                carf = max(exf, cf); cars = max(exs, cs_);
                const int best = max(exb, cb);
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
                        r2buf[tt[j]] = (uint16_t)((rs[j] - sg[j]) + (re - sg[j]) + 1);
                        const bool first = (sbm >> j) & 1u, lastp = (sbm >> (j + 1)) & 1u;
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
            }  // (nvalid)
        } else if (all_time) {
            // ---------------- C0 by bisection (not reached from the host, which takes the pass for such a group) ----------------
            for (int r = tid; r < nvalid; r += NT) {
                int rs, re;
                tie_run(r, rs, re);
                r2buf[sorted[r]] = (uint16_t)(rs + re + 1);
            }
            if (tid == 0 && nvalid) {
                put_stats(0, nvalid, run_end(0), nvalid - run_begin(nvalid - 1));
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
        XS_CLK(2);
        if constexpr (XS_QDM_STOP == 3) { if (r2buf[5] == 0xFFFF && tid == 0) flag_count[1] = 1; __syncthreads(); continue; }
        // Grouped tables, no sim_q wanted: the node a percent rank maps to depends on (2r, N, pmin, pmax) only.  A group of N valid
        // values with untied extremes takes the table k_qdm_node_tables made for its row length (mode 2: nothing is built here);
        // of the others, those that share the middle group's triple take a table built for this cell (mode 1) with the very
        // expressions evaluated per element below (two float64 divisions and a bisection each); the rest go element by element
        // (mode 0).  Modes, table offsets and the cell's table live in the unused tail of gstat (groups beyond G).
        const int tidd = (wid << 6) | lane;  // (the thread index from the per-cell opaque lane: what derives from it is formed here, not kept — spilled — across the cell loop)
        const bool bad_nodes = af && G > 1 && tmeta[3];
        // A single group holding every step (group="time"), interp="nearest", no sim_q wanted: the launch's ONE table (2 * rank ->
        // node of the row of T values with untied extremes, 1-D flavour) is copied where the elements-by-place of the ranks pass
        // lay (dead since the barrier above) — the element loop below spent two float64 divisions and a bisection on every step
        // (4.3 of 9.9 ms per 64 800 cells).  A cell with NaNs (another N), ties at an extreme or NaN factors goes element by element.
        // (Loaded and stored at once: held across the conversion below, the pieces spilled in the instantiation that serves the months.)
        const bool one1 = PASS && af && G == 1 && !sim_q && interp == 0 && gt.K == 1;
        unsigned char *tab1 = smem + Cfg::off_rank;
        if (one1) {
            constexpr int T1V = (2 * TPAD + 2 + 16 * NT - 1) / (16 * NT);
            const int t1_n16 = (2 * gt.N[0] + 2 + 15) >> 4;
            const uint4 *src = reinterpret_cast<const uint4 *>(gtab + (size_t)gt.off[0] * 16);
            uint4 t1v[T1V];
#pragma unroll
            for (int i = 0; i < T1V; i++) t1v[i] = tidd + i * NT < t1_n16 ? src[tidd + i * NT] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int i = 0; i < T1V; i++)
                if (tidd + i * NT < t1_n16) reinterpret_cast<uint4 *>(tab1)[tidd + i * NT] = t1v[i];
        }
        const int Gp = (G + 15) & ~15;
        unsigned char *gmode = reinterpret_cast<unsigned char *>(gstat + 3 * G);   // bits 0-1: 0 / 1 (table in LDS) / 2 (in gtab); bit 2: the launch's
        uint16_t *goff = reinterpret_cast<uint16_t *>(gmode + Gp);                  // where the group's table starts, in units of 16 bytes
        unsigned char *tabL = gmode + 3 * Gp;                                       // (8-byte aligned)
        const size_t room = (size_t)(QC::GMAX - G) * 24;
        const bool tabs = af && G > 1 && !sim_q && !bad_nodes && Q <= 63 && (size_t)(3 * Gp) <= room;
        const R *afc = af ? af + (size_t)c * G * Q : nullptr;
        // The launch's tables are copied into the LDS when they fit (12 monthly groups: 5.6 KB; a day-of-year row: 1.9 KB), and so
        // is the cell's factor table (12 x 20 factors; not 365 x 20): a table byte or a factor fetched from global memory costs a
        // divergent 64-lane gather each, and the two of them in sequence were most of this pass.  Requested here, stored before the
        // barrier that ends the conversion below: their latency lies under it.
        const int cp_bytes = (tabs && gt.K > 0 && (size_t)(3 * Gp) + (size_t)gt.bytes <= room) ? gt.bytes : 0;
        const int af_bytes = (G * Q * (int)sizeof(R) + 15) & ~15;
        const bool af_l = tabs && (size_t)(3 * Gp) + (size_t)cp_bytes + (size_t)af_bytes <= room;
        R *afL = reinterpret_cast<R *>(tabL + cp_bytes);
        const int loc_off = cp_bytes + (af_l ? af_bytes : 0);
        // (room <= 12 KB: at most two 8-byte pieces of the tables and three factors per thread; loaded here, stored behind the
        //  conversion)
        uint2 cpv[2];
        R afv[3];
#pragma unroll
        for (int i = 0; i < 2; i++) cpv[i] = tidd + i * NT < cp_bytes / 8 ? reinterpret_cast<const uint2 *>(gtab)[tidd + i * NT] : make_uint2(0u, 0u);
#pragma unroll
        for (int i = 0; i < 3; i++) afv[i] = (af_l && tidd + i * NT < G * Q) ? afc[tidd + i * NT] : (R)0;
        // the groups of this thread's time steps (tidd + j * NT), for the element pass below: requested here, they cross the conversion
        int ggA[RND];
        if (tabs) {
#pragma unroll
            for (int u = 0; u < RND; u++) ggA[u] = gidx[min(tidd + u * NT, (int)T - 1)];
        }
        int kinds = 0;  // bit 0: a group with values that the launch's tables do not serve, bit 1: one that they do
        for (int g = tidd; g < G; g += NT) {
            const int *gi = reinterpret_cast<const int *>(gstat + 3 * g);
            const int ni = gi[0], nmini = gi[2], nmaxi = gi[4];
            int cls_off = -1;  // the launch's table for (ni, 1, 1), if there is one
            if (tabs && ni > 0 && nmini == 1 && nmaxi == 1)
                for (int k = 0; k < gt.K; k++) cls_off = gt.N[k] == ni ? gt.off[k] : cls_off;
            if (tabs) {
                gmode[g] = cls_off >= 0 ? (cp_bytes ? 5 : 6) : 0;
                goff[g] = (uint16_t)max(cls_off, 0);
                kinds |= ni > 0 ? (cls_off >= 0 ? 2 : 1) : 0;
            }
            if (one1) gmode[0] = (unsigned char)(ni == gt.N[0] && nmini == 1 && nmaxi == 1);
            // (the float64 triple only where something reads it: element by element, or the cell's own table)
            const double N = (double)ni, nmin = (double)nmini, nmax = (double)nmaxi;
            gstat[3 * g] = N;
            if (cls_off < 0) {
                gstat[3 * g + 1] = ((nmin + 1.0) / 2.0) / N;
                gstat[3 * g + 2] = ((N - nmax) + (nmax + 1.0) / 2.0) / N;
            }
        }
#pragma unroll
        for (int i = 0; i < 2; i++)
            if (tidd + i * NT < cp_bytes / 8) reinterpret_cast<uint2 *>(tabL)[tidd + i * NT] = cpv[i];
#pragma unroll
        for (int i = 0; i < 3; i++)
            if (af_l && tidd + i * NT < G * Q) afL[tidd + i * NT] = afv[i];
        kinds = (__syncthreads_or(kinds & 1) ? 1 : 0) | (__syncthreads_or(kinds & 2) ? 2 : 0);  // (the barrier's OR is of predicates, not of bits)
        XS_CLK(3);

        // ---------------- D. percent ranks, interpolation on the quantile axis, apply (coalesced) ----------------
        bool flag_cell = false;
        const int tab_n = tmeta[0];
        const R tab_yf = tyfl[0], tab_yl = tyfl[1];
        const double N0 = gstat[3 * (G / 2)], pmin0 = gstat[3 * (G / 2) + 1], pmax0 = gstat[3 * (G / 2) + 2];
        const int n0 = (int)N0;
        const int tab_len = (2 * n0 + 2 + 15) & ~15;
        const bool use_loc = tabs && (kinds & 1) && n0 > 0 && (size_t)(3 * Gp + loc_off + tab_len) <= room && !(gmode[G / 2] & 4);
        const bool use_tab = tabs && (use_loc || (kinds & 2));
        auto node_code = [&](double sq) -> int { return qdm_node_code<R>(sq, qn, Q); };
        if (use_loc) {
            unsigned char *tabD = tabL + loc_off;
            for (int r2 = tidd; r2 < 2 * n0 + 2; r2 += NT) {
                const double r = (double)r2 / 2.0;
                const double pr = r / N0;
                tabD[r2] = (unsigned char)(r2 ? node_code(pmax0 * (pr - pmin0) / (pmax0 - pmin0)) : 255);
            }
            for (int g = tidd; g < G; g += NT)
                if (!(gmode[g] & 4)) {
                    gmode[g] = (unsigned char)(gstat[3 * g] == N0 && gstat[3 * g + 1] == pmin0 && gstat[3 * g + 2] == pmax0);
                    goff[g] = (uint16_t)(loc_off / 16);
                }
            __syncthreads();
        }
        XS_CLK(4);
        const bool use1 = one1 && gmode[0] == 1 && tab_n == Q;
        if (af && use_tab) {
            // All of a thread's (at most RND) elements at once, stage by stage, every load of a stage in flight together: element by
            // element the pass is a chain of dependent round trips (group index -> mode -> table byte -> factor -> store) that all
            // sixteen waves of the block walk in lock-step behind the barrier above — nobody computes while anybody waits.  The
            // group indexes were requested before the conversion (ggA).
            // Free of branches in the common case (the pass was ~100 instructions per element, 40 of them scalar control flow
            // around guards): a time step beyond the series repeats the last one and is not stored; an element without a rank or a
            // group is code 255 at once; the tables and the factors are read through FLAT pointers (LDS copy or global memory,
            // lane by lane for the tables); the float64 evaluation runs only in a wave that holds an element of a mode-0 group.
            R *const scen_c = scen + (size_t)c * T;
            const R *const ap = af_l ? (const R *)afL : afc;
            const unsigned char *const tl = tabL;
            int rr[RND], code[RND], mb[RND], go[RND];
            R vv[RND], aa[RND];
#pragma unroll
            for (int u = 0; u < RND; u++) {
                const int tc = min(tidd + u * NT, (int)T - 1);
                rr[u] = r2buf[tc];
                vv[u] = ser.at_time(tc);
            }
#pragma unroll
            for (int u = 0; u < RND; u++) {
                const int gc = max(ggA[u], 0);
                mb[u] = gmode[gc] & 3;
                go[u] = (int)goff[gc] * 16 + rr[u];
            }
#pragma unroll
            for (int u = 0; u < RND; u++) code[u] = *((mb[u] == 2 ? gtab : tl) + (mb[u] ? go[u] : 0));
            bool slow = false;
#pragma unroll
            for (int u = 0; u < RND; u++) {
                const bool ranked = rr[u] != 0 && ggA[u] >= 0;
                code[u] = ranked ? (mb[u] ? code[u] : -1) : 255;
                slow |= code[u] < 0;
            }
            if (__any(slow)) {  // (a group whose triple is neither the launch's nor the middle group's)
#pragma unroll
                for (int u = 0; u < RND; u++)
                    if (code[u] < 0) {
                        const int g = ggA[u];
                        const double N = gstat[3 * g], pmin = gstat[3 * g + 1], pmax = gstat[3 * g + 2];
                        const double r = (double)rr[u] / 2.0;
                        const double pr = r / N;
                        code[u] = node_code(pmax * (pr - pmin) / (pmax - pmin));
                    }
            }
#pragma unroll
            for (int u = 0; u < RND; u++) aa[u] = ap[code[u] != 255 ? max(ggA[u], 0) * Q + (code[u] & 0x3F) : 0];
#pragma unroll
            for (int u = 0; u < RND; u++) {
                const bool have = code[u] != 255;
                R a = have ? aa[u] : nan_of<R>();
                flag_cell |= have && a != a;
                if (have && (code[u] & 0xC0) && extrap != 0) a = nan_of<R>();
                // (streaming store: the result must not push this CU's series out of the L2 before pass D has read it again —
                //  the keys have left the LDS in the two-stream walk)
                if (tidd + u * NT < (int)T) __builtin_nontemporal_store(apply_corr<R>(vv[u], a, kind), &scen_c[(unsigned)(tidd + u * NT)]);
            }
        } else if (use1) {
            // (stage by stage, every load of a stage in flight together, as above)
            R *const scen_c = scen + (size_t)c * T;
            int rr[RND], code[RND];
            R vv[RND], aa[RND];
#pragma unroll
            for (int u = 0; u < RND; u++) {
                const int tc = min(tidd + u * NT, (int)T - 1);
                rr[u] = r2buf[tc];
                vv[u] = ser.at_time(tc);
            }
#pragma unroll
            for (int u = 0; u < RND; u++) code[u] = tab1[rr[u]];
#pragma unroll
            for (int u = 0; u < RND; u++) aa[u] = ty[code[u] & 0x3F];
#pragma unroll
            for (int u = 0; u < RND; u++) {
                const bool have = code[u] != 255;
                R a = have ? aa[u] : nan_of<R>();
                if (have && (code[u] & 0xC0) && extrap != 0) a = nan_of<R>();
                if (tidd + u * NT < (int)T) __builtin_nontemporal_store(apply_corr<R>(vv[u], a, kind), &scen_c[(unsigned)(tidd + u * NT)]);
            }
        } else  // (sim_q wanted, a single group, or no room for the table: element by element)
        for (int t = tidd; t < (int)T; t += NT) {
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
        XS_CLK(5);
        const int any = __syncthreads_or((flag_cell || bad_nodes) ? 1 : 0);
        XS_CLK(6);
        if (any && tid == 0) {
            const int slot = atomicAdd(flag_count, 1);
            flag_list[slot] = (int)c;
        }
    }
#if XS_QDM_CLK
    if (blockIdx.x == 0 && tid == 0)
        printf("k_qdm_cell clocks: sort %lld  tiebits %lld  ranks %lld  convert %lld  table %lld  elements %lld  endbar %lld  looptop %lld\n", clk_[0], clk_[1], clk_[2],
               clk_[3], clk_[4], clk_[5], clk_[6], clk_[7]);
#endif
}

}  // namespace xs
