// train_cell4.hpp — the windowed training kernel as TWO 512-thread workgroups per CU (float32, EQM / plain quantiles, the
// padded toggle table; round 5).
//
// k_train_cell3 owns a CU alone (158.6 KB of LDS, 16 waves): its load + sort phases keep the LDS pipe busy and the vector ALU
// idle, its walk the other way round, and every barrier phase idles the whole CU (profiles/r04_phases_*: VALU issue ~7.3 ms +
// LDS-active ~6.6 ms of a 14.5 ms launch, hardly overlapping).  Two independent blocks per CU drift apart by themselves: one
// sorts while the other walks.  For that a block must live on half the LDS (80 KB):
//   * the series goes from HBM into REGISTERS (22 keys per thread) and is sorted as handles — (key bits above 14) << 14 | time
//     step — by the sort of train_cell.hpp on eight waves: region A (45 KB) holds the handles, region B (33.6 KB) the first
//     pass's histogram, then the stable passes' counter rows and rank[] (by time step);
//   * what the walk needs of the sorted order are the VALUES in rank order (one LDS read per order statistic instead of
//     sorted[r] -> keys[t]) — the series is read a second time (coalesced, an L2 hit) and scattered by rank[] over the handles —
//     and rank[] for the toggles, which leaves the LDS: it is copied to a per-block slice of global scratch (22.5 KB,
//     L2-resident) ordered by POSITION — the plan's (base group, occurrence) order of the time steps (train_xlong.hpp:
//     RankGlobal), in which the steps a group transition toggles are contiguous: a list's gather is two or three sectors;
//   * region B then holds the 24 membership vectors of the walk (cell_group_quantiles3g on eight waves: three streams of up to
//     sixteen groups per wave, the row and toggle lists in positions); the hist quantiles of the first pass are read back from
//     the table in global memory by the second (the queue of eight registers of k_train_cell3 would need sixteen).
// Same tables as k_train_cell3, bit for bit (scripts/ubench/train3_bench.hip prints the checksum of both).
// Requirements (host: api.hip): float32, mode 0 / 1, 64 <= T < 11264, 8-byte aligned rows (even cell stride), Q <= 21 (or
// <= 63 with rows below 1024 values), the padded toggle table in positions.
#pragma once
#include "train_cell3.hpp"
#include "train_long.hpp"   // load_l2

namespace xs {

struct Cell4Cfg {
    static constexpr int NT = 512, NWAVE = 8, NL2 = 11, EPT = 2 * NL2;   // 11 pairs = 22 places per thread
    static constexpr int TPAD = NT * EPT;                               // 11264 places of the sort
    static constexpr int WPL = 6;
    static constexpr int NLANE = (TPAD / 32 + WPL - 1) / WPL;           // 59 lanes own words
    static constexpr int VSTRIDE = NLANE * WPL + 4;                     // 358 words per vector (see CellCfg3)
    static constexpr int NAN_RANK = (NLANE * WPL + 2) * 32;
    static constexpr int WROW = 256 + 8;
    static_assert(NAN_RANK + 31 < 65536 && WPL % 2 == 0 && VSTRIDE % 2 == 0, "see CellCfg3");
    static_assert(TPAD <= (1 << 14), "a handle keeps the time step in 14 bits");
    static constexpr size_t off_a = 0;                                  // handles, then values in rank order
    static constexpr size_t off_b = (size_t)TPAD * 4;
    static constexpr size_t b_bytes = (size_t)3 * NWAVE * VSTRIDE * 4;  // the walk's vectors
    static constexpr size_t off_rank = off_b;                           // [TPAD] u16 (the last pass)
    static constexpr size_t off_wcnt = off_b + (size_t)TPAD * 2;        // [NWAVE][WROW] behind rank[]
    static_assert(off_wcnt + (size_t)NWAVE * WROW * 4 <= off_b + b_bytes, "counter rows inside region B");
    static_assert((size_t)(1 << 14) * 2 <= b_bytes, "the first pass's histogram inside region B");
    static constexpr size_t off_misc = (off_b + b_bytes + 15) & ~(size_t)15;
    static constexpr size_t off_lut = off_misc + 64 * 8;
    static constexpr size_t lds_bytes = off_lut + 256 * 4;
    static_assert(2 * lds_bytes <= 160 * 1024, "two blocks per CU");
    static constexpr size_t scratch_bytes = (size_t)TPAD * 2;  // rank[] by position, per block
};

// rank[] in the block's slice of global memory, indexed by position, written by this block: read past the L1
struct RankGlobal4 {
    const uint16_t *p;
    __device__ __forceinline__ int operator[](int pos) const { return (int)load_l2(p + pos); }
};

struct LdsSortedVals {
    using value_type = float;
    const float *v;  // [TPAD] values in rank order
    __device__ __forceinline__ float at_time(int) const { return 0.f; }  // (only the DQM running sums ask for it)
    __device__ __forceinline__ void at_ranks(int r0, int r1, float &v0, float &v1) const { v0 = v[r0]; v1 = v[r1]; }
};

// Load + sort of one series by a 512-thread block.  Leaves the values in rank order in region A, rank[] by position (NAN_RANK
// for steps without a value, and at position T: the tables' padding entry) in `grank`; region B is free.  Returns false (after a
// barrier, nothing written) when the series holds no valid value.  All 512 threads must call it; it ends on a barrier.
// stop (timing aid of scripts/ubench): 1 = return after the load, 2 = after the sort (no values, no copy-out).
__device__ __attribute__((noinline)) bool cell4_load_sort(const float *__restrict__ x, int Ti, unsigned char *smem, uint32_t *red, int &nvalid_out,
                                                const int *__restrict__ pos_of /* [TPAD] */, uint16_t *__restrict__ grank, int stop = 0)
{
    using Cfg = Cell4Cfg;
    constexpr int NT = Cfg::NT, TPAD = Cfg::TPAD, NL = Cfg::NL2, EPT = Cfg::EPT, WROW = Cfg::WROW, NW = Cfg::NWAVE;
    constexpr int D1 = 14, NB1 = 1 << D1;
    int tid = threadIdx.x;
    __asm__ volatile("" : "+v"(tid));  // (see cell_load_sort: nothing derived from it is hoisted out of the cell loop)
    const int lane = tid & 63, wid = tid >> 6;
    uint32_t *H = reinterpret_cast<uint32_t *>(smem + Cfg::off_a);
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem + Cfg::off_b);
    uint16_t *rankl = reinterpret_cast<uint16_t *>(smem + Cfg::off_rank);
    uint32_t *wcnt = reinterpret_cast<uint32_t *>(smem + Cfg::off_wcnt);
    // ---------------- load: every pair of the series in flight, the histogram cleared meanwhile ----------------
    const int Te = Ti & ~1;
    const float xlast = x[Ti - 1];
    float2 v2[NL];
#pragma unroll
    for (int j = 0; j < NL; j++) v2[j] = *reinterpret_cast<const float2 *>(x + min(2 * tid + j * 2 * NT, Te - 2));
    // the handles carry POSITIONS, not time steps (the table is the same for every series: L2 hits beside the series' loads)
    int2 pp[NL];
#pragma unroll
    for (int j = 0; j < NL; j++) pp[j] = *reinterpret_cast<const int2 *>(pos_of + 2 * tid + j * 2 * NT);
    {
        uint4 *hz = reinterpret_cast<uint4 *>(hist);
#pragma unroll
        for (int i = 0; i < NB1 / 8 / NT; i++) hz[tid + i * NT] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
    }
    int nv = 0;
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
    uint32_t k[EPT];
    auto key_of = [&](float f) {  // (cell_load_sort: integer arithmetic, NaN -> 0xFFFFFFFF)
        const uint32_t u = __float_as_uint(f);
        const uint32_t ok = (uint32_t)((int32_t)((u & 0x7FFFFFFFu) - 0x7F800001u) >> 31);
        const uint32_t vk = (u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u)) & ok;
        nv -= (int)ok;
        kmax = max(kmax, vk);
        const uint32_t kk = vk | ~ok;
        kmin = min(kmin, kk);
        return kk;
    };
#pragma unroll
    for (int j = 0; j < NL; j++) {
        const int i = 2 * tid + j * 2 * NT;
        float a = v2[j].x, b = v2[j].y;
        if ((j + 1) * 2 * NT > Te) {  // (wave-uniform: only the last rounds can reach past the series)
            if (i >= Te) { a = (i == Ti - 1) ? xlast : __builtin_nanf(""); b = __builtin_nanf(""); }
        }
        const uint32_t k0 = key_of(a), k1 = key_of(b);
        k[2 * j] = k0; k[2 * j + 1] = k1;
        const uint32_t d0 = k0 & 0x3FFFu, d1 = k1 & 0x3FFFu;
        if (k0 != 0xFFFFFFFFu) atomicAdd(&hist[d0 >> 1], 1u << (16 * (d0 & 1u)));
        if (k1 != 0xFFFFFFFFu) atomicAdd(&hist[d1 >> 1], 1u << (16 * (d1 & 1u)));
    }
#define XS_RED_STEP(ctrl, rmask)                                                                              \
    nv += __builtin_amdgcn_update_dpp(0, nv, ctrl, rmask, 0xf, false);                                        \
    kmin = min(kmin, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)kmin, ctrl, rmask, 0xf, false));          \
    kmax = max(kmax, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)kmax, ctrl, rmask, 0xf, false));
    XS_RED_STEP(0x111, 0xf) XS_RED_STEP(0x112, 0xf) XS_RED_STEP(0x114, 0xf) XS_RED_STEP(0x118, 0xf)
    XS_RED_STEP(0x142, 0xa) XS_RED_STEP(0x143, 0xc)
#undef XS_RED_STEP
    if (lane == 63) { red[3 * wid] = (uint32_t)nv; red[3 * wid + 1] = kmin; red[3 * wid + 2] = kmax; }
    __syncthreads();
    {   // all-reduce over the 8 waves: a lane per wave (period 8 inside every row of 16), rotations inside the row
        static_assert(NW == 8, "rotations by 4, 2, 1");
        nv = (int)red[3 * (lane & 7)]; kmin = red[3 * (lane & 7) + 1]; kmax = red[3 * (lane & 7) + 2];
#define XS_ROW_ROR(x, n) __builtin_amdgcn_update_dpp((int)(x), (int)(x), 0x120 + (n), 0xf, 0xf, false)
        nv += XS_ROW_ROR(nv, 4); kmin = min(kmin, (uint32_t)XS_ROW_ROR(kmin, 4)); kmax = max(kmax, (uint32_t)XS_ROW_ROR(kmax, 4));
        nv += XS_ROW_ROR(nv, 2); kmin = min(kmin, (uint32_t)XS_ROW_ROR(kmin, 2)); kmax = max(kmax, (uint32_t)XS_ROW_ROR(kmax, 2));
        nv += XS_ROW_ROR(nv, 1); kmin = min(kmin, (uint32_t)XS_ROW_ROR(kmin, 1)); kmax = max(kmax, (uint32_t)XS_ROW_ROR(kmax, 1));
#undef XS_ROW_ROR
        nv = __builtin_amdgcn_readfirstlane(nv);
        kmin = (uint32_t)__builtin_amdgcn_readfirstlane((int)kmin);
        kmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)kmax);
    }
    const int nvalid = nv;
    nvalid_out = nv;
    kmin &= ~0x3FFFu;  // the digits counted above are those of (key - kmin) for such a kmin
    const uint32_t range = nvalid ? kmax - kmin : 0u;
    if (stop == 1 || nvalid == 0) { __syncthreads(); return false; }
    const int nbits = 32 - __clz((int)(range + 1u) | 0);
    const int R = nbits > D1 ? nbits - D1 : 0;
    const int npass_rest = (R + 7) >> 3;
    const int db = npass_rest ? (R + npass_rest - 1) / npass_rest : 0;
    const int ND = 1 << db;
    const uint32_t d_nan = (range + 1u) & (NB1 - 1);
    // ---------------- pass 1: scan of the 2^14 packed bins (32 bins = 16 dwords per thread) ----------------
    {
        constexpr int DPT = NB1 / 2 / NT;  // 16
        uint32_t wv[DPT];
        int run = 0;
#pragma unroll
        for (int i = 0; i < DPT; i++) {
            wv[i] = hist[tid * DPT + i];
            if ((int)(d_nan >> 5) == tid && (int)((d_nan & 31u) >> 1) == i) wv[i] += (uint32_t)(TPAD - nvalid) << (16 * (d_nan & 1u));
            run += (int)(wv[i] & 0xFFFFu) + (int)(wv[i] >> 16);
        }
        const int incl = wave_incl_scan(run);
        if (lane == 63) red[64 + wid] = (uint32_t)incl;  // (words 64..: a slow wave may still be reading words 0 .. 23)
        __syncthreads();
        int pre = waves_before(red + 64, lane, wid) + incl - run;
#pragma unroll
        for (int i = 0; i < DPT; i++) {
            const int c0 = (int)(wv[i] & 0xFFFFu), c1 = (int)(wv[i] >> 16);
            hist[tid * DPT + i] = (uint32_t)pre | ((uint32_t)(pre + c0) << 16);
            pre += c0 + c1;
        }
    }
    __syncthreads();
    // ---------------- pass 1: slots by returning atomics ----------------
    {
        const bool last = npass_rest == 0;  // narrow keys: the slot is the rank (parked by position in A)
#pragma unroll
        for (int j = 0; j < EPT; j++) {
            const int t = (j & 1) ? pp[j >> 1].y : pp[j >> 1].x;  // (the position of the time step)
            const uint32_t r = (k[j] == 0xFFFFFFFFu) ? range + 1u : k[j] - kmin;
            const uint32_t d = r & (NB1 - 1);
            const int sh = 16 * (d & 1);
            const uint32_t old = atomicAdd(&hist[d >> 1], 1u << sh);
            const int pos = (int)((old >> sh) & 0xFFFFu);
            if (!last) H[pos] = ((r >> D1) << D1) | (uint32_t)t;
            else H[t] = (uint32_t)pos;
        }
        __syncthreads();
        if (last) {
#pragma unroll
            for (int j = 0; j < NL; j++) {
                const int i = 2 * tid + j * 2 * NT;
                const uint2 ps = *reinterpret_cast<const uint2 *>(H + i);
                const uint32_t r0 = (int)ps.x >= nvalid ? (uint32_t)Cfg::NAN_RANK : ps.x, r1 = (int)ps.y >= nvalid ? (uint32_t)Cfg::NAN_RANK : ps.y;
                *reinterpret_cast<uint32_t *>(rankl + i) = r0 | (r1 << 16);
            }
            __syncthreads();
        }
    }
    // ---------------- the stable passes (train_cell.hpp: wave-private counter rows, lane-ordered returning atomics) ----------------
    uint32_t *wrow = wcnt + wid * WROW;
    if (npass_rest) {
        for (int i = lane; i < ND; i += 64) wrow[i] = 0u;  // (the rows overlay the histogram: cleared behind its last barrier)
        lds_wave_fence();
    }
    for (int p = 0; p < npass_rest; p++) {
        const int shift = D1 + db * p;
        const bool last = p == npass_rest - 1;
        uint32_t hd[EPT], od[EPT];
        {
            const uint32_t *src = H + wid * (64 * EPT) + lane;
#pragma unroll
            for (int j = 0; j < EPT; j++) hd[j] = src[64 * j];
#pragma unroll
            for (int j = 0; j < EPT; j++) od[j] = atomicAdd(&wrow[(hd[j] >> shift) & (uint32_t)(ND - 1)], 1u);
        }
        __syncthreads();
        {   // scan over the counters in digit-major order (digit d of wave w at d 8 + w): thread t owns `per` consecutive ones
            const int NDW = ND << 3, per = NDW > NT ? NDW >> 9 : 1;  // (1, 2 or 4)
            uint32_t c[4];
            int run = 0;
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                const int i = tid * per + kk;
                c[kk] = (kk < per && i < NDW) ? wcnt[(i & 7) * WROW + (i >> 3)] : 0u;
                run += (int)c[kk];
            }
            const int incl = wave_incl_scan(run);
            if (lane == 63) red[wid] = (uint32_t)incl;
            __syncthreads();
            int pre = waves_before(red, lane, wid) + incl - run;
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                const int i = tid * per + kk;
                if (kk < per && i < NDW) wcnt[(i & 7) * WROW + (i >> 3)] = (uint32_t)pre;
                pre += (int)c[kk];
            }
        }
        __syncthreads();
        if (!last) {
#pragma unroll
            for (int j = 0; j < EPT; j++) od[j] += wrow[(hd[j] >> shift) & (uint32_t)(ND - 1)];
#pragma unroll
            for (int j = 0; j < EPT; j++) H[od[j]] = hd[j];  // in place: every handle sits in a register
            for (int i = lane; i < ND; i += 64) wrow[i] = 0u;
        } else {
#pragma unroll
            for (int j = 0; j < EPT; j++) od[j] += wrow[(hd[j] >> shift) & (uint32_t)(ND - 1)];
#pragma unroll
            for (int j = 0; j < EPT; j++) rankl[hd[j] & (uint32_t)((1 << D1) - 1)] = (uint16_t)((int)od[j] >= nvalid ? (uint32_t)Cfg::NAN_RANK : od[j]);
        }
        __syncthreads();
    }
    if (stop == 2) return true;
    // ---------------- rank[] (by position, NAN_RANK for steps without a value) to the block's slice of global memory; the values
    // in rank order over the handles: the series is read a second time, with the positions of its steps ----------------
    {
        float *vals = reinterpret_cast<float *>(smem + Cfg::off_a);
#pragma unroll
        for (int j = 0; j < NL; j++) v2[j] = *reinterpret_cast<const float2 *>(x + min(2 * tid + j * 2 * NT, Te - 2));
#pragma unroll
        for (int j = 0; j < NL; j++) pp[j] = *reinterpret_cast<const int2 *>(pos_of + 2 * tid + j * 2 * NT);
        {
            const uint4 *src = reinterpret_cast<const uint4 *>(rankl);
            uint4 *dst = reinterpret_cast<uint4 *>(grank);
            constexpr int NV = TPAD * 2 / 16;
#pragma unroll
            for (int i = 0; i < (NV + NT - 1) / NT; i++)
                if (tid + i * NT < NV) dst[tid + i * NT] = src[tid + i * NT];
        }
#pragma unroll
        for (int j = 0; j < NL; j++) {
            const int i = 2 * tid + j * 2 * NT;
            float a = v2[j].x, b = v2[j].y;
            if ((j + 1) * 2 * NT > Te) {
                if (i >= Te) { a = (i == Ti - 1) ? xlast : __builtin_nanf(""); b = __builtin_nanf(""); }
            }
            const int ra = (int)rankl[pp[j].x], rb = (int)rankl[pp[j].y];
            if (ra < TPAD) vals[ra] = a;   // (NAN_RANK lies past the places)
            if (rb < TPAD) vals[rb] = b;
        }
    }
    stores_done();
    __syncthreads();
    return true;
}

// P10 / NR: see cell_group_quantiles3g.  STOP (scripts/ubench/train3_bench.hip only): 1 = load, 2 = + sort, 5 = + values and the
// rank copy-out, 3 = + build / copies / fast-forward, 4 = + the loop without the select.
template <bool P10, int STOP = 0, int NR = 1>
__global__ __launch_bounds__(512, 4) void k_train_cell4(const float *__restrict__ x0, const float *__restrict__ x1, int64_t T, int64_t sc,
                                                         const int *__restrict__ row_ptr, const int *__restrict__ rows_pos, int G,
                                                         const float *__restrict__ q, int Q, int kind, int mode, float *__restrict__ af,
                                                         float *out_hq, int *__restrict__ flag_count, int *__restrict__ flag_list, int64_t C,
                                                         const int *__restrict__ tog64_pos /* [G][64] positions, padded with position T */,
                                                         const int *__restrict__ pos_of /* [TPAD] position of a time step (its own index past T) */,
                                                         unsigned char *__restrict__ gscratch, int *__restrict__ cu_count, int stagger)
{
    using Cfg = Cell4Cfg;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *bits_all = reinterpret_cast<uint32_t *>(smem + Cfg::off_b);
    uint32_t *red = reinterpret_cast<uint32_t *>(smem + Cfg::off_misc);
    uint32_t *lut = reinterpret_cast<uint32_t *>(smem + Cfg::off_lut);
    uint16_t *grank = reinterpret_cast<uint16_t *>(gscratch + (size_t)blockIdx.x * Cfg::scratch_bytes);
    const int tid = threadIdx.x, wid = tid >> 6;
    byte_select_lut(lut, tid);  // (the first barrier of the first load comes before any use)
    if (stagger > 0) {
        // The two blocks of a CU start together and do the same work: left alone they sort at the same time and walk at the same
        // time.  The second block to arrive on a CU (a counter per hardware CU id) waits `stagger` clocks once, so that one
        // block's sort phases (LDS pipe) fall under the other's walk (vector ALU) for the rest of the launch.
        uint32_t hwid, xcc;
        __asm__ volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        __asm__ volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if (tid == 0) red[0] = (uint32_t)atomicAdd(&cu_count[((xcc & 15u) << 7) | ((hwid >> 8) & 127u)], 1);
        __syncthreads();
        const uint32_t slot = red[0];
        __syncthreads();
        if (slot & 1u) {
            const uint64_t t0 = __builtin_amdgcn_s_memtime();
            while (__builtin_amdgcn_s_memtime() - t0 < (uint64_t)stagger) __builtin_amdgcn_s_sleep(32);
        }
    }
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        const int npass_arr = mode == 0 ? 1 : 2;
        bool flag_cell = false;
        for (int arr = 0; arr < npass_arr; arr++) {
            int lane = tid & 63;
            __asm__ volatile("" : "+v"(lane));
            const float *x = ((mode == 0 || arr == 1) ? x0 : x1) + c * sc;  // hist first, then ref (see k_train_cell)
            int nvalid;
            const bool any = cell4_load_sort(x, (int)T, smem, red, nvalid, pos_of, grank, STOP == 1 ? 1 : (STOP == 2 ? 2 : 0));
            if constexpr (STOP == 1 || STOP == 2 || STOP == 5) { if (STOP == 5 && any && grank[0] == 0xFFFE) out_hq[0] = 1.f; __syncthreads(); continue; }
            if (!any) {
                // no valid value in this series (block-uniform): every quantile of the pass is NaN (see k_train_cell3)
                const float nanq = nan_of<float>();
                const size_t base = (size_t)c * G * (size_t)Q;
                for (int i = tid; i < G * Q; i += Cfg::NT) {
                    if (mode == 0 || arr == 0) out_hq[base + i] = nanq;
                    else af[base + i] = get_corr<float>(out_hq[base + i], nanq, kind);
                }
                __syncthreads();
                continue;
            }
            const LdsSortedVals vals{reinterpret_cast<const float *>(smem + Cfg::off_a)};
            const RankGlobal4 rk{grank};
            cell_group_quantiles3g<Cfg, RankGlobal4, P10, false, LdsSortedVals, false, (STOP == 3 || STOP == 4) ? STOP : 0, true, NR>(
                vals, rk, bits_all, nvalid, lane, wid, row_ptr, rows_pos, tog64_pos, G, q, Q, kind, mode, arr, c, af, out_hq, nullptr, flag_cell,
                nullptr, nullptr, (int)T, lut, nullptr);
            __syncthreads();
        }
        const int anyf = __syncthreads_or(flag_cell ? 1 : 0);
        if (anyf && tid == 0 && flag_list) {
            const int slot = atomicAdd(flag_count, 1);
            flag_list[slot] = (int)c;
        }
    }
}

}  // namespace xs
