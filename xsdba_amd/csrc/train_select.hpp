// train_select.hpp — training kernel for UN-WINDOWED groups (group="time", "time.month", "time.season", any grouping with
// window == 1 and a few groups): a one-level bucket pass + selection inside the few buckets that matter, instead of a sort.
//
// The reference sorts every group row to read 2*Q order statistics off it (nbutils.py:113-148: arr.sort(), then the
// neighbours of Q virtual indexes).  Only those order statistics are needed, so the kernel
//   A. brings the cell's series from HBM into REGISTERS (order-preserving keys, 2*EPT2 per thread), min / max;
//   B. counts: hist[group][bin] (packed u16 LDS atomics), bin = (key - kbase) >> shift over the occupied key range (kbase: the key above the minimum)
//      (equal-width bins in key space); the values equal to kmin or kmax are counted apart per group (zero-inflated
//      precipitation, capped humidity: they would be long runs of equal keys inside one bin);
//   C. scans the histogram into absolute start positions (each thread a run of bins, block scan of the thread totals);
//   D. scatters every key to its bin's place in an LDS array (returning LDS atomics on the scanned entries, all requests
//      of a thread in flight together): a counting sort by bin — keys inside a bin stay unordered;
//   E. one thread per (group, quantile): the bin holding the ordinal by bisection on the table (which now holds the bins'
//      END positions), the bin's keys sorted in registers (Batcher's network for up to 8 / 16 / 32 / 64 keys, an
//      all-equal test beyond), the upper neighbour from the same bin or the minimum of the next one; type-7
//      interpolation exactly as device_math.hpp::quantile_sorted; hist_q stays in LDS to form af in the ref pass.
// Exact and bit-identical to the sort-based kernels: any monotone binning gives the same order statistics.  A bin with
// more than CAPBIN keys that are not all equal cannot be resolved here: the cell is appended to flag_list and redone by
// the generic kernel.  shift == 0 (few distinct keys) makes every bin one key: nothing to sort.
// HBM traffic per cell = the series once + the tables once = the algorithmic bytes.
#pragma once
#include "train_cell.hpp"

// timing aid for experiment builds only (-DXS_SEL_STOP=n stops every array pass after phase n: A=1 .. D=4); the release
// library is built without it
#ifndef XS_SEL_STOP
#define XS_SEL_STOP 0
#endif

namespace xs {

struct SelLayout {
    int NT, TP, G, Q, NB, log2nb;
    size_t off_hist, off_skeys, off_pre, off_hq, off_gmass, off_gmu, off_red, bytes;
};

constexpr int SEL_NT = 512, SEL_CAPBIN = 61, SEL_MAXG = 32;

// LDS layout, shared by the host (sizing) and the kernel.  log2tot: log2 of the number of bins over all groups.
__host__ __device__ inline SelLayout sel_layout(int TP, int G, int Q, int log2tot)
{
    SelLayout L;
    L.NT = SEL_NT; L.TP = TP; L.G = G; L.Q = Q;
    int lg = 0;
    while ((1 << lg) < G) lg++;
    L.log2nb = log2tot - lg;
    L.NB = 1 << L.log2nb;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = (o + bytes + 15) & ~(size_t)15; return r; };
    L.off_hist = take((size_t)G * L.NB * 2);
    L.off_skeys = take((size_t)TP * 4 + 256);  // (+256: slices are read in whole groups of four keys, up to 64 past a start)
    L.off_pre = take((size_t)(SEL_NT + 1) * 4);
    L.off_hq = take((size_t)G * Q * 4);
    L.off_gmass = take((size_t)2 * SEL_MAXG * 4);
    L.off_gmu = take((size_t)2 * SEL_MAXG * 4);  // DQM: group means of hist, ref
    L.off_red = take(64 * 4);
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
    // the 8 wave totals: a lane each, scanned on the DPP path (the first half of a row of 16), read back with readlane
    static_assert(SEL_NT / 64 == 8, "the wave totals fill half a DPP row");
    int r = (int)red[lane & 7];
    r += __builtin_amdgcn_update_dpp(0, r, 0x111, 0xf, 0xf, false);
    r += __builtin_amdgcn_update_dpp(0, r, 0x112, 0xf, 0xf, false);
    r += __builtin_amdgcn_update_dpp(0, r, 0x114, 0xf, 0xf, false);
    const int w = __builtin_amdgcn_readfirstlane(wid);
    const int before = __builtin_amdgcn_readlane(r, (w + 7) & 7);
    const int woff = w ? before : 0;
    total = __builtin_amdgcn_readlane(r, 7);
    __syncthreads();
    return woff + incl - v;
}

// Keys number rr and rr + 1 (0-based, ascending) among places [lead, mm) of the N keys at `sla` (16-byte aligned; a slice
// starts anywhere, so it is read from its aligned base: the `lead` places in front of it belong to the previous bin and
// count as 0 — they sort first, which is why rr includes them — and the places from mm on count as 0xFFFFFFFF).
// Batcher's odd-even merge sorting network on registers, then a scan for the two places.  Equal keys are equal values,
// so no tie rule is needed.
template <int N> __device__ __forceinline__ void sel_sorted_pick2(const uint32_t *sla, int lead, int mm, int rr, uint32_t &k0, uint32_t &k1)
{
    uint32_t a[N];
#pragma unroll
    for (int i = 0; i < N / 4; i++) {
        const uint4 v = *reinterpret_cast<const uint4 *>(sla + 4 * i);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int u = 0; u < 4; u++) { const int p = 4 * i + u; a[p] = p < lead ? 0u : (p < mm ? w[u] : 0xFFFFFFFFu); }
    }
#pragma unroll
    for (int p = 1; p < N; p *= 2)
#pragma unroll
        for (int k = p; k >= 1; k /= 2)
#pragma unroll
            for (int j = k % p; j <= N - 1 - k; j += 2 * k)
#pragma unroll
                for (int i = 0; i <= (k - 1 < N - j - k - 1 ? k - 1 : N - j - k - 1); i++)
                    if ((i + j) / (2 * p) == (i + j + k) / (2 * p)) {
                        const uint32_t lo = min(a[i + j], a[i + j + k]), hi = max(a[i + j], a[i + j + k]);
                        a[i + j] = lo;
                        a[i + j + k] = hi;
                    }
    k0 = k1 = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < N; i++) {
        k0 = i == rr ? a[i] : k0;
        k1 = i == rr + 1 ? a[i] : k1;
    }
}

// mode 0: quantiles of x0 -> out_hq.  mode 1 (eqm_train): x1 = hist -> out_hq, x0 = ref -> af = corr(hist_q, ref_q).
// mode 2 (dqm_train, DQM = true; _adjustment.py:150-190): the same on the series normalised by their group means — order
// statistics commute with the monotone normalisation, so they are selected on the raw values and normalised afterwards;
// scaling = corr(mean(hist), mean(ref)).  Cells where that argument fails (non-finite mean, non-positive multiplicative
// mean, inf - inf in the interpolation) are appended to flag_list like the sort-based kernels do.
// LT: log2 of the number of histogram bins over all groups (14: 32 KB of LDS)
template <int EPT2, int LT, bool DQM = false>
__global__ __launch_bounds__(SEL_NT, (EPT2 <= 11 ? 4 : 2)) void k_train_sel(
    const float *__restrict__ x0, const float *__restrict__ x1, int64_t T, int64_t sc, const int *__restrict__ gidx, int G,
    const float *__restrict__ q, int Q, int kind, int mode, float *__restrict__ af, float *__restrict__ out_hq,
    float *__restrict__ scaling, int *__restrict__ flag_count, int *__restrict__ flag_list, int64_t C)
{
    constexpr int NT = SEL_NT, NE = 2 * EPT2, TP = NE * NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const SelLayout L = sel_layout(TP, G, Q, LT);
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem + L.off_hist);
    uint16_t *hist16 = reinterpret_cast<uint16_t *>(smem + L.off_hist);
    uint32_t *skeys = reinterpret_cast<uint32_t *>(smem + L.off_skeys);
    uint32_t *pre = reinterpret_cast<uint32_t *>(smem + L.off_pre);
    float *hqk = reinterpret_cast<float *>(smem + L.off_hq);
    uint32_t *gmass = reinterpret_cast<uint32_t *>(smem + L.off_gmass);  // [g][0: == kmin, 1: == kmax]
    float *gmu = reinterpret_cast<float *>(smem + L.off_gmu);            // DQM: [0: hist, 1: ref][g] group means
    uint32_t *red = reinterpret_cast<uint32_t *>(smem + L.off_red);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int NB = L.NB, log2nb = L.log2nb;
    const int ND = G * NB / 2;               // histogram dwords in use
    constexpr int DPT = (1 << LT) / 2 / NT;  // histogram dwords per scan thread
    const int tpg = NB / (2 * DPT);          // scan threads per group (2 * DPT bins each)
    const int Ti = (int)T;
    constexpr uint32_t MAXK = 0xFFFFFFFFu;

    // the group of each of the thread's elements, 8 bits per element (255: none), once per block: the same for every cell
    uint32_t gpk[(NE + 3) / 4];
    int ung = 0;
#pragma unroll
    for (int e = 0; e < NE; e++) {
        const int i = 2 * tid + (e >> 1) * 2 * NT + (e & 1);
        const int g = i < Ti ? gidx[i] : -1;
        const bool ok = g >= 0 && g < G;
        const uint32_t g8 = ok ? (uint32_t)g : 255u;
        if (e & 3) gpk[e >> 2] |= g8 << (8 * (e & 3)); else gpk[e >> 2] = g8;
        ung |= (i < Ti && !ok) ? 1 : 0;
    }
    const bool ungrouped = __syncthreads_or(ung) != 0;  // some time step belongs to no group

#ifdef XS_SEL_STAGGER
    // (experiment builds of scripts/ubench/sel_bench.hip: the blocks start XS_SEL_STAGGER clocks apart at most, by a hash of their
    //  number — do blocks that stay in step with each other load, count and select at the same time?)
    if (XS_SEL_STAGGER > 0) {
        uint32_t h = (uint32_t)blockIdx.x * 2654435761u;
        h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
        const uint64_t wait = (uint64_t)(h % (uint32_t)XS_SEL_STAGGER), t0 = __builtin_amdgcn_s_memtime();
        while (__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(16);
    }
#endif
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
            uint32_t kmin = MAXK, kmx1 = 0u;  // kmx1 = max(key + 1): a missing value (MAXK) wraps to 0 and never wins
            // key of a value that is kept only when `keep` (three-operation sign fold: ~u for negatives, u | msb otherwise)
            auto mkkey = [](float v, bool keep) {
                const uint32_t u = __float_as_uint(v);
                const uint32_t m = (uint32_t)((int32_t)u >> 31) | 0x80000000u;
                return (v == v && keep) ? (u ^ m) : 0xFFFFFFFFu;
            };
            if ((reinterpret_cast<uintptr_t>(x) & 7) == 0 && (Ti & 1) == 0) {  // 8-byte aligned rows of even length
                float2 v2[EPT2];
#pragma unroll
                for (int j = 0; j < EPT2; j++) {
                    const int i = t2 + j * 2 * NT;
                    v2[j] = *reinterpret_cast<const float2 *>(x + min(i, Ti - 2));
                }
                // the LDS tables are cleared while the loads are in flight
                for (int i = tid; i < ND / 4; i += NT) reinterpret_cast<uint4 *>(hist)[i] = make_uint4(0u, 0u, 0u, 0u);
                if (tid < 2 * G) gmass[tid] = 0u;
#pragma unroll
                for (int j = 0; j < EPT2; j++) {
                    const bool inr = t2 + j * 2 * NT < Ti;  // (pairs never straddle the end: both even)
                    key[2 * j] = mkkey(v2[j].x, inr);
                    key[2 * j + 1] = mkkey(v2[j].y, inr);
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
                    key[e] = mkkey(v[e], i < Ti);
                }
            }
            if (ungrouped) {  // (block-uniform) time steps without a group count as missing
#pragma unroll
                for (int e = 0; e < NE; e++)
                    if (((gpk[e >> 2] >> (8 * (e & 3))) & 255u) == 255u) key[e] = MAXK;
            }
#pragma unroll
            for (int j = 0; j < EPT2; j++) {
                kmin = min(kmin, min(key[2 * j], key[2 * j + 1]));
                kmx1 = max(kmx1, max(key[2 * j] + 1u, key[2 * j + 1] + 1u));
            }
            // wave reduction on the DPP path (lane 63 ends up with the wave's extremes), then a lane per wave and rotations
            // inside the DPP row: no LDS crossbar shuffles, no loop over the waves
#define XS_SEL_RED(ctrl, rmask)                                                                               \
            kmin = min(kmin, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)kmin, ctrl, rmask, 0xf, false)); \
            kmx1 = max(kmx1, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)kmx1, ctrl, rmask, 0xf, false));
            XS_SEL_RED(0x111, 0xf) XS_SEL_RED(0x112, 0xf) XS_SEL_RED(0x114, 0xf) XS_SEL_RED(0x118, 0xf)
            XS_SEL_RED(0x142, 0xa) XS_SEL_RED(0x143, 0xc)
#undef XS_SEL_RED
            if (lane == 63) { red[16 + 2 * wid] = kmin; red[16 + 2 * wid + 1] = kmx1; }
            __syncthreads();  // (also: tables cleared)
            static_assert(NT / 64 == 8, "the per-wave extremes fill half a DPP row");
            kmin = red[16 + 2 * (lane & 7)];
            kmx1 = red[16 + 2 * (lane & 7) + 1];
#define XS_SEL_ROR(n)                                                                                                  \
            kmin = min(kmin, (uint32_t)__builtin_amdgcn_update_dpp((int)kmin, (int)kmin, 0x120 + (n), 0xf, 0xf, false)); \
            kmx1 = max(kmx1, (uint32_t)__builtin_amdgcn_update_dpp((int)kmx1, (int)kmx1, 0x120 + (n), 0xf, 0xf, false));
            XS_SEL_ROR(4) XS_SEL_ROR(2) XS_SEL_ROR(1)
#undef XS_SEL_ROR
            // (uniform values: kept in scalar registers)
            kmin = (uint32_t)__builtin_amdgcn_readfirstlane((int)kmin);
            kmx1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)kmx1);
            const bool any_valid = kmx1 != 0u;
            const uint32_t kmax = kmx1 - 1u;
            // Where the bins start.  A series whose minimum is an exact zero (zero-inflated precipitation: the zeros are counted
            // apart anyway) starts them at its smallest value ABOVE zero: key(0) lies 2^30 keys below key(1e-3), so that a range
            // from zero left ~60 of a group's 1024 bins to all the values between 1 and 100 — runs beyond the in-register sort,
            // and the cell went to the generic kernel (EQM time.month on such data: 5 x the time).  One more block reduction,
            // only for such series (block-uniform).
            uint32_t kbase = kmin + 1u;  // (the first key that is binned: the minimum itself is counted apart)
            if (any_valid && kmin == 0x80000000u) {
                uint32_t k2 = MAXK;
#pragma unroll
                for (int e = 0; e < NE; e++) k2 = min(k2, key[e] > kmin ? key[e] : MAXK);
                k2 = min(k2, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)k2, 0x111, 0xf, 0xf, false));
                k2 = min(k2, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)k2, 0x112, 0xf, 0xf, false));
                k2 = min(k2, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)k2, 0x114, 0xf, 0xf, false));
                k2 = min(k2, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)k2, 0x118, 0xf, 0xf, false));
                k2 = min(k2, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)k2, 0x142, 0xa, 0xf, false));
                k2 = min(k2, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)k2, 0x143, 0xc, 0xf, false));
                __syncthreads();  // (everyone has read the extremes)
                if (lane == 63) red[16 + wid] = k2;
                __syncthreads();
                k2 = red[16 + (lane & 7)];
                k2 = min(k2, (uint32_t)__builtin_amdgcn_update_dpp((int)k2, (int)k2, 0x124, 0xf, 0xf, false));
                k2 = min(k2, (uint32_t)__builtin_amdgcn_update_dpp((int)k2, (int)k2, 0x122, 0xf, 0xf, false));
                k2 = min(k2, (uint32_t)__builtin_amdgcn_update_dpp((int)k2, (int)k2, 0x121, 0xf, 0xf, false));
                k2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)k2);
                if (k2 < kmax) kbase = k2;  // (else: nothing lies strictly between the extremes, nothing is binned)
            }
            // an element lies strictly between the extremes <=> key - kbase < hi (unsigned: the minimum and whatever lies below
            // kbase wrap around, kmax and the missing values are >= hi)
            const uint32_t hi = (any_valid && kmax > kbase) ? kmax - kbase : 0u;
            const int nbits = hi ? 32 - __clz((int)hi) : 0;
            const int shift = max(0, nbits - log2nb);
            if (XS_SEL_STOP == 1) { if (kmin == 12345u) out_hq[0] = 0.f; __syncthreads(); continue; }

            // ---------------- B. counting pass ----------------
            // (the group bytes are made opaque per pass: what derives from them — table rows, counter addresses — is recomputed
            //  here instead of being hoisted out of the cell loop and spilled)
#pragma unroll
            for (int i = 0; i < (NE + 3) / 4; i++) __asm__ volatile("" : "+v"(gpk[i]));
#pragma unroll
            for (int e = 0; e < NE; e++) {
                __asm__ volatile("" : "+v"(key[e]));  // (nothing derived from the keys stays live from pass to pass)
                const uint32_t k = key[e];
                const uint32_t g = (gpk[e >> 2] >> (8 * (e & 3))) & 255u;
                const uint32_t rel = k - kbase;
                if (rel < hi) {
                    const uint32_t b = (g << log2nb) + (rel >> shift);
                    (void)__hip_atomic_fetch_add(&hist[b >> 1], 1u << ((b << 4) & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else if (k != MAXK) {
                    // the extreme values (plain atomics: long runs of them serialise on one LDS word — about twice the time
                    // of a run-free cell on zero-inflated data, still several times faster than sorting)
                    (void)__hip_atomic_fetch_add(&gmass[2 * g + (k != kmin ? 1 : 0)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            __syncthreads();
            if (XS_SEL_STOP == 2) { if (hist[tid] == 0x12345u) out_hq[0] = 0.f; __syncthreads(); continue; }

            // ---------------- C. the histogram -> absolute start positions (all groups end to end) ----------------
            {
                // (the thread's run of bins is read twice — totals, then prefixes — rather than held in registers across the block scan)
                const bool own = DPT * tid < ND;
                uint32_t s2 = 0u;  // both halves summed at once (each half-sum <= T < 65536)
#pragma unroll
                for (int d = 0; d < DPT / 4; d++) {
                    const uint4 v = own ? reinterpret_cast<const uint4 *>(hist)[DPT / 4 * tid + d] : make_uint4(0u, 0u, 0u, 0u);
                    s2 += v.x + v.y + v.z + v.w;
                }
                const int total = (int)(s2 & 0xFFFFu) + (int)(s2 >> 16);
                int btot;
                int run = sel_block_excl_scan(total, red, lane, wid, btot);
                pre[tid] = (uint32_t)run;
                if (tid == NT - 1) pre[NT] = (uint32_t)btot;
                if (own) {
#pragma unroll
                    for (int d = 0; d < DPT / 4; d++) {
                        uint4 v = reinterpret_cast<const uint4 *>(hist)[DPT / 4 * tid + d];
                        auto step = [&](uint32_t &w) {
                            const int c0 = (int)(w & 0xFFFFu), c1 = (int)(w >> 16);
                            w = (uint32_t)run | ((uint32_t)(run + c0) << 16);
                            run += c0 + c1;
                        };
                        step(v.x); step(v.y); step(v.z); step(v.w);
                        reinterpret_cast<uint4 *>(hist)[DPT / 4 * tid + d] = v;
                    }
                }
            }
            __syncthreads();
            if (XS_SEL_STOP == 3) { if (hist[tid] == 0x12345u) out_hq[0] = 0.f; __syncthreads(); continue; }

            // ---------------- D. scatter: every key to its bin's place (counting sort by bin) ----------------
            // (shift == 0: a bin is one key and the table alone answers — unless the group sums of DQM need the keys laid out)
            const bool scattered = DQM || shift != 0;
            if (scattered) {
                uint32_t ret[NE];
#pragma unroll
                for (int i = 0; i < (NE + 3) / 4; i++) __asm__ volatile("" : "+v"(gpk[i]));
#pragma unroll
                for (int e = 0; e < NE; e++) {
                    __asm__ volatile("" : "+v"(key[e]));
                    const uint32_t k = key[e];
                    const uint32_t g = (gpk[e >> 2] >> (8 * (e & 3))) & 255u;
                    const uint32_t rel = k - kbase;
                    ret[e] = 0xFFFFFFFFu;
                    if (rel < hi) {
                        const uint32_t b = (g << log2nb) + (rel >> shift);
                        const uint32_t old = __hip_atomic_fetch_add(&hist[b >> 1], 1u << ((b << 4) & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        ret[e] = (old >> ((b << 4) & 31u)) & 0xFFFFu;
                    }
                }
#pragma unroll
                for (int e = 0; e < NE; e++)
                    if (ret[e] != 0xFFFFFFFFu) skeys[ret[e]] = key[e];
            }
            __syncthreads();
            if (XS_SEL_STOP == 4) { if (skeys[tid] == 0x12345u) out_hq[0] = 0.f; __syncthreads(); continue; }

            // ---------------- E. one thread per (group, quantile): bin, selection inside it, interpolation, stores ----------
            if (DQM) {
                // group means: float64 sum of each group's contiguous slice of the bin-ordered keys (+ the extreme values that
                // were counted apart), one wave per group, fixed reduction tree
                for (int g = wid; g < G; g += NT / 64) {
                    const int gs = (int)pre[g * tpg], ge = (int)pre[(g + 1) * tpg];
                    double sum = 0.0;
                    for (int i = gs + lane; i < ge; i += 64) sum += (double)from_key(skeys[i]);
                    sum = wave_total_dpp(sum);
                    if (lane == 0) {
                        const int nlo = (int)gmass[2 * g], nhi = (int)gmass[2 * g + 1], n = nlo + (ge - gs) + nhi;
                        if (nlo) sum += (double)nlo * (double)from_key(kmin);
                        if (nhi) sum += (double)nhi * (double)from_key(kmax);
                        gmu[arr * SEL_MAXG + g] = n ? (float)(sum / (double)n) : __builtin_nanf("");
                    }
                }
                __syncthreads();
            }
            // The table now holds every bin's END position (scattered) or still its start (shift == 0, not scattered).
            for (int o = tid; o < G * Q; o += NT) {
                const int g = o / Q, kq = o - g * Q;
                const int gstart = (int)pre[g * tpg], gend = (int)pre[(g + 1) * tpg];
                const int nlo = (int)gmass[2 * g], nhi = (int)gmass[2 * g + 1];
                const int n = nlo + (gend - gstart) + nhi;
                const uint16_t *tab = hist16 + ((size_t)g << log2nb);
                // bin (index inside the group) whose keys hold absolute position pa
                auto bin_of = [&](int pa) {
                    int b = 0;
                    if (scattered) {  // END positions: the first bin whose end exceeds pa
                        for (int s = NB >> 1; s > 0; s >>= 1) b = (int)tab[b + s - 1] <= pa ? b + s : b;
                    } else {           // start positions: the last bin that starts at or before pa
                        for (int s = NB >> 1; s > 0; s >>= 1) b = (int)tab[b + s] <= pa ? b + s : b;
                    }
                    return b;
                };
                float res = __builtin_nanf("");
                int o0 = 0;
                bool two = false;
                float gamma = 0.f;
                if (n > 0) sel_ordinals(n, q[kq], o0, two, gamma);
                float mu = 0.f, inv = 0.f;
                if (DQM) {
                    mu = gmu[arr * SEL_MAXG + g];
                    inv = invert<float>(mu, kind);
                    if (n > 0 && (!isfinite(mu) || (kind == '*' && !(inv > 0.f && isfinite(inv))))) flag_cell = true;
                    if (kq == 0 && arr == 1) scaling[(size_t)c * G + g] = get_corr<float>(gmu[g], mu, kind);  // corr(mean(hist), mean(ref))
                }
                // first the quantile's own neighbours; a second round only for the reference's NaN rule (inf - inf): a[n - 1]
#pragma nounroll
                for (int round = 0; round < 2 && n > 0; round++) {
                    const int od = round == 0 ? o0 : n - 1;
                    const bool tw = round == 0 && two;
                    // ---- keys of ordinal od and, when tw, of od + 1
                    uint32_t k0 = MAXK, k1 = MAXK;
                    const int o1 = od + 1;
                    if (od < nlo) k0 = kmin; else if (od >= n - nhi) k0 = kmax;
                    if (tw) { if (o1 < nlo) k1 = kmin; else if (o1 >= n - nhi) k1 = kmax; }
                    const bool need0 = od >= nlo && od < n - nhi, need1 = tw && o1 >= nlo && o1 < n - nhi;
                    if (need0 || need1) {
                        const int pa = gstart + ((need0 ? od : o1) - nlo);  // the first position that has to be read
                        const int b = bin_of(pa);
                        if (!scattered) {
                            const uint32_t kb = kbase + (uint32_t)b;
                            if (need0) { k0 = kb; if (need1) k1 = kbase + (uint32_t)bin_of(pa + 1); }
                            else k1 = kb;
                        } else {
                            const int s0 = b > 0 ? (int)tab[b - 1] : gstart, e0 = (int)tab[b];
                            const int m = e0 - s0, r = pa - s0;
                            uint32_t a0 = MAXK, a1 = MAXK;
                            if (m <= SEL_CAPBIN) {
                                const int sa = s0 & ~3, lead = s0 - sa, mm = m + lead, rr = r + lead;
                                const uint32_t *sla = skeys + sa;
                                if (mm <= 8) sel_sorted_pick2<8>(sla, lead, mm, rr, a0, a1);
                                else if (mm <= 16) sel_sorted_pick2<16>(sla, lead, mm, rr, a0, a1);
                                else if (mm <= 32) sel_sorted_pick2<32>(sla, lead, mm, rr, a0, a1);
                                else sel_sorted_pick2<64>(sla, lead, mm, rr, a0, a1);
                            } else {
                                // a long run: resolved only when it is one value
                                uint32_t mn = MAXK, mx = 0u;
                                for (int i = 0; i < m; i++) { const uint32_t kj = skeys[s0 + i]; mn = min(mn, kj); mx = max(mx, kj); }
                                if (mn == mx) a0 = a1 = mn; else flag_cell = true;
                            }
                            if (need0) {
                                k0 = a0;
                                if (need1) {
                                    if (r + 1 < m) k1 = a1;
                                    else {  // the minimum of the next bin that holds anything
                                        int b1 = b + 1;
                                        while (b1 < NB - 1 && (int)tab[b1] == e0) b1++;
                                        const int e1 = (int)tab[b1];
                                        uint32_t mn = MAXK;
                                        for (int i = e0; i < e1; i++) mn = min(mn, skeys[i]);
                                        k1 = mn;
                                    }
                                }
                            } else k1 = a0;  // (od itself was an extreme value: position pa is od + 1)
                        }
                    }
                    // ---- device_math.hpp::quantile_sorted with a(o0) = k0, a(o0 + 1) = k1
                    float v0 = from_key(k0), v1 = from_key(k1);
                    if (DQM) { v0 = apply_corr<float>(v0, inv, kind); v1 = apply_corr<float>(v1, inv, kind); }
                    if (round == 1) { res = v0; break; }
                    if (!two) { res = v0; break; }
                    const float diff = v1 - v0;
                    if (gamma >= 0.5f) res = (float)((double)v1 - (double)diff * (1.0 - (double)gamma));
                    else res = v0 + diff * gamma;
                    if (res == res) break;
                    if (DQM) { flag_cell = true; break; }  // inf - inf: the generic kernel redoes the cell
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
