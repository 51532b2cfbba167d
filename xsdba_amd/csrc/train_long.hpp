// train_long.hpp — the per-cell training kernel for series of 11 265 .. 22 528 time steps (e.g. 60 years daily).
//
// One series no longer fits the LDS next to its sort scratch, so the sort is split:
//   1. each HALF of the series goes through cell_load_sort<11> (the sort of k_train_cell); its keys in rank order and
//      the matching time indexes are parked in a per-block global scratch slice (L2-resident);
//   2. MERGE: both key sequences come back into LDS; merge path — every thread finds by one bisection where its 22 places
//      of the merged order start in the two halves (equal keys: first half first) and merges sequentially from there ->
//      rank[t] (LDS) and the keys in final rank order (global scratch);
//   3. phase C is the one of k_train_cell (cell_group_quantiles) with 704-word vectors; values are fetched from the
//      series itself (DQM sums) and from the rank-ordered key array (order statistics), both L2 hits.
// Same results as k_train_cell / k_train_rows: the rank order is a total order of the same keys, ties broken by time
// (which never changes a quantile value).
#pragma once
#include "train_cell.hpp"

namespace xs {

// EPT: elements per thread of a half's sort — 11 (series up to 22 528 steps) or 8 (up to 16 384: 31 to 44 years of daily steps pay
// for 16 384 places instead of 22 528)
template <int EPT> struct LongCfg {
    using H = CellCfg<EPT>;
    static constexpr int NT = 1024, HALF = H::TPAD, TPAD = 2 * HALF;
    static constexpr int NWORDS = TPAD / 32, WPL = (NWORDS + 63) / 64, BITW = WPL * 64, NWAVE = NT / 64;
    static constexpr int TOGN = 2;  // the padded toggle table of this geometry: [G][128] (60 years of day-of-year windows: 120 steps)
    // merge view of the LDS: [K0 | K1 | rank]; phase-C view: [2 * NWAVE vectors | rank]
    static constexpr size_t off_k0 = 0, off_k1 = (size_t)HALF * 4, off_rank = (size_t)TPAD * 4;
    static constexpr size_t off_bits = 0;
    static_assert((size_t)2 * NWAVE * BITW * 4 <= off_rank, "the vectors must end before rank[]");
    static_assert(off_rank + (size_t)TPAD * 2 <= H::off_misc, "rank[] must end before the reduction scratch");
    static constexpr size_t lds_bytes = H::lds_bytes;
    // per-block global scratch: keys in rank order of each half, their time indexes, keys in final rank order
    static constexpr size_t scr_ks = 0, scr_ts = (size_t)TPAD * 4, scr_by_rank = scr_ts + (size_t)TPAD * 2;
    static constexpr size_t scratch_bytes = scr_by_rank + (size_t)TPAD * 4;
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

struct GlobalValsL2 {
    using value_type = float;
    const float *x;
    const uint32_t *by_rank;
    __device__ __forceinline__ float at_time(int t) const { return x[t]; }
    __device__ __forceinline__ void at_ranks(int r0, int r1, float &v0, float &v1) const
    {
        const uint32_t k0 = load_l2(by_rank + r0), k1 = load_l2(by_rank + r1);
        v0 = from_key(k0);
        v1 = from_key(k1);
    }
};

// Phase C behind a call: the walk's register allocation is then its own (inlined, its loops shared the 128 registers of a
// 1024-thread block with whatever the sorts and the merge keep alive across the cell loop, and spilled in its group loop).
template <bool DQM, bool TOG64, int EPT>
__device__ __attribute__((noinline)) void long_phase_c(const float *x, const uint32_t *by_rank, const uint16_t *rank, uint32_t *bits_all, int nvalid,
                                                       int lane, int wid, const int *row_ptr, const int *row_sorted, const int *tog_ptr,
                                                       const int *tog_idx, const int *tog64, int G, const float *q, int Q, int kind, int mode,
                                                       int arr, int64_t c, float *af, float *out_hq, float *scaling, bool &flag_cell)
{
    auto nostamp = [](int) {};
    const GlobalValsL2 vals{x, by_rank};
    cell_group_quantiles<LongCfg<EPT>, false, DQM, TOG64>(vals, rank, bits_all, nvalid, lane, wid, row_ptr, row_sorted, tog_ptr, tog_idx, tog64, G, q, Q,
                                                     kind, mode, arr, c, af, out_hq, scaling, flag_cell, 0, nostamp);
}

template <bool DQM, bool TOG64, int EPT = 11>
__global__ __launch_bounds__(1024) void k_train_long(
    const float *__restrict__ x0, const float *__restrict__ x1, int64_t T, int64_t sc, const int *__restrict__ row_ptr,
    const int *__restrict__ row_sorted, const int *__restrict__ tog_ptr, const int *__restrict__ tog_idx, int G,
    const float *__restrict__ q, int Q, int kind, int mode, float *__restrict__ af, float *out_hq, float *scaling,
    int *__restrict__ flag_count, int *__restrict__ flag_list, int64_t C, const int *__restrict__ tog64,
    unsigned char *__restrict__ gscratch, int debug_phase)
{
    using H = CellCfg<EPT>;
    using L = LongCfg<EPT>;
    constexpr int NT = L::NT, HALF = L::HALF;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t *keys = reinterpret_cast<const uint32_t *>(smem + H::off_keys);  // sort view
    uint32_t *K0 = reinterpret_cast<uint32_t *>(smem + L::off_k0), *K1 = reinterpret_cast<uint32_t *>(smem + L::off_k1);
    uint16_t *rank = reinterpret_cast<uint16_t *>(smem + L::off_rank);
    uint32_t *bits_all = reinterpret_cast<uint32_t *>(smem + L::off_bits);
    uint32_t *red = reinterpret_cast<uint32_t *>(smem + H::off_misc);
    unsigned char *scr = gscratch + (size_t)blockIdx.x * L::scratch_bytes;
    uint32_t *ks = reinterpret_cast<uint32_t *>(scr + L::scr_ks);        // [2][HALF]
    uint16_t *ts = reinterpret_cast<uint16_t *>(scr + L::scr_ts);        // [2][HALF]
    uint32_t *by_rank = reinterpret_cast<uint32_t *>(scr + L::scr_by_rank);

    const int tid = threadIdx.x, wid = tid >> 6;
    const int Ti = (int)T;
    const int T0 = ((Ti / 2) + 1) & ~1;  // even: the second half keeps the 8-byte alignment of the row
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        const int npass_arr = mode == 0 ? 1 : 2;
        bool flag_cell = false;
        for (int arr = 0; arr < npass_arr; arr++) {
            int lane = tid & 63;
            __asm__ volatile("" : "+v"(lane));
            const float *x = ((mode == 0 || arr == 1) ? x0 : x1) + c * sc;
            // ---------------- 1. sort each half ----------------
            // The first half's keys in rank order are parked in the block's global scratch slice (the second sort needs the
            // whole LDS), and so are the time steps of both halves in rank order.
            constexpr int EPH = HALF / NT;
            int nv[2];
            for (int h = 0; h < 2; h++) {
                const int Th = h ? Ti - T0 : T0;
                const uint16_t *sorted = cell_load_sort<EPT>(x + (h ? T0 : 0), (int64_t)Th, smem, red, 2, nv[h], NoLoadHook(), -1, true);
                if (h == 0) {
                    if (sorted) {
                        for (int r = tid; r < HALF; r += NT) {
                            const int id = sorted[r];
                            ks[r] = keys[min(id, HALF - 1)];
                            ts[r] = (uint16_t)id;
                        }
                    } else {  // a half without a valid value was not sorted: all its steps are NaN steps, in any order
                        for (int r = tid; r < HALF; r += NT) { ks[r] = 0xFFFFFFFFu; ts[r] = (uint16_t)r; }
                    }
                    stores_done();
                    __syncthreads();  // the next sort overwrites keys / sorted
                } else {
                    // the second half's keys go straight to their place of the merge (K1 lies over the index buffers)
                    uint32_t k1r[EPH];
#pragma unroll
                    for (int j = 0; j < EPH; j++) {
                        const int r = tid + j * NT;
                        const int id = sorted ? (int)sorted[r] : r;
                        ts[HALF + r] = (uint16_t)id;
                        k1r[j] = sorted ? keys[min(id, HALF - 1)] : 0xFFFFFFFFu;
                    }
                    __syncthreads();
#pragma unroll
                    for (int j = 0; j < EPH; j++) K1[tid + j * NT] = k1r[j];
                    stores_done();
                }
            }
            const int nv0 = nv[0], nv1 = nv[1], nvalid = nv0 + nv1;
            if (debug_phase == 1) continue;  // timing aid: the two half sorts only
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
            // ---------------- 2. merge: final rank of every time step ----------------
            for (int r = tid; r < HALF; r += NT) K0[r] = load_l2(ks + r);
            __syncthreads();
            if (debug_phase == 11) continue;
            {
                // The final rank of a step = its place in its own half + the number of keys of the other half in front of it
                // (equal keys: the first half first).  A thread owns 11 consecutive places of a half: two full bisections (14
                // probes each, independent) place its first and last key in the other half, the nine keys between them are
                // bisected inside that bracket — about 11 places wide on average, so a handful of probes instead of 14 —, all
                // nine chains in flight together.  Replaces a merge path whose 22 dependent compare-and-advance steps per
                // thread cost 13.9 ms per 32 400 cells, and one full bisection per element (6.1 ms): profiles/r03_long_phases.txt.
                auto place_half = [&](auto hsel, const uint32_t (&own)[EPH], const uint32_t (&id)[EPH]) {
                    constexpr int h = decltype(hsel)::value;
                    const uint32_t *Koth = h ? K0 : K1;  // (Koth[n - 1] = the n-th key of the other half, counted from 1)
                    const int nown = h ? nv1 : nv0, noth = h ? nv0 : nv1, toff = h ? T0 : 0, Th = h ? Ti - T0 : T0;
                    // before(kk, k): a key kk of the other half sorts in front of the key k of this one
                    auto before = [](uint32_t kk, uint32_t k) { return h ? kk <= k : kk < k; };
                    int lo = 0, hi = 0;
#pragma unroll 1
                    for (int step = 8192; step; step >>= 1) {
                        const int nl = lo + step, nh = hi + step;
                        // (a probe past the other half's valid keys reads some word of the LDS and is discarded)
                        const uint32_t kl = Koth[nl - 1], kh = Koth[nh - 1];
                        lo = (nl <= noth && before(kl, own[0])) ? nl : lo;
                        hi = (nh <= noth && before(kh, own[EPH - 1])) ? nh : hi;
                    }
                    int pos[EPH];
#pragma unroll
                    for (int j = 0; j < EPH; j++) pos[j] = lo;
                    pos[EPH - 1] = hi;
                    const int width = hi - lo;
#pragma unroll 1
                    for (int step = 8192; step; step >>= 1) {
                        if (!__any(step <= width)) continue;  // (wave-uniform)
#pragma unroll
                        for (int j = 1; j < EPH - 1; j++) {
                            const int np = pos[j] + step;
                            const uint32_t kk = Koth[np - 1];
                            pos[j] = (np <= hi && before(kk, own[j])) ? np : pos[j];
                        }
                    }
                    if (debug_phase == 12) { int a_ = 0; for (int j = 0; j < EPH; j++) a_ += pos[j] + (int)id[j]; rank[tid + h * NT] = (uint16_t)a_; return; }  // (timing aid)
#pragma unroll
                    for (int j = 0; j < EPH; j++) {
                        const int r = tid * EPH + j;
                        if (r < nown) {
                            const int fin = r + pos[j];
                            rank[(int)id[j] + toff] = (uint16_t)fin;
                            if (debug_phase != 13) by_rank[fin] = own[j];
                        } else if ((int)id[j] < Th) rank[(int)id[j] + toff] = 0xFFFFu;  // NaN steps: never a member
                    }
                };
                {
                    uint32_t kr[EPH], idr[EPH];
#pragma unroll
                    for (int j = 0; j < EPH; j++) { kr[j] = K1[tid * EPH + j]; idr[j] = load_l2(ts + HALF + tid * EPH + j); }
                    place_half(std::integral_constant<int, 1>{}, kr, idr);
                }
                {
                    uint32_t kr[EPH], idr[EPH];
#pragma unroll
                    for (int j = 0; j < EPH; j++) { kr[j] = K0[tid * EPH + j]; idr[j] = load_l2(ts + tid * EPH + j); }
                    place_half(std::integral_constant<int, 0>{}, kr, idr);
                }
            }
            stores_done();
            __syncthreads();  // rank[] complete, K0 / K1 dead: the vectors take their place
            if (debug_phase == 2 || debug_phase == 12 || debug_phase == 13) continue;  // timing aid: sorts + merge
            // ---------------- 3. group quantiles ----------------
            long_phase_c<DQM, TOG64, EPT>(x, by_rank, rank, bits_all, nvalid, lane, wid, row_ptr, row_sorted, tog_ptr, tog_idx, tog64, G, q, Q, kind, mode,
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
