// train_cell_f64.hpp — the per-cell training kernel for float64 series (T <= 11 264).
//
// The LDS cannot hold 64-bit keys next to the sort scratch, and a 64-bit radix sort would need ~13 passes.  Instead the
// series is sorted by a 31-bit key that keeps the TOP bits of the occupied range of its order-preserving 64-bit keys,
//     key31 = (key64(x) - min key64) >> shift,   shift = max(0, bits(max - min) - 31),
// a monotone map, so the 32-bit sort of k_train_cell applies unchanged.  Equal doubles give equal keys in any order;
// distinct doubles collide only when they agree in their top 31 range bits (never for data that came from float32,
// ~T^2 / 2^32 pairs for continuous data) and are put in order by an odd-even transposition restricted to runs of
// equal keys, comparing the doubles themselves (read from global memory / L2 only for elements inside such runs).
// Phase C is the one of k_train_cell instantiated for double: order statistics and DQM sums come from the series in
// global memory through the sorted index array; quantile interpolation, factors and scaling are computed in float64.
#pragma once
#include "train_cell3.hpp"

namespace xs {

// values of a float64 series: the series itself (global memory / L2), addressed through the sorted index array in LDS
struct GlobalF64Vals {
    using value_type = double;
    const double *x;
    const uint16_t *sorted;
    int T;
    __device__ __forceinline__ double at_time(int t) const { return x[t]; }
    __device__ __forceinline__ void at_ranks(int r0, int r1, double &v0, double &v1) const
    {
        const int id0 = min((int)sorted[r0], T - 1), id1 = min((int)sorted[r1], T - 1);
        v0 = x[id0];
        v1 = x[id1];
    }
};

// Phases A + B for a float64 series: 31-bit range keys in LDS, radix sort, tie repair, rank[].  Returns the sorted
// index array.  All 1024 threads must call it; it ends on a barrier.
// rank_out != nullptr: rank[] goes into the u16 index buffer the sort leaves free (k_train_cell3_f64: the union region is
// then all bit-vectors) and its address is returned there; else into the union region as for k_train_cell.
template <int EPT>
__device__ __forceinline__ const uint16_t *cell_load_sort_f64(const double *__restrict__ x, int64_t T, unsigned char *smem,
                                                              uint32_t *red, int &nvalid_out, const uint16_t **rank_out = nullptr)
{
    using Cfg = CellCfg<EPT>;
    constexpr int NT = Cfg::NT, TPAD = Cfg::TPAD;
    uint32_t *keys = reinterpret_cast<uint32_t *>(smem + Cfg::off_keys);
    uint16_t *rank = reinterpret_cast<uint16_t *>(smem + Cfg::off_rank);
    uint16_t *idx0 = reinterpret_cast<uint16_t *>(smem + Cfg::off_idx0), *idx1 = reinterpret_cast<uint16_t *>(smem + Cfg::off_idx1);
    int tid = threadIdx.x;
    __asm__ volatile("" : "+v"(tid));  // see cell_load_sort
    const int lane = tid & 63, wid = tid >> 6;
    const int Ti = (int)T;
    // ---- A. load (unconditional, clamped), 64-bit keys, occupied range
    uint64_t k64[EPT];
#pragma unroll
    for (int e = 0; e < EPT; e++) {
        const int i = tid + e * NT;
        const double v = x[min(i, Ti - 1)];
        k64[e] = i < Ti ? to_key(v) : 0xFFFFFFFFFFFFFFFFull;
    }
    int nv = 0;
    uint64_t kmin = 0xFFFFFFFFFFFFFFFFull, kmax = 0ull;
#pragma unroll
    for (int e = 0; e < EPT; e++)
        if (k64[e] != 0xFFFFFFFFFFFFFFFFull) { nv++; kmin = k64[e] < kmin ? k64[e] : kmin; kmax = k64[e] > kmax ? k64[e] : kmax; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        nv += __shfl_down(nv, o, 64);
        const uint64_t a = __shfl_down(kmin, o, 64), b = __shfl_down(kmax, o, 64);
        kmin = a < kmin ? a : kmin;
        kmax = b > kmax ? b : kmax;
    }
    uint64_t *red64 = reinterpret_cast<uint64_t *>(red);  // [NWAVE][2] + counts behind them
    if (lane == 0) { red64[2 * wid] = kmin; red64[2 * wid + 1] = kmax; red[4 * Cfg::NWAVE + wid] = (uint32_t)nv; }
    __syncthreads();
    nv = 0; kmin = 0xFFFFFFFFFFFFFFFFull; kmax = 0ull;
    for (int w = 0; w < Cfg::NWAVE; w++) {
        nv += (int)red[4 * Cfg::NWAVE + w];
        kmin = red64[2 * w] < kmin ? red64[2 * w] : kmin;
        kmax = red64[2 * w + 1] > kmax ? red64[2 * w + 1] : kmax;
    }
    const int nvalid = nv;
    nvalid_out = nv;
    const uint64_t range64 = nvalid ? kmax - kmin : 0ull;
    const int nb = 64 - __clzll((long long)(range64 | 1ull));
    const int shift = nb > 31 ? nb - 31 : 0;
#pragma unroll
    for (int e = 0; e < EPT; e++) {
        const int i = tid + e * NT;
        keys[i] = k64[e] != 0xFFFFFFFFFFFFFFFFull ? (uint32_t)((k64[e] - kmin) >> shift) : 0xFFFFFFFFu;
    }
    __syncthreads();
    // ---- B. the 32-bit sort (keys are already reduced: kmin = 0), without rank[]
    uint16_t *sorted = const_cast<uint16_t *>(radix_sort_indexes<EPT>(smem, red, tid, nvalid, 0u, (uint32_t)(range64 >> shift), 2));
    // ---- tie repair: inside runs of equal keys, order by the doubles themselves (odd-even transposition)
    for (int guard = 0; guard < TPAD; guard++) {
        bool swapped = false;
        for (int phase = 0; phase < 2; phase++) {
            for (int p = 2 * tid + phase; p + 1 < nvalid; p += 2 * NT) {
                const int a = sorted[p], b = sorted[p + 1];
                if (keys[a] == keys[b] && x[a] > x[b]) { sorted[p] = (uint16_t)b; sorted[p + 1] = (uint16_t)a; swapped = true; }
            }
            __syncthreads();
        }
        if (!__syncthreads_or(swapped ? 1 : 0)) break;
    }
    if (rank_out) { rank = sorted == idx0 ? idx1 : idx0; *rank_out = rank; }
    for (int r = tid; r < TPAD; r += NT) rank[sorted[r]] = (uint16_t)r;
    __syncthreads();
    return sorted;
}

template <int EPT, bool DQM, bool TOG64>
__global__ __launch_bounds__(1024) void k_train_cell_f64(
    const double *__restrict__ x0, const double *__restrict__ x1, int64_t T, int64_t sc, const int *__restrict__ row_ptr,
    const int *__restrict__ row_sorted, const int *__restrict__ tog_ptr, const int *__restrict__ tog_idx, int G,
    const double *__restrict__ q, int Q, int kind, int mode, double *__restrict__ af, double *out_hq, double *scaling,
    int *__restrict__ flag_count, int *__restrict__ flag_list, int64_t C, const int *__restrict__ tog64)
{
    using Cfg = CellCfg<EPT>;
    constexpr int NT = Cfg::NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t *rank = reinterpret_cast<uint16_t *>(smem + Cfg::off_rank);
    uint32_t *bits_all = reinterpret_cast<uint32_t *>(smem + Cfg::off_bits);
    uint32_t *red = reinterpret_cast<uint32_t *>(smem + Cfg::off_misc);
    const int tid = threadIdx.x, wid = tid >> 6;
    auto nostamp = [](int) {};
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        const int npass_arr = mode == 0 ? 1 : 2;
        bool flag_cell = false;
        for (int arr = 0; arr < npass_arr; arr++) {
            int lane = tid & 63;
            __asm__ volatile("" : "+v"(lane));
            const double *x = ((mode == 0 || arr == 1) ? x0 : x1) + c * sc;  // hist first, then ref (see k_train_cell)
            int nvalid;
            const uint16_t *sorted = cell_load_sort_f64<EPT>(x, T, smem, red, nvalid);
            if (G == 1 && !DQM && row_ptr[1] - row_ptr[0] == (int)T) {  // group="time": the sorted order is the row
                const int n = nvalid;
                for (int k = tid; k < Q; k += NT) {
                    auto load = [&](int i) { return x[sorted[i]]; };
                    const double res = n ? quantile_sorted<double>(load, n, q[k]) : __builtin_nan("");
                    const size_t o = (size_t)c * Q + k;
                    if (mode == 0 || arr == 0) out_hq[o] = res;
                    else af[o] = get_corr<double>(out_hq[o], res, kind);
                }
                __syncthreads();
                continue;
            }
            {
                const GlobalF64Vals vals{x, sorted, (int)T};
                cell_group_quantiles<Cfg, false, DQM, TOG64>(vals, rank, bits_all, nvalid, lane, wid, row_ptr, row_sorted, tog_ptr,
                                                             tog_idx, tog64, G, q, Q, kind, mode, arr, c, af, out_hq, scaling,
                                                             flag_cell, 0, nostamp);
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

// three group streams per wave (train_cell3.hpp) for float64 series
template <int EPT, bool P10, bool DQM>
__global__ __launch_bounds__(1024) void k_train_cell3_f64(const double *__restrict__ x0, const double *__restrict__ x1, int64_t T,
                                                           int64_t sc, const int *__restrict__ row_ptr,
                                                           const int *__restrict__ row_sorted, int G, const double *__restrict__ q,
                                                           int Q, int kind, int mode, double *__restrict__ af, double *out_hq,
                                                           double *scaling, int *__restrict__ flag_count,
                                                           int *__restrict__ flag_list, int64_t C, const int *__restrict__ tog64)
{
    using Cfg3 = CellCfg3<EPT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *bits_all = reinterpret_cast<uint32_t *>(smem + Cfg3::off_bits);
    uint32_t *red = reinterpret_cast<uint32_t *>(smem + Cfg3::off_misc);
    const int tid = threadIdx.x, wid = tid >> 6;
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        const int npass_arr = mode == 0 ? 1 : 2;
        bool flag_cell = false;
        for (int arr = 0; arr < npass_arr; arr++) {
            int lane = tid & 63;
            __asm__ volatile("" : "+v"(lane));
            const double *x = ((mode == 0 || arr == 1) ? x0 : x1) + c * sc;
            int nvalid;
            const uint16_t *rank;
            const uint16_t *sorted = cell_load_sort_f64<EPT>(x, T, smem, red, nvalid, &rank);
            const GlobalF64Vals vals{x, sorted, (int)T};
            cell_group_quantiles3<EPT, P10, DQM>(vals, rank, bits_all, nvalid, lane, wid, row_ptr, row_sorted, tog64, G, q, Q, kind, mode,
                                                 arr, c, af, out_hq, scaling, flag_cell);
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
