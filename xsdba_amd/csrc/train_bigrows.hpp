// train_bigrows.hpp — k_train_rows for group rows that do not fit the LDS (round 5).
//
// nbutils._nan_quantile_1d sorts whatever it is given (nbutils.py:137): a 30-year HOURLY series under group="time" is one row
// of 262 800 values per cell.  k_train_rows sorts a row in LDS (<= 32 768 float32 / 16 384 float64 keys); beyond that the same
// pipeline runs with the rows in global memory, a batch of rows at a time:
//   k_big_fill    one block per (cell, group) row: gather the row through row_idx into P = next_pow2(L) order-preserving keys
//                 (NaN and padding -> MAXK, sorted last), the valid count; mode 2 (dqm_train): the NaN-skipping mean first
//                 (float64 block reduction), the keys are those of the normalised values;
//   bitonic sort  of every row of the batch: tiles of BIG_TILE keys are sorted in LDS (k_big_local: every stage up to the tile
//                 size), then for k = 2 BIG_TILE .. P the stages whose partner lies outside the tile run on global memory
//                 (k_big_step) and the rest again in LDS (k_big_local with k0 = k);
//   k_big_quant   the type-7 quantiles off the sorted keys with the very routine of k_train_rows (quantile_sorted), ref_q kept
//                 in a [rows][Q] scratch between the two passes, af / scaling as there; mode 3: vecquantiles.
// Same results as k_train_rows bit for bit for modes 0 / 1 / 3 (a sorted row is a sorted row); mode 2 differs by the
// reduction order of the group mean (float64: within the DQM tolerance of the tests).
#pragma once
#include "kernels.hpp"

namespace xs {

constexpr int BIG_TILE = 8192;  // keys a block sorts in LDS (32 KB float32 / 64 KB float64), 1024 threads

template <typename R>
__global__ __launch_bounds__(1024) void k_big_fill(const R *__restrict__ x, int64_t st, int64_t sc, const int *__restrict__ row_ptr,
                                                    const int *__restrict__ row_idx, int G, int64_t row0, int P, int kind, int dqm,
                                                    typename KeyOf<R>::U *__restrict__ keys, int *__restrict__ nvalid_out,
                                                    R *__restrict__ mu_out)
{
    using U = typename KeyOf<R>::U;
    __shared__ double red[2 * 16 + 2];
    const int64_t b = row0 + blockIdx.x;            // the (cell, group) row
    const int64_t c = b / G;
    const int g = (int)(b - c * G);
    const int r0 = row_ptr[g], L = row_ptr[g + 1] - r0;
    const R *xc = x + c * sc;
    U *kr = keys + (size_t)blockIdx.x * (size_t)P;
    R inv = (R)0;
    if (dqm) {
        double s = 0;
        int n = 0;
        for (int i = threadIdx.x; i < L; i += blockDim.x) {
            const R v = xc[(int64_t)row_idx[r0 + i] * st];
            if (v == v) { s += (double)v; n++; }
        }
        block_sum_count(s, n, red);
        const R mu = n ? (R)(s / (double)n) : nan_of<R>();
        if (threadIdx.x == 0) mu_out[blockIdx.x] = mu;
        inv = invert<R>(mu, kind);
    }
    int nv = 0;
    for (int i = threadIdx.x; i < P; i += blockDim.x) {
        U k = KeyOf<R>::MAXK;
        if (i < L) {
            R v = xc[(int64_t)row_idx[r0 + i] * st];
            if (dqm) v = apply_corr<R>(v, inv, kind);
            k = to_key(v);
            nv += (v == v);
        }
        kr[i] = k;
    }
    double dummy = 0;
    block_sum_count(dummy, nv, red);
    if (threadIdx.x == 0) nvalid_out[blockIdx.x] = nv;
}

// One tile of BIG_TILE keys of one row in LDS: the stages (k, j) with j < BIG_TILE of the bitonic network, for k = 2 .. BIG_TILE
// (k0 == 0: the tile is sorted, ascending or descending by its place in the row) or for the single k = k0 > BIG_TILE (the tail of
// a merge level whose wide stages k_big_step has done).  grid = rows * (P / BIG_TILE).
template <typename U>
__global__ __launch_bounds__(1024) void k_big_local(U *__restrict__ keys, int P, int k0)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_big[];
    U *t = reinterpret_cast<U *>(smem_big);
    const int tiles = P / BIG_TILE;
    const int64_t row = blockIdx.x / tiles;
    const int tile = (int)(blockIdx.x - row * tiles);
    U *src = keys + (size_t)row * (size_t)P + (size_t)tile * BIG_TILE;
    for (int i = threadIdx.x; i < BIG_TILE; i += blockDim.x) t[i] = src[i];
    __syncthreads();
    const int base = tile * BIG_TILE;  // place of the tile's first key in its row: the direction of a stage depends on it
    auto stages = [&](int k, int jtop) {
        for (int j = jtop; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < BIG_TILE / 2; i += blockDim.x) {
                const int l = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                const int r = l | j;
                const U a = t[l], b = t[r];
                const bool up = ((base + l) & k) == 0;
                if ((a > b) == up) { t[l] = b; t[r] = a; }
            }
            __syncthreads();
        }
    };
    if (k0 == 0) {
        for (int k = 2; k <= BIG_TILE; k <<= 1) stages(k, k >> 1);
    } else stages(k0, BIG_TILE >> 1);
    for (int i = threadIdx.x; i < BIG_TILE; i += blockDim.x) src[i] = t[i];
}

// One stage (k, j), j >= BIG_TILE, on global memory: a thread per pair.  grid covers rows * P / 2 pairs.
template <typename U>
__global__ __launch_bounds__(256) void k_big_step(U *__restrict__ keys, int P, int k, int j, int64_t npairs)
{
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= npairs) return;
    const int half = P >> 1;
    const int64_t row = id / half;
    const int i = (int)(id - row * half);
    const int l = ((i & ~(j - 1)) << 1) | (i & (j - 1));
    const int r = l | j;
    U *kr = keys + (size_t)row * (size_t)P;
    const U a = kr[l], b = kr[r];
    const bool up = (l & k) == 0;
    if ((a > b) == up) { kr[l] = b; kr[r] = a; }
}

// pass 0 (x0: the quantiles' only array in modes 0 / 3, ref in modes 1 / 2) and pass 1 (hist) of k_train_rows on sorted rows
template <typename R>
__global__ __launch_bounds__(64) void k_big_quant(const typename KeyOf<R>::U *__restrict__ keys, const int *__restrict__ nvalid, int P, int G,
                                                   int64_t row0, const R *__restrict__ q, int Q, int kind, int mode, int pass,
                                                   R *__restrict__ af, R *__restrict__ out_hq, R *__restrict__ scaling, R *__restrict__ refq,
                                                   const R *__restrict__ mu_ref, const R *__restrict__ mu_hist)
{
    const int64_t b = row0 + blockIdx.x;
    const int64_t c = b / G;
    const int g = (int)(b - c * G);
    const size_t obase = ((size_t)c * G + g) * (size_t)Q;
    const typename KeyOf<R>::U *kr = keys + (size_t)blockIdx.x * (size_t)P;
    const int nv = nvalid[blockIdx.x];
    auto load = [&](int i) { return from_key(kr[i]); };
    if (mode == 3) {
        if (threadIdx.x == 0) out_hq[c] = vecquantile_sorted<R>(load, nv, q[c]);
        return;
    }
    for (int k = threadIdx.x; k < Q; k += blockDim.x) {
        const R v = quantile_sorted<R>(load, nv, q[k]);
        if (mode == 0) out_hq[obase + k] = v;
        else if (pass == 0) refq[(size_t)blockIdx.x * Q + k] = v;
        else {
            out_hq[obase + k] = v;
            af[obase + k] = get_corr<R>(v, refq[(size_t)blockIdx.x * Q + k], kind);
        }
    }
    if (mode == 2 && pass == 1 && threadIdx.x == 0) scaling[(size_t)c * G + g] = get_corr<R>(mu_hist[blockIdx.x], mu_ref[blockIdx.x], kind);
}

}  // namespace xs
