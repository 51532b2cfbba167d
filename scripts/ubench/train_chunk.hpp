// train_chunk.hpp — windowed training as a WAVE per (cell, chunk of consecutive groups) in a LOCAL rank space (round 5).
//
// k_train_cell3 sorts the whole series once per block and walks every group's membership as a bit-vector over all T ranks:
// 343 words per vector, six per lane, so that a group step (three groups x 20 quantiles) recounts eighteen words per lane and
// every select goes owner lane -> word among eight -> bit: ~390 instructions per step, and the loop is bound by instruction
// issue (profiles/r03_issue_rate.txt: ~2.8 cycles per instruction of this mix at four waves per SIMD).  But the groups of one
// wave only ever touch the steps of their own windows: a chunk of 23 day-of-year groups with a 31-day window is 53 columns of
// the (year, day) matrix — 1 590 of the 10 950 steps.  Ranked among themselves, those steps give a vector of 50 words: ONE
// word per lane.  So here
//   * the plan cuts the groups in chunks whose rows together hold at most NE - 32 steps (api.hip: chunk tables — the steps of
//     each chunk in time order, the middle group's row and every toggle list as LOCAL indexes into them);
//   * a wave takes a (cell, chunk) task on its own — no block barrier anywhere: it gathers the chunk's steps from the series
//     (coalesced runs of the window's columns; a step is fetched by ~2.3 waves, from L2), sorts them with the wave-private
//     stable counting passes of train_cell.hpp (8-bit digits on range-reduced keys, lane-ordered returning LDS atomics, the
//     keys in registers across a pass so that it sorts in place) and leaves the values in rank order + the local rank of
//     every step in 6 bytes per step of LDS;
//   * the walk is the three-stream walk of train_cell3.hpp on one-word-per-lane vectors: the recount is one popcount per
//     stream, the owner lane's word IS the word (no word-level select), the next member is nearly always in the same or
//     the next word (the vectors are dense: a group holds 31 of the chunk's 53 columns).
// Same tables as k_train_cell3 bit for bit: equal values may rank in another order, which never changes a quantile.
// No limit on the length of the series (nothing of size T lives on chip).  float32, mode 0 / 1 (plain / EQM quantiles),
// Q <= 21, rows below 1024 values, toggle lists of at most 64 steps.
#pragma once
#include "train_cell3.hpp"

namespace xs {

template <int NJ> struct ChunkCfg {
    static constexpr int NE = NJ * 64;            // local slots (a chunk holds at most NE - 32 steps)
    static constexpr int W = NE / 32;             // words of a vector; the last one is the scratch word of NaN / padding steps
    static_assert(W <= 64 && W >= 4, "one word per lane");
    static constexpr int NLV = W - 1;             // lanes whose word counts
    static constexpr int NAN_RANK = (W - 1) * 32;
    static constexpr int MAXE = NE - 32;
    static constexpr int PADE = NE - 1;           // the local index that is no step (the tables' padding)
    static constexpr int NWAVE = 4, NT = 256;
    static constexpr int VW = 64;                 // dwords per vector in LDS (word L + 1 of any lane stays inside)
    // per wave: values in rank order (the sort's handles) | local ranks | the three vectors; the sort's 2^11 packed counters
    // overlay the ranks and the vectors
    static constexpr size_t off_rank = (size_t)NE * 4, off_cnt = off_rank + (size_t)NE * 2;
    static constexpr size_t wave_bytes = off_cnt + 1024;
    static_assert(3 * VW * 4 <= 1024 && (size_t)NE * 2 + 1024 >= 4096, "vectors and counters");
    static_assert(NE <= 2048, "a handle keeps the local index in 11 bits");
    static constexpr size_t lds_bytes = NWAVE * wave_bytes + 1024;  // + the select-in-a-byte table
};

// The chunk tables (device pointers; built by the host from the group plan):
struct ChunkTabs {
    const int *g0;        // [nch + 1] first group of every chunk
    const int *time_of;   // [nch][NE] time step of every local slot, ascending (| 1 << 30: a step of the row of the chunk's middle
                          // group, the one the walk starts from); -1 past the chunk's steps
    const int *tog;       // [G][64] toggle list of the step g - 1 -> g as local indexes of the chunk of g (lanes past the list:
                          // pairs of its own entries, else PADE); unused for the first group of a chunk
    int nch;
};

// STOP (scripts/ubench/train3_bench.hip): 1 = load + keys, 2 = + sort, 3 = + build / copies / fast-forward, 4 = + loop without select
template <int NJ, int STOP = 0>
__global__ __launch_bounds__(256, 3) void k_train_chunk(const float *__restrict__ x0, const float *__restrict__ x1, int64_t sc, int64_t C, int G,
                                                         const float *__restrict__ q, int Q, int kind, int mode, float *__restrict__ af,
                                                         float *out_hq, int *__restrict__ flag_count, int *__restrict__ flag_list,
                                                         int *__restrict__ cell_flag, ChunkTabs tb)
{
    using Cfg = ChunkCfg<NJ>;
    constexpr int NE = Cfg::NE, W = Cfg::W, NLV = Cfg::NLV, VW = Cfg::VW, QL = 21;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lane = tid & 63;
    unsigned char *wb = smem + (size_t)wid * Cfg::wave_bytes;
    float *vals = reinterpret_cast<float *>(wb);
    uint32_t *K = reinterpret_cast<uint32_t *>(wb);
    uint16_t *lrank = reinterpret_cast<uint16_t *>(wb + Cfg::off_rank);
    uint32_t *cnt = reinterpret_cast<uint32_t *>(wb + Cfg::off_cnt);
    uint32_t *vec = cnt;  // [3][VW]
    uint32_t *lut = reinterpret_cast<uint32_t *>(smem + (size_t)Cfg::NWAVE * Cfg::wave_bytes);
    byte_select_lut(lut, tid);
    __syncthreads();  // (the only block barrier: the waves are on their own from here)
    const int sid = lane / QL, k = lane - QL * sid;  // lane 63: sid 3, takes no part
    const int fsh = 10 * (sid < 3 ? sid : 0);
    uint32_t *bA = vec, *bB = vec + VW, *bC = vec + 2 * VW;
    const uint32_t *bits_s = sid == 0 ? bA : (sid == 1 ? bB : bC);
    const int64_t ntask = C * (int64_t)tb.nch;
    for (int64_t task = (int64_t)blockIdx.x * Cfg::NWAVE + wid; task < ntask; task += (int64_t)gridDim.x * Cfg::NWAVE) {
        const int64_t c = task / tb.nch;
        const int ck = (int)(task - c * tb.nch);
        const int r0 = tb.g0[ck], r1 = tb.g0[ck + 1], n = r1 - r0;
        const int nA = (n + 2) / 3, nB = (n - nA + 1) / 2, nC = n - nA - nB;  // nA >= nB >= nC; nA iterations
        const int gb = r0 + nB, gc3 = gb + nA;
        const int *ctime = tb.time_of + (size_t)ck * NE;
        bool flag_cell = false;
        const int npass_arr = mode == 0 ? 1 : 2;
        float *const hq_c = out_hq + (size_t)c * G * (size_t)Q, *const af_c = af + (mode == 0 ? (size_t)0 : (size_t)c * G * (size_t)Q);
        for (int arr = 0; arr < npass_arr; arr++) {
            __asm__ volatile("" : "+v"(lane));  // (nothing derived from the lane is carried across the passes in registers)
            const float *x = ((mode == 0 || arr == 1) ? x0 : x1) + c * sc;  // hist first, then ref (see k_train_cell)
            const bool ref_pass = mode != 0 && arr != 0;
            // ---------------- gather the chunk's steps, keys, range ----------------
            uint32_t r[NJ];
            int nv = 0;
            uint32_t kmin = 0xFFFFFFFFu, kmax = 0u, inrow = 0u;  // inrow: bit j = slot j 64 + lane is a step of the middle group's row
            {
                int tt[NJ];
#pragma unroll
                for (int j = 0; j < NJ; j++) tt[j] = ctime[j * 64 + lane];
                float v[NJ];
#pragma unroll
                for (int j = 0; j < NJ; j++) {
                    v[j] = x[tt[j] < 0 ? 0 : (tt[j] & 0x3FFFFFFF)];
                    inrow |= (uint32_t)(tt[j] >= 0 && (tt[j] & 0x40000000)) << j;
                }
#pragma unroll
                for (int j = 0; j < NJ; j++) {
                    const uint32_t u = __float_as_uint(v[j]);
                    uint32_t ok = (uint32_t)((int32_t)((u & 0x7FFFFFFFu) - 0x7F800001u) >> 31);  // all ones for a number
                    ok &= ~(uint32_t)(tt[j] >> 31);                                               // ... of a real step
                    const uint32_t vk = (u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u)) & ok;
                    nv -= (int)ok;
                    kmax = max(kmax, vk);
                    const uint32_t kk = vk | ~ok;
                    kmin = min(kmin, kk);
                    r[j] = kk;
                }
            }
#define XS_RED_STEP(ctrl, rmask)                                                                              \
            nv += __builtin_amdgcn_update_dpp(0, nv, ctrl, rmask, 0xf, false);                                \
            kmin = min(kmin, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)kmin, ctrl, rmask, 0xf, false));  \
            kmax = max(kmax, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)kmax, ctrl, rmask, 0xf, false));
            XS_RED_STEP(0x111, 0xf) XS_RED_STEP(0x112, 0xf) XS_RED_STEP(0x114, 0xf) XS_RED_STEP(0x118, 0xf)
            XS_RED_STEP(0x142, 0xa) XS_RED_STEP(0x143, 0xc)
#undef XS_RED_STEP
            const int nvalid = __builtin_amdgcn_readlane(nv, 63);
            kmin = (uint32_t)__builtin_amdgcn_readlane((int)kmin, 63);
            kmax = (uint32_t)__builtin_amdgcn_readlane((int)kmax, 63);
            if constexpr (STOP == 1) { if (nvalid == 0x7FFFFFF1) out_hq[0] = 1.f; continue; }
            if (nvalid == 0) {
                // no valid value among the chunk's steps (wave-uniform): every quantile of its groups is NaN
                const float nanq = nan_of<float>();
                for (int i = lane; i < n * Q; i += 64) {
                    const size_t o = (size_t)r0 * Q + i;
                    if (!ref_pass) hq_c[o] = nanq;
                    else af_c[o] = get_corr<float>(hq_c[o], nanq, kind);
                }
                continue;
            }
            // ---------------- sort (train_cell.hpp's handle sort on one wave): the low D1 bits of the reduced keys by a counting
            // pass out of the registers — the first pass need not be stable —, scattering HANDLES, (key bits above D1) << 11 | local
            // index; the up to 21 bits left in one or two stable passes on the handles, in place.  Counters: 2^11 packed u16 bins
            // over the rank region + the vectors' region (both dead during the sort).
            const uint32_t range = kmax - kmin;  // NaN / padding -> range + 1
            const int nbits = 32 - __clz((int)(range + 1u));
            const int D1 = min(nbits, 11);
            const int R = nbits - D1;                         // <= 21
            const int nrest = (R + 10) / 11;                  // 0..2
            const int db = nrest ? (R + nrest - 1) / nrest : 0;
            uint32_t *cntp = reinterpret_cast<uint32_t *>(lrank);  // [1024] dwords = 2048 packed counters
#pragma unroll
            for (int j = 0; j < NJ; j++) r[j] = (r[j] == 0xFFFFFFFFu) ? range + 1u : r[j] - kmin;
            uint32_t od[NJ];
            // one counting pass: digits -> od[] = final slot of every element (`dig(j)` is its digit, nbit the digit width)
            // (`touch` makes the digits' source registers opaque between the two uses: recomputed, not kept in 52 more registers)
            auto count_pass = [&](auto dig, auto touch) {
#pragma unroll
                for (int i = 0; i < 4; i++) *reinterpret_cast<uint4 *>(cntp + (i * 64 + lane) * 4) = make_uint4(0u, 0u, 0u, 0u);
                lds_wave_fence();
                // (a wave's returning atomics are served in program order, the lanes of one instruction that hit the same word in
                //  ascending lane order — train_cell.hpp: that is stability)
#pragma unroll
                for (int j = 0; j < NJ; j++) {
                    const uint32_t d = dig(j);
                    const uint32_t old = atomicAdd(&cntp[d >> 1], 1u << (16 * (d & 1u)));
                    od[j] = (old >> (16 * (d & 1u))) & 0xFFFFu;
                }
                lds_wave_fence();
                {   // exclusive scan of all 2048 counters (unused ones are zero): lane l owns the sixteen dwords [16 l, 16 l + 16), moved as
                    // four 16-byte pieces (a dword at a time they are 32 lanes on one bank)
                    uint4 c4[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) c4[i] = *reinterpret_cast<const uint4 *>(cntp + lane * 16 + i * 4);
                    int run = 0;
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        run += (int)(c4[i].x & 0xFFFFu) + (int)(c4[i].x >> 16) + (int)(c4[i].y & 0xFFFFu) + (int)(c4[i].y >> 16) +
                               (int)(c4[i].z & 0xFFFFu) + (int)(c4[i].z >> 16) + (int)(c4[i].w & 0xFFFFu) + (int)(c4[i].w >> 16);
                    int pre = wave_incl_scan(run) - run;
                    auto ex2 = [&](uint32_t c) {  // the two counters of a dword -> their exclusive prefixes
                        const int c0 = (int)(c & 0xFFFFu), c1 = (int)(c >> 16);
                        const uint32_t o = (uint32_t)pre | ((uint32_t)(pre + c0) << 16);
                        pre += c0 + c1;
                        return o;
                    };
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        uint4 o;
                        o.x = ex2(c4[i].x); o.y = ex2(c4[i].y); o.z = ex2(c4[i].z); o.w = ex2(c4[i].w);
                        *reinterpret_cast<uint4 *>(cntp + lane * 16 + i * 4) = o;
                    }
                }
                lds_wave_fence();
                touch();
#pragma unroll
                for (int j = 0; j < NJ; j++) {
                    const uint32_t d = dig(j);
                    od[j] += (cntp[d >> 1] >> (16 * (d & 1u))) & 0xFFFFu;
                }
                lds_wave_fence();  // (every read of the counters is done: the rank region may be written)
            };
            const uint32_t m1 = (1u << D1) - 1u;
            auto touch_r = [&]() {
#pragma unroll
                for (int j = 0; j < NJ; j++) __asm__ volatile("" : "+v"(r[j]));
            };
            count_pass([&](int j) { return r[j] & m1; }, touch_r);
            if (nrest == 0) {
#pragma unroll
                for (int j = 0; j < NJ; j++) lrank[j * 64 + lane] = (uint16_t)((int)od[j] >= nvalid ? Cfg::NAN_RANK : (int)od[j]);
            } else {
#pragma unroll
                for (int j = 0; j < NJ; j++) K[od[j]] = ((r[j] >> D1) << 11) | (uint32_t)(j * 64 + lane);
#pragma nounroll
                for (int p = 0; p < nrest; p++) {
                    const int shift = 11 + db * p;
                    const uint32_t md = (1u << db) - 1u;
                    uint32_t hd[NJ];
                    lds_wave_fence();
#pragma unroll
                    for (int j = 0; j < NJ; j++) hd[j] = K[j * 64 + lane];
                    auto touch_h = [&]() {
#pragma unroll
                        for (int j = 0; j < NJ; j++) __asm__ volatile("" : "+v"(hd[j]));
                    };
                    count_pass([&](int j) { return (hd[j] >> shift) & md; }, touch_h);
                    if (p + 1 < nrest) {
#pragma unroll
                        for (int j = 0; j < NJ; j++) K[od[j]] = hd[j];
                    } else {
#pragma unroll
                        for (int j = 0; j < NJ; j++) lrank[hd[j] & 2047u] = (uint16_t)((int)od[j] >= nvalid ? Cfg::NAN_RANK : (int)od[j]);
                    }
                }
            }
            if constexpr (STOP == 2) { lds_wave_fence(); if (lrank[lane] == 0xFFFE) out_hq[0] = 1.f; continue; }
            // ---------------- the values in rank order over the handles, and the vector of the middle group gb: its members are
            // marked in the chunk's step table (bit 30), so every lane flips the bits of its own steps; copied to B and C ----------------
            const int *tog = tb.tog;
            for (int i = lane; i < 3 * VW; i += 64) vec[i] = 0u;
            lds_wave_fence();
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                const int lr = (int)lrank[j * 64 + lane];
                vals[lr] = from_key(r[j] + kmin);  // (steps without a value all write the slot NAN_RANK, which nobody reads)
                if ((inrow >> j) & 1u) (void)__hip_atomic_fetch_xor(&bA[lr >> 5], 1u << (lr & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            lds_wave_fence();
            {
                const uint32_t w = bA[lane];
                bB[lane] = w; bC[lane] = w;
                lds_wave_fence();
            }
            if (nC > 0) {
                for (int s0 = 1; s0 <= nA; s0 += 8) {
                    int ts[8], rs[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) ts[u] = s0 + u <= nA ? tog[(size_t)(gb + s0 + u) * 64 + lane] : Cfg::PADE;
#pragma unroll
                    for (int u = 0; u < 8; u++) rs[u] = (int)lrank[ts[u]];
#pragma unroll
                    for (int u = 0; u < 8; u++) (void)__hip_atomic_fetch_xor(&bC[rs[u] >> 5], 1u << (rs[u] & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                lds_wave_fence();
            }
            if constexpr (STOP == 3) { if (bC[lane] == 0x12345u) out_hq[0] = 1.f; continue; }
            // ---------------- lock-step walk of the three streams (train_cell3.hpp), one word per lane ----------------
            int preA = Cfg::PADE, preB = Cfg::PADE, preC = Cfg::PADE;
            if (nB > 0) {  // step 0's flips (B only: gb -> gb - 1), then the lists of step 1
                const int r_ = (int)lrank[tog[(size_t)gb * 64 + lane]];
                (void)__hip_atomic_fetch_xor(&bB[r_ >> 5], 1u << (r_ & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (1 < nA) preA = tog[(size_t)(gb + 1) * 64 + lane];
            if (1 < nB) preB = tog[(size_t)(gb - 1) * 64 + lane];
            if (1 < nC) preC = tog[(size_t)(gc3 + 1) * 64 + lane];
            int nA_prev = -1, nB_prev = -1, nC_prev = -1, c_case = 3, c_lo = 0;
            float c_gamma = 0.f;
            const int gdir = sid == 1 ? -1 : 1, odir = gdir * Q;
            const int nlim = sid == 0 ? nA : (sid == 1 ? nB : (sid == 2 ? nC : 0));
            int g = sid == 0 ? gb : (sid == 1 ? gb - 1 : gc3);
            int o = g * Q + k;
            int Lprev4 = -1;
            for (int it = 0; it < nA; it++, g += gdir, o += odir) {
                // the ranks of the next step's flips (its lists were requested a step ago), the lists of the step after it
                const int rnA = (int)lrank[preA], rnB = (int)lrank[preB], rnC = (int)lrank[preC];
                preA = it + 2 < nA ? tog[(size_t)(gb + it + 2) * 64 + lane] : Cfg::PADE;
                preB = it + 2 < nB ? tog[(size_t)(gb - it - 2) * 64 + lane] : Cfg::PADE;
                preC = it + 2 < nC ? tog[(size_t)(gc3 + it + 2) * 64 + lane] : Cfg::PADE;
                auto next_flips = [&]() {
                    if (it + 1 < nA) (void)__hip_atomic_fetch_xor(&bA[rnA >> 5], 1u << (rnA & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (it + 1 < nB) (void)__hip_atomic_fetch_xor(&bB[rnB >> 5], 1u << (rnB & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (it + 1 < nC) (void)__hip_atomic_fetch_xor(&bC[rnC >> 5], 1u << (rnC & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                };
                lds_wave_fence();
                // per-lane member counts of the three streams in 10-bit fields (rows below 1024 values), one prefix scan
                const uint32_t wa = bA[lane], wbb = bB[lane], wc = bC[lane];
                const int cl = lane < NLV ? (__popc(wa) | (__popc(wbb) << 10) | (__popc(wc) << 20)) : 0;
                const int incl = wave_incl_scan(cl);
                const int ex = incl - cl;
                const int tot = __builtin_amdgcn_readlane(incl, 63);
                const int cntA = tot & 1023, cntB = (tot >> 10) & 1023, cntC = (tot >> 20) & 1023;
                if constexpr (STOP == 4) { if (cntA + ex + cntB + cntC == 0x7FFFFFF1) out_hq[0] = 1.f; next_flips(); continue; }
                const int ns = (int)__builtin_amdgcn_ubfe((unsigned)tot, (unsigned)fsh, 10u);
                const bool vs = it < nlim;
                const bool act = vs && k < Q;
                float hq_prev = 0.f;
                // the hist quantile of the first pass: requested here, used after the select (written by this wave; past the L1)
                if (ref_pass && act) hq_prev = __hip_atomic_load(hq_c + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool same_counts = cntA == nA_prev && cntB == nB_prev && cntC == nC_prev;
                nA_prev = cntA; nB_prev = cntB; nC_prev = cntC;
                int casek, lo;  // 0: general lerp, 1: clamp to a[n-1], 2: clamp to a[0], 3: empty
                float gamma;
                if (same_counts) { casek = c_case; lo = c_lo; gamma = c_gamma; }
                else {
                    const double qk = (double)q[k < Q ? k : 0];
                    const double vi = (double)ns * qk + (1.0 + qk * (1.0 - 1.0 - 1.0)) - 1.0;
                    lo = 0; gamma = 0.f;
                    if (ns == 0) casek = 3;
                    else if (vi != vi || vi >= (double)(ns - 1)) { casek = 1; lo = ns - 1; }
                    else if (vi < 0.0) { casek = 2; lo = 0; }
                    else { casek = 0; lo = (int)floor(vi); gamma = (float)(vi - (double)lo); }
                    c_case = casek; c_lo = lo; c_gamma = gamma;
                }
                // owner lane of member `lo`: the last lane whose exclusive count is <= lo (train_cell3.hpp: the previous step's
                // lane or a neighbour nearly always; the bisection when some lane's owner moved further)
                auto field = [&](int e_) { return (int)__builtin_amdgcn_ubfe((unsigned)e_, (unsigned)fsh, 10u); };
                int L4 = 0, eL = 0;
                bool miss = true;
                if (Lprev4 >= 0) {
                    const int a0 = Lprev4 - 4;
                    const int e_m = field(__builtin_amdgcn_ds_bpermute(a0, ex));
                    const int e_0 = field(__builtin_amdgcn_ds_bpermute(a0 + 4, ex));
                    const int e_1 = field(__builtin_amdgcn_ds_bpermute(a0 + 8, ex));
                    const int e_2 = field(__builtin_amdgcn_ds_bpermute(a0 + 12, ex));
                    const bool dn = e_0 > lo, st = e_1 > lo;
                    L4 = dn ? a0 : (st ? Lprev4 : Lprev4 + 4);
                    eL = dn ? e_m : (st ? e_0 : e_1);
                    miss = dn ? (e_m > lo || Lprev4 == 0) : (!st && e_2 <= lo);
                }
                if (__any(miss)) {
                    L4 = 0; eL = 0;
#pragma unroll
                    for (int s = 32; s > 0; s >>= 1) {
                        const int cand4 = L4 + 4 * s;
                        const int e_ = field(__builtin_amdgcn_ds_bpermute(cand4, ex));
                        const bool ok = e_ <= lo;
                        L4 = ok ? cand4 : L4;
                        eL = ok ? e_ : eL;
                    }
                }
                L4 = min(L4, 4 * (NLV - 1));
                Lprev4 = L4;
                const int L = L4 >> 2;
                int rem = lo - eL;
                // the owner lane's word and the one behind it
                const uint32_t wsel = bits_s[L], n1 = bits_s[L + 1];
                uint32_t w = wsel;
                int pos = 0;
#pragma unroll
                for (int s = 16; s > 4; s >>= 1) {  // the half, the byte, then the byte's table
                    const int cw = __popc(w & ((1u << s) - 1u));
                    const bool g2 = rem >= cw;
                    rem -= g2 ? cw : 0;
                    w = g2 ? (w >> s) : w;
                    pos += g2 ? s : 0;
                }
                pos += (int)((lut[w & 0xFFu] >> (3 * (rem & 7))) & 7u);
                const int lr = L * 32 + pos;
                const uint32_t rest = (wsel >> pos) >> 1;
                int hr = rest ? lr + __ffs((int)rest) : L * 32 + 31 + __ffs((int)n1);
                const bool need_slow = casek == 0 && (rest | n1) == 0u && sid < 3;
                if (__any(need_slow)) {
                    // the first member of the next lane that holds any: that lane from the ballot of the non-empty lanes of the stream
                    const unsigned long long balA = __ballot((cl & 1023) != 0), balB = __ballot(((cl >> 10) & 1023) != 0), balC = __ballot(((cl >> 20) & 1023) != 0);
                    const unsigned long long bal = sid == 0 ? balA : (sid == 1 ? balB : balC);
                    const unsigned long long restm = L < 63 ? (bal >> (L + 1)) : 0ull;
                    const int L2 = min(restm ? L + __ffsll(restm) : L, NLV - 1);
                    const uint32_t w2 = bits_s[L2];
                    if (need_slow) hr = L2 * 32 + __ffs((int)w2) - 1;
                }
                next_flips();  // the vectors have been read for the last time in this step
                const int rs0 = min(max(lr, 0), NE - 1), rs1 = min(max(casek == 0 ? hr : lr, 0), NE - 1);
                const float v = vals[rs0], vhi = vals[rs1];
                float res;
                if (casek == 3) res = nan_of<float>();
                else if (casek != 0) res = v;
                else {
                    const float diff = vhi - v;
                    if (gamma >= 0.5f) res = (float)((double)vhi - (double)diff * (1.0 - (double)gamma));
                    else res = v + diff * gamma;
                    if (res != res && act) flag_cell = true;  // inf - inf: the generic kernel redoes the cell
                }
                if (act) {
                    if (!ref_pass) hq_c[o] = res;
                    else af_c[o] = get_corr<float>(hq_prev, res, kind);
                }
            }
        }
        if (__any(flag_cell) && lane == 0 && flag_list) {
            if (atomicExch(&cell_flag[c], 1) == 0) {  // (a cell is listed once, whichever of its chunks asks)
                const int slot = atomicAdd(flag_count, 1);
                flag_list[slot] = (int)c;
            }
        }
    }
}

}  // namespace xs
