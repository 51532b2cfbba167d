// order_probe.hpp — run-time check of the one undocumented hardware property the per-cell sorts stand on (train_cell.hpp):
// a returning LDS atomic issued by ONE wave instruction serves the lanes that hit the same word in ASCENDING LANE ORDER (and a
// wave's instructions in program order), so that `old` = counter before + number of lower lanes with the same digit — which is
// what makes the counting passes on wave-private counter rows stable.  Measured offline over 1.4e11 atomics
// (scripts/ubench/atomic_order.hip -> profiles/r04_atomic_order.txt); a firmware or silicon revision that changed it would give
// silently wrong quantiles, so every context probes it once at creation (~0.5 ms; 3.7 million atomics): lane -> digit maps that are uniform, skewed,
// in runs, mirrored, descending, with a hot bin and strided, on plain and on packed 16-bit counters, sixteen waves of a block
// hammering their own rows at once.  A context whose probe fails takes the generic kernels (k_train_rows / k_qdm_adjust: bitonic
// sorts and counting, no such assumption) for everything the per-cell sorts would have served.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace xs {

__device__ __forceinline__ uint32_t probe_mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__device__ __forceinline__ uint32_t probe_digit(int p, uint32_t it, int b, int w, int l, uint32_t nd)
{
    const uint32_t r = probe_mix(it * 0x9E3779B9u + (uint32_t)b * 7919u + (uint32_t)w * 104729u + (uint32_t)l * 2654435761u);
    switch (p) {
    case 0: return r % nd;                                               // uniform
    case 1: return (probe_mix(r) % nd) * (probe_mix(r >> 3) % nd) / nd;  // skewed towards 0
    case 2: return ((uint32_t)l / (1u + it % 7u) + it) % nd;             // runs of equal digits
    case 3: return ((uint32_t)(l & 31) * (1u + it % 5u)) % nd;           // both halves of the wave alike
    case 4: return ((uint32_t)(63 - l) + r % 3u) % nd;                   // descending
    case 5: return (r % 3u == 0) ? (it % nd) : (r % nd);                 // a hot bin
    default: return ((uint32_t)l * (it % 64u + 1u)) % nd;                // strided
    }
}

constexpr int PROBE_ND = 256;  // digits per wave-private row (the sorts use up to 256 plain, 2^14 packed)

__global__ __launch_bounds__(1024) void k_atomic_order_probe(int rounds, unsigned long long *bad)
{
    __shared__ uint32_t cnt[16][PROBE_ND + 8];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    uint32_t *mine = cnt[w];
    unsigned long long nbad = 0;
    for (int packed = 0; packed < 2; packed++)
        for (int p = 0; p < 7; p++)
            for (uint32_t nd = 16; nd <= (uint32_t)PROBE_ND; nd *= 16) {  // 16 digits (long same-word runs) and 256 (the sorts' widest rows)
                for (uint32_t i = lane; i < (uint32_t)PROBE_ND + 8; i += 64) mine[i] = 0u;
                __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                for (int it = 0; it < rounds; it++) {
                    const uint32_t d = probe_digit(p, (uint32_t)it, blockIdx.x, w, lane, nd);
                    uint32_t before, old;
                    if (packed) {
                        before = (mine[d >> 1] >> (16 * (d & 1))) & 0xFFFFu;
                        __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        old = (atomicAdd(&mine[d >> 1], 1u << (16 * (d & 1))) >> (16 * (d & 1))) & 0xFFFFu;
                    } else {
                        before = mine[d];
                        __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        old = atomicAdd(&mine[d], 1u);
                    }
                    __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    uint32_t below = 0;
                    for (int l = 0; l < 64; l++) {
                        const uint32_t dl = (uint32_t)__builtin_amdgcn_readlane((int)d, l);
                        below += (l < lane && dl == d) ? 1u : 0u;
                    }
                    if (old != before + below) nbad++;
                }
            }
    if (nbad) atomicAdd(bad, nbad);
}

}  // namespace xs
